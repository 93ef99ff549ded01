#!/usr/bin/env python3
"""where lh_bam_append's time goes on a many-core host: one aligned chunk, appended under several thread settings (one writer; several writers at once)"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi, workload  # noqa: E402

lib = capi.load_library()
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 400
nbc = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ctg = workload.hg38_like_contigs(int(mb * 1e6))
l_pac = sum(c[1] for c in ctg)
pac = lib.synth_genome(l_pac, seed=workload.GENOME_SEED)
idx = lib.index_build_device(pac, l_pac, ctg)
r = lib.synth_reads(pac, l_pac, ctg, seed=7, n_barcodes=nbc, pairs_per_barcode=100)
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
d = tempfile.mkdtemp(prefix="lh_bamprobe_", dir=base)
fq = os.path.join(d, "c.fastq.gz")
lib.write_fastq9(fq, r, gz_level=1)
t = time.time()
rd = lib.ingest(fq, trim=7, max_pairs=nbc * 100)
b = rd.next(views_only=True)
print("ingest: %d pairs in %.2f s = %.0f pairs/s (one reader)" % (b.n_pairs, time.time() - t, b.n_pairs / (time.time() - t)))
res = idx.context(b.n_pairs).align_barcodes(b)
cont = idx.contigs()
names, lens = [c[0] for c in cont], [c[1] for c in cont]
lib.L.lh_bam_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]


def one(tag, threads, level=None):
    out = os.path.join(d, tag)
    os.makedirs(out)
    w = lib.bam_writer(out, names, lens, threads=threads)
    if level is not None:
        lib.L.lh_bam_set_level(w.h, level)
    t0 = time.time()
    w.append(res, b)
    ta = time.time() - t0
    x, y, z = C.c_double(), C.c_double(), C.c_double()
    lib.L.lh_bam_timings(w.h, C.byref(x), C.byref(y), C.byref(z))
    t0 = time.time()
    w.close()
    tc = time.time() - t0
    shutil.rmtree(out)
    return ta, tc, x.value, y.value, z.value


for th in (1, 8, 32, 64, 128, 256):
    ta, tc, x, y, z = one("t%d" % th, th)
    print("1 writer x %3d threads: append %.2f s (records %.2f, join %.2f, compress+write %.2f), close %.2f s -> %.0f pairs/s" % (th, ta, x, y, z, tc, b.n_pairs / (ta + tc)))
ta, tc, x, y, z = one("l1", 64, level=1)
print("1 writer x  64 threads, zlib level 1: append %.2f s (records %.2f, join %.2f, compress+write %.2f) -> %.0f pairs/s" % (ta, x, y, z, b.n_pairs / (ta + tc)))
for nw, th in ((8, 30), (4, 60), (2, 120)):
    outs = [None] * nw
    ths = [threading.Thread(target=lambda k=k: outs.__setitem__(k, one("w%d_%d" % (nw, k), th))) for k in range(nw)]
    t0 = time.time()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.time() - t0
    print("%d writers x %3d threads at once: %.2f s wall -> %.0f pairs/s in total; first writer: append %.2f (records %.2f, join %.2f, compress+write %.2f)"
          % (nw, th, dt, nw * b.n_pairs / dt, outs[0][0], outs[0][2], outs[0][3], outs[0][4]))
shutil.rmtree(d, ignore_errors=True)
