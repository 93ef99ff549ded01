"""the whole drop-in path on one GPU: 9-line FASTQ -> lh_ingest -> K1..K8 -> lh_bam_* (bc_sorted + position buckets).
Prints the rate of every stage; one host thread feeds the GPU here (a production driver runs one reader per FASTQ chunk).

  python tools/fastq_to_bam.py <index prefix> <reads.fastq[.gz]> <out dir> [--max-pairs N] [--threads T]
  python tools/fastq_to_bam.py --demo [--barcodes N]     # synthetic genome + reads under /tmp"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("index", nargs="?"); ap.add_argument("fastq", nargs="?"); ap.add_argument("out", nargs="?")
ap.add_argument("--demo", action="store_true"); ap.add_argument("--barcodes", type=int, default=2000)
ap.add_argument("--max-pairs", type=int, default=1 << 18); ap.add_argument("--threads", type=int, default=0); ap.add_argument("--trim", type=int, default=7)
a = ap.parse_args()
lib = capi.load_library()
if a.demo:
    contigs = synth.make_genome([16000000], seed=20261002)
    a.index, a.fastq, a.out = "/tmp/lh_demo_genome.fa", "/tmp/lh_demo_reads.fastq", "/tmp/lh_demo_bams"
    if not os.path.exists(a.index + ".bwt"):
        lib.index_build(a.index, ["chr20"], contigs, threads=0)
    rs = synth.make_reads(contigs, ["chr20"], n_barcodes=a.barcodes, pairs_per_barcode=100, seed=5)
    open(a.fastq, "w").write(synth.to_fastq9(rs))
os.makedirs(a.out, exist_ok=True)
idx = lib.index_load(a.index)
cont = idx.contigs()
names, lens = [c[0] for c in cont], [c[1] for c in cont]
w = lib.bam_writer(a.out, names, lens, command_line=" ".join(sys.argv), threads=a.threads)
ctx = idx.context(a.max_pairs)
opts = lib.opts()
t_in = t_gpu = t_out = 0.0
n = 0
t0 = time.perf_counter()
rd = lib.ingest(a.fastq, trim=a.trim, max_pairs=a.max_pairs)
while True:
    t = time.perf_counter(); b = rd.next(); t_in += time.perf_counter() - t
    if b.n_pairs == 0:
        break
    t = time.perf_counter(); res = ctx.align_barcodes(b, opts); t_gpu += time.perf_counter() - t
    t = time.perf_counter(); w.append(res, b); t_out += time.perf_counter() - t
    n += b.n_pairs
    if b.at_eof:
        break
t = time.perf_counter(); w.close(); t_out += time.perf_counter() - t
dt = time.perf_counter() - t0
print("pairs %d  wall %.2f s = %.3f M pairs/s end to end (stages run one after the other in this tool)" % (n, dt, n / dt / 1e6))
print("  ingest   %.2f s  %.2f M pairs/s (1 thread)" % (t_in, n / t_in / 1e6))
print("  align    %.2f s  %.2f M pairs/s (upload + K1..K8 + download)" % (t_gpu, n / t_gpu / 1e6))
print("  bam      %.2f s  %.2f M pairs/s (%d files)" % (t_out, n / t_out / 1e6, len(os.listdir(a.out))))
