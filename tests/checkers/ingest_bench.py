"""host-side throughput of the 9-line FASTQ reader (lh_ingest_*) against the Python restatement of reader.go"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from lariat_amd import capi, synth
import __graft_entry__ as ge
import fastq_oracle

lib = capi.Library(ge.LIB)
contigs = synth.make_genome([2000000], seed=3)
rs = synth.make_reads(contigs, ["chr20"], n_barcodes=int(sys.argv[1]) if len(sys.argv) > 1 else 2000, pairs_per_barcode=100, seed=4)
path = "/tmp/lh_ingest_bench.fastq"
open(path, "w").write(synth.to_fastq9(rs))
mb = os.path.getsize(path) / 1e6
t = time.perf_counter(); n = 0
for b in lib.ingest(path, trim=7, max_pairs=1 << 18):
    n += b.n_pairs
dt = time.perf_counter() - t
print("lh_ingest : %d pairs, %.0f MB in %.2f s = %.2f M pairs/s, %.0f MB/s (one host thread)" % (n, mb, dt, n / dt / 1e6, mb / dt))
t = time.perf_counter()
sets = fastq_oracle.read_all(path, 7)
dt2 = time.perf_counter() - t
print("oracle(py): %d pairs in %.2f s = %.3f M pairs/s" % (sum(len(r) for r, _, _ in sets), dt2, n / dt2 / 1e6))
