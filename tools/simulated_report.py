"""N4 on the GPU: synthetic genome WITH planted duplications / repeat families (so that MAPQ spans its range) -> 9-line FASTQ ->
lh_ingest -> K1..K8 (HIP) -> lh_records_text -> lariat's -simulated counters (lariat.go:517-542) + the check.py report
(go/check.py:41-105).  Telemetry: how well the MAPQ the pipeline reports matches the empirical error rate on simulated reads.

  python tools/simulated_report.py [--genome-mb 16] [--barcodes 2000] [--n-dup 60] [--rep-families 20]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lariat_amd import capi, simulated, synth

ap = argparse.ArgumentParser()
ap.add_argument("--genome-mb", type=float, default=16.0); ap.add_argument("--barcodes", type=int, default=2000)
ap.add_argument("--n-dup", type=int, default=60); ap.add_argument("--dup-identity", type=float, default=0.99)
ap.add_argument("--rep-families", type=int, default=20); ap.add_argument("--max-pairs", type=int, default=1 << 17)
a = ap.parse_args()
lib = capi.load_library()
t0 = time.perf_counter()
L = int(a.genome_mb * 1e6)
names = ["chr20", "chr21"]
contigs = synth.make_genome([L * 5 // 8, L * 3 // 8], seed=20261002, n_dup=a.n_dup, dup_identity=a.dup_identity, n_rep_family=a.rep_families)
prefix = "/tmp/lh_sim_%d_%d_%d.fa" % (L, a.n_dup, a.rep_families)
if not os.path.exists(prefix + ".bwt"):
    lib.index_build(prefix, names, contigs, threads=0)
rs = synth.make_reads(contigs, names, n_barcodes=a.barcodes, pairs_per_barcode=100, seed=20261008)
fq = "/tmp/lh_sim_reads.fastq"
open(fq, "w").write(synth.to_fastq9(rs, trim_prefix=7))
idx = lib.index_load(prefix)
ctx = idx.context(a.max_pairs)
st = simulated.SimulatedStats()
lines = []
n = 0
for b in lib.ingest(fq, trim=7, max_pairs=a.max_pairs):
    res = ctx.align_barcodes(b)
    st.add(res, b.column("name"), b.bc_pair_off, b.bc_do_rfa)
    lines += lib.records_text(res, b, names).splitlines()
    n += b.n_pairs
print("genome %.0f Mb (%d duplications at %.1f %% identity, %d repeat families), %d pairs, %.1f s" %
      (a.genome_mb, a.n_dup, 100 * a.dup_identity, a.rep_families, n, time.perf_counter() - t0))
print("-simulated counters (lariat.go:517-542):", st.as_dict())
print("check.py report, its own correctness rule (both mates against field 5, |d| < 200):")
print(simulated.format_report(simulated.check_report(lines)))
print("check.py report, mate-aware (read 2 against field 6), primary records only:")
print(simulated.format_report(simulated.check_report(lines, mate_aware=True, include_secondary=False)))
