"""N4 — accuracy accounting for simulated reads (lariat_amd/simulated.py; lariat.go:517-542 and go/check.py:41-105).
Alignments come from the CPU oracle, records from the product's lh_records_text; test_gpu_inference.py repeats it on the HIP result."""
import numpy as np
import pytest

import helpers
from lariat_amd import capi, simulated, synth


@pytest.fixture(scope="module")
def hostlib():
    import os
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    return capi.Library(os.environ.get("LH_HOST_LIB") or ge.LIB)   # (LH_HOST_LIB: the host code in another build, e.g. the emulator's AddressSanitizer one)


def test_truth_of():
    assert simulated.truth_of("mol:ACGT-1:chr3:100:90000:80771341:80771523") == ("chr3", 80771341.0, 80771523.0)
    assert simulated.truth_of(b"mol:x:c:1:2:3:4 \n".strip()) == ("c", 3.0, 4.0)
    assert simulated.truth_of("read7") is None and simulated.truth_of("mol:a:b:c:d:e:f") is None


def test_check_report_bins_and_fractions():
    nm = "mol:b:chrA:0:50000:1000:1300"
    lines = ["%s\t99\tchrA\t1001\t60\t150M\t=" % nm,          # correct, bin 45
             "%s\t147\tchrA\t1301\t60\t150M\t=" % nm,         # read 2: wrong by check.py's rule (compared with field 5), right when mate aware
             "%s\t65\tchrB\t1001\t9\t150M\t=" % nm,           # wrong contig, bin 5
             "%s\t69\t*\t0\t0\t*\t*" % nm,                    # unmapped
             "@HD\tVN:1.3"]
    r = simulated.check_report(lines)
    assert r["records"] == 4 and r["Unmapped"] == 0.25 and r["Proper pair"] == 0.5 and r["mapq = 0"] == 0.25 and r["mapq >= 30"] == 0.5
    b = {x["bin"]: x for x in r["mapq_bins"]}
    assert b[45]["n"] == 2 and b[45]["frac_correct"] == 0.5 and abs(b[45]["emp_mapq"] - 3.0103) < 1e-3 and b[5]["n"] == 2 and b[5]["frac_correct"] == 0
    b2 = {x["bin"]: x for x in simulated.check_report(lines, mate_aware=True)["mapq_bins"]}
    assert b2[45]["frac_correct"] == 1.0 and b2[45]["emp_mapq"] == float("inf")
    assert "mapq bin 45" in simulated.format_report(r)


def test_stats_and_report_on_oracle_alignments(hostlib, oracle, tmp_path):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=6, pairs_per_barcode=40, seed=77, junk_frac=0.05)
    p = tmp_path / "r.fastq"
    p.write_text(synth.to_fastq9(rs, trim_prefix=7))
    st = simulated.SimulatedStats()
    text = []
    n_active_rfa = 0
    for b in hostlib.ingest(str(p), trim=7):
        res = oidx.align_barcodes(b, threads=4)
        nm = b.column("name")
        st.add(res, nm, b.bc_pair_off, b.bc_do_rfa)
        read_of = np.searchsorted(res.cand_off, np.flatnonzero(res.active), side="right") - 1
        bc_of = np.searchsorted(b.bc_pair_off, read_of >> 1, side="right") - 1
        n_active_rfa += int((b.bc_do_rfa[bc_of] != 0).sum())
        text += hostlib.records_text(res, b, names).splitlines()
    d = st.as_dict()
    assert d["total"] == n_active_rfa > 0 and d["correct_mapq10"] <= d["total_mapq10"] <= d["total"] and d["correct"] <= d["total"]
    # reads without any hit (the junk pairs) stay active placeholders with pos -1: the reference counts them too
    assert d["placeholders"] > 0 and d["correct_mapq10"] / (d["total_mapq10"] - d["placeholders"]) > 0.99
    rep = simulated.check_report(text, mate_aware=True, include_secondary=False)
    assert rep["records"] == 2 * rs.n_pairs
    top = [x for x in rep["mapq_bins"] if x["bin"] == 45][0]
    assert top["frac_correct"] > 0.99 and top["n"] > rs.n_pairs
