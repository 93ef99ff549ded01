"""Pin the oracle (CPU restatement) against every fixture the reference's own tests hold for the hot path
(SURVEY.md §8c).  CPU only."""
import numpy as np
import pytest

import helpers
from lariat_amd import capi


def test_phix_index_bytes_reproduced(oracle):
    """Fixture go/src/test/inputs/phix/PhiX.fa.{bwt,sa,pac,ann,amb}: rebuilding the index from the FASTA must give
    byte-identical files, and loading + re-serialising must round-trip."""
    names, seqs = helpers.read_fasta(helpers.PHIX)
    built = oracle.index_build_naive(names, [capi.sequence_convert(s) for s in seqs])
    loaded = oracle.index_load(helpers.PHIX)
    for which, ext in enumerate(["bwt", "sa", "pac", "ann", "amb"]):
        ref = open(helpers.PHIX + "." + ext, "rb").read()
        assert built.image(which) == ref, ext
        assert loaded.image(which) == ref, ext


def test_gobwa1(oracle):
    """go/src/test/gobwa_test.go:13-28 TestGobwa1"""
    ref = oracle.index_load(helpers.PHIX)
    assert ref.contigs()[0][0] == "PhiX"
    b = capi.Batch([capi.sequence_convert(helpers.PHIX_READ_A), capi.sequence_convert(helpers.PHIX_READ_B)], [0, 1])
    res = ref.align_barcodes(b, oracle.opts(run_inference=0))
    a0 = list(res.cands_of_read(0))
    assert res.pos[a0[0]] == 210                       # Check(algns[0].Offset == 210)
    assert ref.contigs()[res.rid[a0[0]]][0] == "PhiX"  # Check(algns[0].Contig == "PhiX")
    a1 = list(res.cands_of_read(1))
    assert len(a1) == 1                                # Check(len(algns) == 1)
    # what the survey derived by hand: forward at 2170, one mismatch at read offset 67
    assert res.pos[a1[0]] == 2170 and not res.reversed[a1[0]] and res.nm[a1[0]] == 1
    assert list(res.mm_read_loc[res.mm_off[a1[0]]:res.mm_off[a1[0] + 1]]) == [67]
    assert res.cigar_str(a1[0]) == "70M" and res.cigar_str(a0[0]) == "70M"


def test_lariat_zero_length_read(oracle):
    """go/src/test/lariat_test.go:12-24: empty read1 / unalignable reads must not crash; every read_id gets >= 1
    (placeholder) entry."""
    ref = oracle.index_load(helpers.PHIX)
    recs = helpers.read_fastq9(helpers.GOLDEN + "/zero_length_read_test.fastq.gz", trim=7)
    assert len(recs) == 3 and recs[1]["r1"] == ""
    reads = []
    for r in recs:
        reads += [capi.sequence_convert(r["r1"]), capi.sequence_convert(r["r2"])]
    b = capi.Batch(reads, [0, 3])
    res = ref.align_barcodes(b, oracle.opts(improper_pair_penalty=-17.0, run_inference=0))
    assert res.n_reads == 6
    assert (np.diff(res.cand_off) >= 1).all()
    # human reads vs PhiX: placeholders (rid -1, pos -1)
    assert (res.rid == -1).all() and (res.pos == -1).all()
    # and the full inference path survives placeholders too
    res = ref.align_barcodes(b, oracle.opts(improper_pair_penalty=-17.0))
    assert (res.active_idx >= 0).all()


def test_get_seq(oracle):
    """GoBwaReference.GetSeq (gobwa.go:50-80)"""
    ref = oracle.index_load(helpers.PHIX)
    _, seqs = helpers.read_fasta(helpers.PHIX)
    fa = seqs[0]
    assert ref.get_seq(0, 210, 280, False).decode() == helpers.PHIX_READ_A
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[c] for c in reversed(fa[100:160]))
    assert ref.get_seq(0, 100, 160, True).decode() == rc


def test_revcomp_read_maps_reversed(oracle):
    ref = oracle.index_load(helpers.PHIX)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[c] for c in reversed(helpers.PHIX_READ_A))
    b = capi.Batch([capi.sequence_convert(rc), capi.sequence_convert(helpers.PHIX_READ_A)], [0, 1])
    res = ref.align_barcodes(b, oracle.opts(run_inference=0))
    a = list(res.cands_of_read(0))[0]
    assert res.reversed[a] == 1 and res.pos[a] == 210 and res.aend[a] == 280 and res.cigar_str(a) == "70M"


def test_oracle_end_to_end_synthetic(oracle):
    names, contigs = helpers.small_genome()
    idx = oracle.index_build_naive(names, contigs)
    rs = helpers.small_reads(names, contigs)
    res = idx.align_barcodes(helpers.batch_of(rs), threads=4)
    act = res.active_idx
    assert (act >= 0).all()
    ok = 0
    for p in range(rs.n_pairs):
        for m in (0, 1):
            a = act[2 * p + m]
            truth = rs.truth_pos1[p] if m == 0 else rs.truth_pos2[p]
            ok += int(res.rid[a] == rs.truth_contig[p] and abs(res.pos[a] - truth) < 20)
    assert ok >= 0.95 * 2 * rs.n_pairs
    # exactly one active candidate per read
    for r in range(res.n_reads):
        c = list(res.cands_of_read(r))
        assert res.active[c].sum() == 1
    # threading over barcodes does not change results (barcodes are independent, lariat.go:348-350)
    res1 = idx.align_barcodes(helpers.batch_of(rs), threads=1)
    for f in ("pos", "mapq", "active", "molecule_id", "score"):
        assert (getattr(res, f) == getattr(res1, f)).all(), f
