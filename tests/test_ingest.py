"""N2 — the 9-line FASTQ reader (lariat_amd/csrc/ingest.cpp) against the restatement of fastqreader/reader.go in
oracle/fastq_oracle.py: set boundaries, the complete / RFA flags, trimming, nt4 conversion, tie-break seeds, text columns.
Host-only: runs without a GPU."""
import gzip
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

import helpers
from lariat_amd import capi

sys.path.insert(0, os.path.join(helpers.ROOT, "oracle"))
import fastq_oracle  # noqa: E402

GOLDEN = os.path.join(helpers.ROOT, "tests", "golden", "zero_length_read_test.fastq.gz")


@pytest.fixture(scope="module")
def hostlib():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    return capi.Library(os.environ.get("LH_HOST_LIB") or ge.LIB)   # (LH_HOST_LIB: the host code in another build, e.g. the emulator's AddressSanitizer one)


def check_against_oracle(hostlib, path, trim, cap=0, chunk=0, max_pairs=1 << 20):
    want = fastq_oracle.read_all(path, trim, cap or 30000, chunk or 200)
    got_sets = 0
    pairs_seen = 0
    rd = hostlib.ingest(path, trim=trim, cap=cap, chunk=chunk, max_pairs=max_pairs)
    for b in rd:
        assert b.first_set_index == got_sets
        cols = {c: b.column(c) for c in ("name", "rgid", "qual1", "qual2", "trim_bases", "trim_quals", "bc", "rawbc", "bcqual", "si", "siqual")}
        if max_pairs < (1 << 20) and b.n_sets > 1:
            assert b.n_pairs <= max_pairs   # only a single oversize set may exceed the budget
        for s in range(b.n_sets):
            recs, complete, do_rfa = want[got_sets]
            p0, p1 = int(b.bc_pair_off[s]), int(b.bc_pair_off[s + 1])
            assert p1 - p0 == len(recs), (got_sets, p1 - p0, len(recs))
            assert bool(b.set_complete[s]) == complete and bool(b.bc_do_rfa[s]) == do_rfa, got_sets
            for k, r in enumerate(recs):
                p = p0 + k
                assert cols["name"][p] == r.name and cols["rgid"][p] == r.rgid
                assert cols["bc"][p] == r.bc and cols["rawbc"][p] == r.rawbc and cols["bcqual"][p] == r.bcq
                assert cols["si"][p] == r.si and cols["siqual"][p] == r.siq
                assert cols["qual1"][p] == r.q1 and cols["qual2"][p] == r.q2
                assert cols["trim_bases"][p] == r.tb and cols["trim_quals"][p] == r.tq
                assert np.array_equal(b.read(2 * p), capi.sequence_convert(r.r1)) and np.array_equal(b.read(2 * p + 1), capi.sequence_convert(r.r2))
                assert int(b.name_seed[p]) == fastq_oracle.name_seed(r.name)
            got_sets += 1
        pairs_seen += b.n_pairs
    assert got_sets == len(want)
    return want


def test_name_seed_is_md5():
    import __graft_entry__ as ge
    lib = capi.Library(ge.LIB)
    for n in (0, 1, 8, 55, 56, 57, 63, 64, 65, 119, 120, 121, 200):
        name = bytes((37 * i + n) % 251 for i in range(n))
        assert lib.name_seed(name) == struct.unpack("<Q", hashlib.md5(name).digest()[:8])[0]


def test_reference_fixture(hostlib):
    """the reference's own input for test/lariat_test.go (an empty read 1 among ordinary records)"""
    want = check_against_oracle(hostlib, GOLDEN, trim=7)
    assert sum(len(r) for r, _, _ in want) > 0
    check_against_oracle(hostlib, GOLDEN, trim=0)
    check_against_oracle(hostlib, GOLDEN, trim=7, max_pairs=3)


def make_fastq(spec, seed=3, tail=""):
    """spec: list of (barcode line, n records); returns 9-line FASTQ text"""
    rng = np.random.default_rng(seed)
    out = []
    n = 0
    for bc, cnt in spec:
        for _ in range(cnt):
            l1, l2 = int(rng.integers(0, 40)), int(rng.integers(1, 40))
            r1 = "".join(rng.choice(list("ACGTNacgt"), size=l1))
            r2 = "".join(rng.choice(list("ACGTN"), size=l2))
            hdr = "@read%d" % n + ("" if n % 3 == 0 else " some text RG:%d" % (n % 5))
            out += [hdr, r1, "I" * l1, r2, "J" * l2, bc, "K" * 16, "ACGTACGT", "L" * 8]
            n += 1
    return "\n".join(out) + "\n" + tail


def test_work_unit_rules(hostlib, tmp_path):
    # whitelisted barcodes of various sizes (RFA needs >= 5 pairs), corrected,raw barcode lines, a barcode that is not
    # whitelisted (cut every `chunk` pairs), one that exceeds the cap (incomplete set, then the "abnormal break" rule)
    spec = [("AAAC-1", 3), ("AAAG-1,AAAT", 7), ("AACC-1", 5), ("AAGG", 47), ("ACCC-1", 131), ("ACGG-1", 4), ("AGGG", 20), ("ATTT-1", 60), ("CAAA-1", 1)]
    text = make_fastq(spec)
    p = tmp_path / "a.fastq"
    p.write_text(text)
    want = check_against_oracle(hostlib, str(p), trim=7, cap=50, chunk=20)
    sizes = [len(r) for r, _, _ in want]
    assert 50 in sizes and 20 in sizes            # the cap and the non-whitelist chunk both occurred
    assert any(not c for _, c, _ in want)          # an incomplete set
    assert any(d for _, _, d in want) and not all(d for _, _, d in want)
    for mp in (1, 10, 64):                         # batches of whole sets
        check_against_oracle(hostlib, str(p), trim=7, cap=50, chunk=20, max_pairs=mp)
    gz = tmp_path / "a.fastq.gz"
    gz.write_bytes(gzip.compress(text.encode()))
    check_against_oracle(hostlib, str(gz), trim=7, cap=50, chunk=20)


def test_malformed_tails(hostlib, tmp_path):
    base = make_fastq([("AAAC-1", 6), ("AAAG-1", 6)], seed=5)
    cases = {
        "junk_before": "this is not a record\n\n" + base,
        "no_final_newline": base[:-1],
        "truncated_record": base + "@late\nACGT\nIIII\nAC\n",
        "partial_header": base + "@late",
        "empty": "",
        "only_junk": "x\ny\n",
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".fastq")
        p.write_text(text)
        check_against_oracle(hostlib, str(p), trim=7)


def test_default_constants_at_scale(hostlib, tmp_path):
    """the reference's own constants: a 30,000-pair cap and 200-pair chunks"""
    spec = [("AAAC-1", 30450), ("GGGG", 450), ("TTTT-1", 12)]
    p = tmp_path / "big.fastq"
    p.write_text(make_fastq(spec, seed=9))
    want = check_against_oracle(hostlib, str(p), trim=7)
    sizes = [len(r) for r, _, _ in want]
    assert sizes[0] == 30000 and not want[0][1]   # cut at the cap, flagged incomplete, no RFA
    assert sizes[1] == 201 and not want[1][1]     # "abnormal break" of the continuation
    assert 200 in sizes


# ---- the reference's own ingest goldens: go/src/test/fastq_reader_test.go on inputs/1.fq (a gz file despite its name) ----
FQ1 = os.path.join(helpers.ROOT, "tests", "golden", "1.fq")
# fastq_reader_test.go:19-27 — the third record, read with trim 2
REC3 = dict(
    r1="CCGCCCTAGCCAGGAGAGAAGCACTTCTTACCTGGGTTTCTTAGAGGCTTTGGCTGGCAATATTGTCAGCACCAGAGAGGACTTCTCGATGGCTGA",
    q1="BFFFFFFFFFFIIIIIFFIIIIIIIIFIIIIIFIFIFFIIFIIIIIIIIIIIIIIIFFFFFFFFFFFFFFFFFFFBFFFFFFFFFFFFFFFFFFFF",
    r2="GTGGTAGTCTCCTGTTCAGCCATCGAGAAGTCCTCTCTGGTGCTGACAATATTGCCAGCCAAAGCCTCTAAGAAACCCAGGTAAGAAGTGCTTCTCTC",
    q2="BBBFFFFFFFFFFIIFIIFIIIIIIIIIIIIIIIIFIIIFFFIIIIIIIIIIIIIIIIIIIIFIIIIIIIIIIFFFFFFFFBBFFFFFBBFFFFFFFF",
    bc="AAACAGAGAAAGAT", bcq="BBBFFFFFFFFFFI", si="CCGAACGC", siq="BBBFFFFF", name="HWI-D00684:80:HFCKCADXX:2:2113:9410:56703")
# fastq_reader_test.go:34-40 — ReadBarcodeSet twice
SET1_FIRST_NAME = "HWI-D00684:80:HFCKCADXX:2:2113:17628:14813"
SET1_SECOND_R1 = "CTGCTGCTCTCTCCATGTTTTTCCTGCACTCCTTGCAGGGACCTGAATAGCATGAACTGACTTTTCCTTGACGTAGTTGCTTCGTAGGATACTTCT"
SET2_FIRST_NAME = "HWI-D00684:80:HFCKCADXX:2:2112:14227:100270"
SET2_SECOND_R1 = "CGGGCAGCAGCCATGGGATGCAGGACCTGCAGTCCACACATGTCACATGAATCTCCATGGAGAGGCACACAGTTCTCCCCATCTCAGCACTCTCTC"


def _acgt(a):
    return "".join("ACGTN"[v] for v in a)


def test_reference_goldens_oracle():
    """oracle/fastq_oracle.py against the strings the reference's own tests assert"""
    sets = fastq_oracle.read_all(FQ1, 2, 30000, 200)
    recs = [r for s in sets for r in s[0]]
    r = recs[2]
    assert (r.r1.decode(), r.q1.decode(), r.r2.decode(), r.q2.decode()) == (REC3["r1"], REC3["q1"], REC3["r2"], REC3["q2"])
    assert (r.bc.decode(), r.bcq.decode(), r.si.decode(), r.siq.decode(), r.name.decode()) == (REC3["bc"], REC3["bcq"], REC3["si"], REC3["siq"], REC3["name"])
    assert sets[0][0][0].name.decode() == SET1_FIRST_NAME and sets[0][0][1].r1.decode() == SET1_SECOND_R1
    assert sets[1][0][0].name.decode() == SET2_FIRST_NAME and sets[1][0][1].r1.decode() == SET2_SECOND_R1
    # structure of the fixture: 666 complete records + a truncated tail, 89 barcode sets, none of them whitelisted
    assert len(recs) == 666 and len(sets) == 89
    assert [len(s[0]) for s in sets[:6]] == [9, 11, 4, 16, 9, 8] and max(len(s[0]) for s in sets) == 25
    assert not any(s[2] for s in sets)   # barcodes without '-': worthRunningRFA is false for every set


def test_reference_goldens_product(hostlib):
    """lh_ingest_* (the product's reader) against the same strings, then against the oracle record by record"""
    rd = hostlib.ingest(FQ1, trim=2, max_pairs=1 << 20)
    batches = list(rd)
    assert len(batches) == 1
    b = batches[0]
    assert b.n_pairs == 666 and b.n_sets == 89
    cols = {c: b.column(c) for c in ("name", "qual1", "qual2", "bc", "bcqual", "si", "siqual", "trim_bases", "trim_quals")}
    assert _acgt(b.read(4)) == REC3["r1"] and _acgt(b.read(5)) == REC3["r2"]
    assert cols["qual1"][2].decode() == REC3["q1"] and cols["qual2"][2].decode() == REC3["q2"]
    assert (cols["bc"][2].decode(), cols["bcqual"][2].decode(), cols["si"][2].decode(), cols["siqual"][2].decode(), cols["name"][2].decode()) == \
        (REC3["bc"], REC3["bcq"], REC3["si"], REC3["siq"], REC3["name"])
    assert len(cols["trim_bases"][2]) == 2 and len(cols["trim_quals"][2]) == 2
    assert cols["name"][0].decode() == SET1_FIRST_NAME and _acgt(b.read(2)) == SET1_SECOND_R1
    p2 = int(b.bc_pair_off[1])
    assert p2 == 9 and cols["name"][p2].decode() == SET2_FIRST_NAME and _acgt(b.read(2 * (p2 + 1))) == SET2_SECOND_R1
    assert not b.bc_do_rfa.any()
    check_against_oracle(hostlib, FQ1, 2)
    check_against_oracle(hostlib, FQ1, 7, max_pairs=40)


def test_truncated_gz_is_an_error_not_a_short_run(hostlib, tmp_path):
    """gunzip exits non-zero on a truncated member: the reader must not report a clean end of input (a silently short BAM)"""
    raw = open(FQ1, "rb").read()
    p = tmp_path / "cut.fastq.gz"
    p.write_bytes(raw[: len(raw) * 2 // 3])
    with pytest.raises(capi.LhError) as e:
        for _ in hostlib.ingest(str(p), trim=2, max_pairs=50):
            pass
    assert e.value.code == capi.LH_E_IO


def test_quality_line_shorter_than_trim_is_a_read_error(hostlib, tmp_path):
    """reader.go:135-139 slices read 1's quality with the trim count and panics when it is shorter; here: a read error, on both sides"""
    rec = ["@n1 1:N:0:", "ACGTACGTACGT", "III", "ACGTACGTAC", "IIIIIIIIII", "AAACAGAGAAAGAT-1", "IIIIIIIIIIIIIIII", "ACGTACGT", "IIIIIIII"]
    p = tmp_path / "bad.fastq"
    p.write_text("\n".join(rec) + "\n")
    assert fastq_oracle.read_all(str(p), 7, 30000, 200) == []
    with pytest.raises(capi.LhError):
        list(hostlib.ingest(str(p), trim=7))
