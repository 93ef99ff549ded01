"""N1 (first half) — BAM record content (lariat_amd/csrc/records.cpp) against the restatement of bamwriter.go's AppendBam in
oracle/bam_oracle.py.  The alignments come from the CPU oracle, so this runs without a GPU; test_gpu_inference.py checks the
same text on the HIP result."""
import os
import sys

import numpy as np
import pytest

import helpers
from lariat_amd import capi, synth

sys.path.insert(0, os.path.join(helpers.ROOT, "oracle"))
import bam_oracle  # noqa: E402

COLS = ("name", "rgid", "qual1", "qual2", "trim_bases", "trim_quals", "bc", "rawbc", "bcqual", "si", "siqual")


@pytest.fixture(scope="module")
def hostlib():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    return capi.Library(os.environ.get("LH_HOST_LIB") or ge.LIB)   # (LH_HOST_LIB: the host code in another build, e.g. the emulator's AddressSanitizer one)


def write_fastq(tmp_path, rs, name="r.fastq"):
    p = tmp_path / name
    p.write_text(synth.to_fastq9(rs, trim_prefix=7))
    return str(p)


def oracle_text(res, b, contig_names):
    cols = {c: b.column(c) for c in COLS}
    return bam_oracle.records_text(res, cols, b.seq, b.seq_off, b.bc_pair_off, b.set_complete, contig_names)


def test_record_content(hostlib, oracle, tmp_path):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    # noisy reads, junk pairs (unmapped / improper records), a barcode below the RFA threshold, split candidates
    rs = synth.make_reads(contigs, names, n_barcodes=5, pairs_per_barcode=40, seed=61, sub_lo=0.002, sub_hi=0.04, indel_rate=0.003, junk_frac=0.08)
    for p in range(3, rs.n_pairs, 9):   # chimeric reads: the second half comes from another read's locus -> split (supplementary) records
        a, d = 2 * p, 2 * ((p * 7 + 11) % rs.n_pairs)
        la, ld = int(rs.seq_off[a + 1] - rs.seq_off[a]), int(rs.seq_off[d + 1] - rs.seq_off[d])
        h = min(la, ld) // 2
        rs.seq[rs.seq_off[a] + la - h:rs.seq_off[a] + la] = rs.seq[rs.seq_off[d] + ld - h:rs.seq_off[d] + ld]
    path = write_fastq(tmp_path, rs)
    n_rec = n_split = n_unmapped = 0
    all_text = ""
    for b in hostlib.ingest(path, trim=7, max_pairs=90):
        res = oidx.align_barcodes(b, threads=4)
        got = hostlib.records_text(res, b, names)
        want = oracle_text(res, b, names)
        assert got == want
        all_text += got
        lines = got.splitlines()
        assert len(lines) >= 2 * b.n_pairs
        for ln in lines:
            f = ln.split("\t")
            flag = int(f[1])
            assert flag & 1 and bool(flag & 0x40) != bool(flag & 0x80)
            n_split += bool(flag & 256)
            n_unmapped += bool(flag & 4)
            assert (f[2] == "*") == bool(flag & 4)
            assert [t[:2] for t in f[11:13]] == ["RX", "QX"]
        n_rec += len(lines)
    assert n_rec > 0 and n_unmapped > 0 and n_split > 0
    assert "SA:Z:" in all_text and "H" in "".join(ln.split("\t")[5] for ln in all_text.splitlines() if int(ln.split("\t")[1]) & 256)
    # the tags lariat's downstream tools need
    assert "BX:Z:" in all_text and "AS:i:" in all_text and "XS:i:" in all_text and "AM:Z:" in all_text


def test_debug_tags(hostlib, oracle, tmp_path):
    """-debugBamTags (bamwriter.go:498-558): records.cpp derives MapQData from the result's per-candidate fields; the expected text
    prints the MapQData the oracle's molecule structures hold (lariat.go:687-719, 917-958; split.go:154)"""
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=6, pairs_per_barcode=40, seed=67, sub_lo=0.002, sub_hi=0.04, indel_rate=0.003, junk_frac=0.08)
    rs.bc_pair_off = np.array([0, 3] + list(rs.bc_pair_off[2:]), dtype=rs.bc_pair_off.dtype)   # a barcode below the RFA threshold (candidate_molecules == nil)
    for p in range(3, rs.n_pairs, 9):   # chimeric reads -> split records (their MapQData holds the two scores only)
        a, d = 2 * p, 2 * ((p * 7 + 11) % rs.n_pairs)
        la, ld = int(rs.seq_off[a + 1] - rs.seq_off[a]), int(rs.seq_off[d + 1] - rs.seq_off[d])
        h = min(la, ld) // 2
        rs.seq[rs.seq_off[a] + la - h:rs.seq_off[a] + la] = rs.seq[rs.seq_off[d] + ld - h:rs.seq_off[d] + ld]
    path = write_fastq(tmp_path, rs)
    seen = set()
    n_sb = n_rd = n_norfa = 0
    for b in hostlib.ingest(path, trim=7, max_pairs=120):
        res = oidx.align_barcodes(b, threads=4)
        got = hostlib.records_text(res, b, names, debug_tags=True)
        cols = {c: b.column(c) for c in COLS}
        want = bam_oracle.records_text(res, cols, b.seq, b.seq_off, b.bc_pair_off, b.set_complete, names, debug_tags=True, md_int=res.md_int, md_sb_conf=res.md_sb_conf)
        assert got == want
        # and the derivation restated in Python (what the GPU test uses, where no oracle molecules exist)
        assert got == bam_oracle.records_text(res, cols, b.seq, b.seq_off, b.bc_pair_off, b.set_complete, names, debug_tags=True, bc_do_rfa=b.bc_do_rfa)
        plain = hostlib.records_text(res, b, names).splitlines()
        for ln, pl in zip(got.splitlines(), plain):
            f = ln.split("\t")
            tags = [t[:2] for t in f[11:]]
            seen.update(tags)
            n_sb += "XX" in tags
            rd = [t for t in f[11:] if t.startswith("RD:Z:")]
            n_rd += int(rd[0][5:]) > 0
            n_norfa += "CS:Z:0" in f and "CM:Z:0" in f and "CP:Z:0" not in f and not int(f[1]) & 256
            base = [t for k, t in enumerate(f) if k < 11 or t[:2] in ("RX", "QX", "TR", "TQ", "BC", "QT", "RG", "XS", "AS", "AM", "XT", "SA", "BX", "DM")]
            assert base == [t for k, t in enumerate(pl.split("\t")) if k < 11 or t[:2] not in ("XC", "AC", "XM")]   # everything else is unchanged
    assert {"AA", "CP", "CM", "CU", "CS", "RD", "MS", "MC", "PP", "PS", "PL", "AC", "PC", "XX", "XL", "XP", "XR", "XC"} <= seen
    assert n_sb > 0 and n_rd > 0 and n_norfa > 0


def test_pair_flags_are_consistent(hostlib, oracle, tmp_path):
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = helpers.small_reads(names, contigs, n_barcodes=3, pairs=50, junk=0.05, seed=63)
    path = write_fastq(tmp_path, rs)
    for b in hostlib.ingest(path, trim=7):
        res = oidx.align_barcodes(b, threads=4)
        prim = {}
        for ln in hostlib.records_text(res, b, names).splitlines():
            f = ln.split("\t")
            if not int(f[1]) & 256:
                prim.setdefault(f[0], []).append(f)
        for name, recs in prim.items():
            assert len(recs) == 2
            a, c = recs
            fa, fc = int(a[1]), int(c[1])
            assert bool(fa & 0x2) == bool(fc & 0x2)                      # proper pair on both or neither
            if not fa & 0x8:
                assert a[6] == c[2] and int(a[7]) == int(c[3])          # RNEXT/PNEXT point at the mate's record
                assert bool(fa & 0x20) == bool(fc & 0x10)
            if int(a[8]) != 0 or int(c[8]) != 0:
                assert int(a[8]) == -int(c[8]) or 0 in (int(a[8]), int(c[8]))


def test_bam_files_round_trip(hostlib, oracle, tmp_path):
    """the container: every record line comes back from bc_sorted_bam.bam, each record sits in exactly one position bucket
    chosen as AppendBams does, file names and short-contig packing follow CreateBAMs, BGZF blocks are well formed"""
    import bam_reader
    names, contigs = helpers.small_genome()   # 300 kb, 200 kb, 100 kb
    lens = [len(c) for c in contigs]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=4, pairs_per_barcode=60, seed=67, sub_hi=0.03, indel_rate=0.002, junk_frac=0.06)
    path = write_fastq(tmp_path, rs)
    outdir = tmp_path / "out"
    outdir.mkdir()
    chunk = 150000   # chrA (300 kb) and chrB (200 kb) split into two files each, chrC (100 kb) gets one
    w = hostlib.bam_writer(str(outdir), names, lens, read_groups="s:lib:1:fc:1,bad", position_chunk_size=chunk, first_chunk=True, command_line="lariat_amd test", threads=3)
    want = []
    for b in hostlib.ingest(path, trim=7, max_pairs=100):
        res = oidx.align_barcodes(b, threads=4)
        want += hostlib.records_text(res, b, names).splitlines()
        w.append(res, b)
    w.close()
    dbg = tmp_path / "dbg"
    dbg.mkdir()
    w2 = hostlib.bam_writer(str(dbg), names, lens, position_chunk_size=chunk)   # CreateBAMs(..., debugTags = true, ...)
    w2.set_debug_tags(True)
    want_dbg = []
    for b in hostlib.ingest(path, trim=7, max_pairs=100):
        res = oidx.align_barcodes(b, threads=4)
        want_dbg += hostlib.records_text(res, b, names, debug_tags=True).splitlines()
        w2.append(res, b)
    w2.close()
    assert [ln for ln, _ in bam_reader.read_bam(str(dbg / "bc_sorted_bam.bam"))[2]] == want_dbg and "\tCP:Z:" in want_dbg[0]
    files = sorted(os.listdir(outdir))
    assert files == ["000000-chrA_0000000000_pos_bucketed.bam", "000000-chrA_0000150000_pos_bucketed.bam", "000001-chrB_0000000000_pos_bucketed.bam",
                     "000001-chrB_0000150000_pos_bucketed.bam", "000002-chrC_0000000000_pos_bucketed.bam", "ZZZ_unmapped_pos_bucketed.bam", "bc_sorted_bam.bam"]
    text, refs, lines = bam_reader.read_bam(str(outdir / "bc_sorted_bam.bam"))
    assert refs == list(zip(names, lens))
    assert [ln for ln, _ in lines] == want
    assert text.count("@SQ") == 3 and "@RG\tID:s:lib:1:fc:1\tLB:lib.1\tPL:ILLUMINA\tPU:s:lib:1:fc:1\tSM:s" in text and "ID:bad" not in text
    assert "@PG\tID:lariat\tPN:longranger.lariat\tCL:lariat_amd test" in text and text.count("@CO") == 3
    seen = []
    for f in files[:-1]:
        t, _, ls = bam_reader.read_bam(str(outdir / f))
        for ln, bin_ in ls:
            fld = ln.split("\t")
            if f.startswith("ZZZ"):
                assert int(fld[3]) == -1 and bin_ == 4680
            else:
                contig, off = f.split("-", 1)[1].rsplit("_", 3)[0], int(f.split("_")[-3])
                assert fld[2] == contig and off <= int(fld[3]) < off + chunk
            seen.append(ln)
        assert (t.count("@CO") == 3) == (f.startswith("000000-chrA_0000000000") or f.startswith("ZZZ"))
    assert sorted(seen) == sorted(want)


def test_bam_short_contig_packing(hostlib, tmp_path):
    """CreateBAMs packs consecutive short contigs into one file until position_chunk_size would be exceeded"""
    d = tmp_path / "o"
    d.mkdir()
    w = hostlib.bam_writer(str(d), ["a", "b", "c", "d", "e"], [40, 50, 30, 250, 20], position_chunk_size=100, first_chunk=False)
    w.close()
    assert sorted(os.listdir(d)) == ["000000-a_0000000000_pos_bucketed.bam", "000002-c_0000000000_pos_bucketed.bam", "000003-d_0000000000_pos_bucketed.bam",
                                     "000003-d_0000000100_pos_bucketed.bam", "000003-d_0000000200_pos_bucketed.bam", "ZZZ_unmapped_pos_bucketed.bam", "bc_sorted_bam.bam"]


def test_bam_shard_concat_equals_single_process(hostlib, oracle, tmp_path):
    """the multi-GPU output path: two ranks align contiguous barcode ranges and write their own file sets; lh_bam_concat of the
    shards in rank order gives bc_sorted_bam.bam record-for-record as one process writes it, and every position bucket holds
    the same records (bamwriter.go:139-191 file set)"""
    import bam_reader
    from lariat_amd import shard
    names, contigs = helpers.small_genome()
    lens = [len(c) for c in contigs]
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=6, pairs_per_barcode=40, seed=71, sub_hi=0.02, junk_frac=0.05)
    path = write_fastq(tmp_path, rs)
    batches = list(hostlib.ingest(path, trim=7, max_pairs=40))   # one barcode per batch
    assert len(batches) == 6
    results = [oidx.align_barcodes(b, threads=4) for b in batches]

    def write(dirname, which, first):
        d = tmp_path / dirname
        d.mkdir()
        w = hostlib.bam_writer(str(d), names, lens, read_groups="s:lib:1:fc:1", position_chunk_size=150000, first_chunk=first, command_line="x", threads=2)
        for k in which:
            w.append(results[k], batches[k])
        w.close()
        return str(d)

    single = write("single", range(6), True)
    sizes = np.concatenate([[0], np.cumsum([b.n_pairs for b in batches])])
    ranges = shard.barcode_ranges(sizes, 2)
    shards = [write("rank%d" % r, range(b0, b1), r == 0) for r, (b0, b1) in enumerate(ranges)]
    out = tmp_path / "joined"
    out.mkdir()
    hostlib.bam_concat(shards, str(out))
    assert sorted(os.listdir(out)) == sorted(os.listdir(single))
    for f in sorted(os.listdir(single)):
        t1, r1, l1 = bam_reader.read_bam(os.path.join(single, f))
        t2, r2, l2 = bam_reader.read_bam(str(out / f))
        assert t1 == t2 and r1 == r2
        if f == "bc_sorted_bam.bam":
            assert l1 == l2 and len(l1) > 400
        else:
            assert sorted(l1) == sorted(l2)


def test_bam_record_limits_are_errors(hostlib, oracle, tmp_path):
    """the BAM format holds read names of at most 254 bytes (l_read_name is one byte, NUL included): a longer one is LH_E_LIMIT from
    lh_bam_append with nothing appended — not a truncated name or a corrupt record"""
    import bam_reader
    names, contigs = helpers.small_genome()
    oidx = oracle.index_build_naive(names, contigs)
    rs = synth.make_reads(contigs, names, n_barcodes=1, pairs_per_barcode=6, seed=71)
    text = synth.to_fastq9(rs, trim_prefix=7).split("\n")
    text[9] = "@" + "n" * 300 + " rest"   # the second record's header (9 lines per record)
    p = tmp_path / "long.fastq"
    p.write_text("\n".join(text))
    d = tmp_path / "o"
    d.mkdir()
    w = hostlib.bam_writer(str(d), names, [len(c) for c in contigs])
    for b in hostlib.ingest(str(p), trim=7, max_pairs=100):
        res = oidx.align_barcodes(b, threads=2)
        with pytest.raises(capi.LhError) as e:
            w.append(res, b)
        assert e.value.code == capi.LH_E_LIMIT and "254" in str(e.value)
    w.close()
    assert bam_reader.read_bam(str(d / "bc_sorted_bam.bam"))[2] == []   # nothing was appended
