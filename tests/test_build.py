"""CPU-side checks of the product: the index builder reproduces the reference's fixture bytes, and the C-ABI library
loads and exports every symbol include/lariat_hip.h declares (no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest

import helpers
from lariat_amd import capi


def test_abi_symbols_exported():
    import __graft_entry__ as ge
    lib_path = ge.build()
    lib = capi.Library(lib_path)
    hdr = open(os.path.join(helpers.ROOT, "include", "lariat_hip.h")).read()
    declared = set(re.findall(r"\b(lh_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib.L, sym), sym


def test_no_device_fails_loudly():
    """without a GPU the product must refuse to work (no CPU fallback)"""
    import __graft_entry__ as ge
    lib = capi.Library(ge.build())
    if lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.LhError) as e:
        lib.index_load(helpers.PHIX)
    assert e.value.code in (3, 5)


def test_index_build_reproduces_phix_fixture(tmp_path):
    """lh_index_build vs go/src/test/inputs/phix/PhiX.fa.{bwt,sa,pac,ann,amb} (byte-exact)"""
    import __graft_entry__ as ge
    lib = capi.Library(ge.build())
    names, seqs = helpers.read_fasta(helpers.PHIX)
    prefix = str(tmp_path / "PhiX.fa")
    lib.index_build(prefix, names, [capi.sequence_convert(s) for s in seqs], threads=2)
    for ext in ["bwt", "sa", "pac", "ann", "amb"]:
        assert open(prefix + "." + ext, "rb").read() == open(helpers.PHIX + "." + ext, "rb").read(), ext


def test_index_build_matches_oracle_on_multicontig(tmp_path, oracle):
    import __graft_entry__ as ge
    lib = capi.Library(ge.build())
    names, contigs = helpers.small_genome(seed=3)
    prefix = str(tmp_path / "g.fa")
    lib.index_build(prefix, names, contigs, threads=4)
    oidx = oracle.index_build_naive(names, contigs)
    for w, ext in enumerate(["bwt", "sa", "pac", "ann", "amb"]):
        assert open(prefix + "." + ext, "rb").read() == oidx.image(w), ext


def test_reference_with_ambiguous_bases(tmp_path):
    """references with N runs (every real Long Ranger reference has them; gobwa.go:130 loads .amb): `bwa index` replaces an
    ambiguous base by lrand48() & 3 under srand48(11) and records the holes in .amb / .ann (bntseq.c bns_fasta2bntseq).  The
    generator is pinned on the C library's own lrand48; product and oracle builders write the same five files; an index with
    holes loads, saves back byte-identically, and aligns like the oracle."""
    import ctypes
    import numpy as np
    import oracle_py
    import __graft_entry__ as ge
    libc = ctypes.CDLL(None)
    libc.lrand48.restype = ctypes.c_long
    libc.srand48(11)
    want_draws = [libc.lrand48() & 3 for _ in range(40)]
    rng = np.random.default_rng(4)
    a = "".join("ACGT"[v] for v in rng.integers(0, 4, size=3000))
    b = "".join("ACGT"[v] for v in rng.integers(0, 4, size=2500))
    #            a run of N, a run of another ambiguity letter right behind it (two holes), a lower-case stretch, a hole at a contig's start
    c1 = a[:700] + "N" * 23 + "RRR" + a[726:1500].lower() + "N" + a[1501:]
    c2 = "NNNNN" + b[5:2000] + "YN" + b[2002:]
    lib = capi.Library(ge.LIB)
    contigs = [c1.encode(), c2.encode()]
    pac, l_pac, n_ambs, holes = lib.reference_pack(contigs)
    assert holes == [(700, 23, "N"), (723, 3, "R"), (1500, 1, "N"), (3000, 5, "N"), (5000, 1, "Y"), (5001, 1, "N")]
    assert list(n_ambs) == [3, 3] and l_pac == 5500
    amb_pos = [p for o, l, _ in holes for p in range(o, o + l)]
    got_draws = [int(pac[p >> 2] >> ((~p & 3) << 1) & 3) for p in amb_pos]
    assert got_draws == want_draws[: len(amb_pos)]
    prefix = str(tmp_path / "ref.fa")
    lib.index_build(prefix, ["c1", "c2"], [np.frombuffer(c, dtype=np.uint8) for c in contigs], threads=2)
    o = oracle_py.load()
    oidx = o.index_build_naive(["c1", "c2"], [np.frombuffer(c, dtype=np.uint8) for c in contigs])
    for k, ext in enumerate((".bwt", ".sa", ".pac", ".ann", ".amb")):
        assert open(prefix + ext, "rb").read() == oidx.image(k), ext
    assert open(prefix + ".amb").read().splitlines()[:3] == ["5500 2 6", "700 23 N", "723 3 R"]
    assert open(prefix + ".ann").read().splitlines()[2] == "0 3000 3"
    lo = o.index_load(prefix)
    assert lo.image(4) == oidx.image(4)
