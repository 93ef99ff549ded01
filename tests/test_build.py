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
