"""A slice of the differential fuzzer inside the suite (the fuzzer's evidence lived in profiles/*.log only): 100 random
cases of tests/checkers/fuzz_gpu.py — random genomes with duplications and repeat families, index built on the device or uploaded, table
depths, super-block sizes, SA sampling, ALT contigs, read lengths 50..240, substitution / indel / junk rates, ambiguous bases, per-barcode
RFA switches, scoring options, launch flags, one-call and split (resident slot + two-step download) boundary — HIP path against the oracle,
stage dumps and every result field.  The long runs (thousands of cases) are in profiles/r03_fuzz_gpu.log."""
import importlib.util
import os

import pytest

import helpers
from lariat_amd import capi

pytestmark = pytest.mark.gpu


def test_differential_fuzz_slice(oracle):
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(helpers.ROOT, "tests", "checkers", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    lib = capi.load_library()
    for seed in range(52000, 52100):
        fz.run_case(lib, oracle, seed)
