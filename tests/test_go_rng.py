"""Go's math/rand value stream (lariat.go:1486,1499,1510 draw tagBestAlignments' jitter from it) — known answers.

The 607-entry seed table is derived by tools/gen_go_rng_cooked.py from the table's published definition; what pins it to the
real Go runtime are outputs of Go programs that every Go user has seen: an unseeded (Seed(1)) program prints
rand.Int() = 5577006791947779410, 8674665223082153551, 6129484611666145821, ...; rand.Intn(100) = 81, 87, 47, 59, 81, 18, 25, 40, 56, 0;
and the third draw as Float64 is 0.6645600532184904 (gobyexample.com "random numbers").
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_go_rng_cooked as gen
import oracle_py

SEED1_INT63 = gen.KNOWN_SEED1_INT63
SEED1_INTN100 = [81, 87, 47, 59, 81, 18, 25, 40, 56, 0]


def _committed_table():
    txt = open(os.path.join(ROOT, "lariat_amd", "csrc", "go_rng_cooked.inc")).read()
    vals = [int(t.rstrip("ul,"), 16) for t in txt.split() if t.startswith("0x")]
    assert len(vals) == 607
    return vals


def _oracle_stream(seed, n):
    o = oracle_py.load()
    i63 = np.zeros(n, dtype=np.int64)
    f64 = np.zeros(n, dtype=np.float64)
    o.L.lo_go_rand_stream.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    o.L.lo_go_rand_stream(seed, n, i63.ctypes.data, f64.ctypes.data)
    return i63, f64


def test_committed_tables_are_what_the_generator_derives():
    assert _committed_table() == gen.cooked()
    assert open(os.path.join(ROOT, "oracle", "go_rng_cooked.inc")).read() == open(os.path.join(ROOT, "lariat_amd", "csrc", "go_rng_cooked.inc")).read()
    assert _committed_table()[0] == (-4181792142133755926) % (1 << 64)   # rng.go's first literal


def test_seed1_known_answers():
    g = gen.GoRand(_committed_table(), 1)
    got = [g.int63() for _ in range(10)]
    assert got == SEED1_INT63
    # Intn(100) -> Int31n: Int63() >> 32, rejection above max, then % n (no rejection occurs in these ten draws)
    assert [(v >> 32) % 100 for v in got] == SEED1_INTN100
    assert got[2] / float(1 << 63) == 0.6645600532184904


def test_oracle_stream_is_gos():
    i63, f64 = _oracle_stream(1, 2000)
    assert list(i63[:10]) == SEED1_INT63
    assert f64[2] == 0.6645600532184904
    for seed in (1, 0, -1, 42, -(1 << 63), (1 << 63) - 1, 2147483647, -2147483648, 0x1234567890abcdef - (1 << 64) if 0 else 0x1234567890abcdef):
        g = gen.GoRand(_committed_table(), seed)
        want = [g.int63() for _ in range(1500)]   # > 607 + 273: every branch of the ring
        got, f = _oracle_stream(seed, 1500)
        assert list(got) == want
        assert np.array_equal(f, np.array(want, dtype=np.float64) / float(1 << 63))


def _check_device_stream(lib):
    for seed in (1, 0, -7, 0x1234567890abcdef, -(1 << 63)):
        n = 1500
        fast, ring, f = lib.diag_go_rand(seed, n)
        want, wf = _oracle_stream(seed, n)
        wu = want.astype(np.uint64)
        assert np.array_equal(ring & np.uint64((1 << 63) - 1), wu)
        assert np.array_equal(fast[:273] & np.uint64((1 << 63) - 1), wu[:273])
        assert np.array_equal(f, wf)
    assert [int(v) & ((1 << 63) - 1) for v in lib.diag_go_rand(1, 10)[1]] == SEED1_INT63


def test_emulated_kernel_stream_is_gos():
    """k_rfa.h's generator (state-free path and state ring), compiled against the CPU emulator"""
    import subprocess
    import helpers
    from lariat_amd import capi
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    _check_device_stream(capi.Library(os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")))


import pytest


@pytest.mark.gpu
def test_gpu_kernel_stream_is_gos():
    from lariat_amd import capi
    _check_device_stream(capi.load_library())
