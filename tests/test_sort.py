"""Go 1.9's sort.Sort (go/src/inference/lariat.go:1546 ByPosition, split.go:108): its order of EQUAL keys is part of lariat's result,
so K8 restates the algorithm — serially (lh_sort.h dev_gosort) and, for large barcodes, with the ranges of its quickSort spread over
the lanes of the wave (wave_gosort).  Both against the oracle's restatement (oracle/gosort_impl.h), on the kernel sources under the
CPU emulator here and on the GPU in test_gpu_front.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
from lariat_amd import capi

EMU = os.environ.get("LH_EMU_LIB") or os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")   # (LH_EMU_LIB: e.g. an AddressSanitizer build, tests/hipemu/Makefile asan)


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    return capi.Library(EMU)


def oracle_perm(oracle, first, keys):
    first = np.ascontiguousarray(first, dtype=np.int32)
    k = np.ascontiguousarray(keys, dtype=np.int64).copy()
    perm = np.concatenate([np.arange(first[i + 1] - first[i], dtype=np.int32) for i in range(len(first) - 1)])
    oracle.L.lo_gosort.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    oracle.L.lo_gosort.restype = None
    oracle.L.lo_gosort(len(first) - 1, first.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int64)), perm.ctypes.data_as(C.POINTER(C.c_int32)))
    return perm, k


def cases():
    rng = np.random.default_rng(12)
    out = []
    for sizes, hi in (([0, 1, 2, 5, 12, 13, 40, 41, 100], 6), ([700, 3, 1500], 40), ([2500], 3), ([300] * 70, 10), ([5000], 1 << 40)):
        first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        keys = rng.integers(0, hi, size=int(first[-1]))   # few distinct values: ties everywhere
        out.append((first, keys))
    n = 3000   # runs already sorted, reversed, and a sawtooth: the patterns that drive quickSort towards its depth limit
    out.append((np.array([0, n, 2 * n, 3 * n], dtype=np.int32), np.concatenate([np.arange(n), np.arange(n)[::-1], np.arange(n) % 7])))
    # (r06) ranges longer than LH_GOSORT_WAVE_MIN (96) are partitioned by the whole wave (wave_go_pivot): sizes around that limit and around multiples of 64, keys all
    # equal, two values, mostly distinct with a few equal pairs (a contig's positions in K8: the case the product meets), organ pipe
    sizes = [95, 96, 97, 98, 127, 128, 129, 160, 191, 192, 193, 257, 1153, 2304]
    first = np.concatenate([[0], np.cumsum(sizes * 4)]).astype(np.int32)
    parts = []
    for rep in range(4):
        for sz in sizes:
            if rep == 0: k = np.zeros(sz, dtype=np.int64)
            elif rep == 1: k = rng.integers(0, 2, size=sz)
            elif rep == 2:
                k = rng.permutation(sz).astype(np.int64) * 7
                for _ in range(1 + sz // 100): k[int(rng.integers(0, sz))] = k[int(rng.integers(0, sz))]
            else: k = np.minimum(np.arange(sz), sz - np.arange(sz)) // 3
            parts.append(k)
    out.append((first, np.concatenate(parts)))
    return out


def check(lib, oracle):
    for first, keys in cases():
        want, sorted_keys = oracle_perm(oracle, first, keys)
        ps, pw = lib.diag_gosort(first, keys)
        assert np.array_equal(ps, want) and np.array_equal(pw, want)
        for k in range(len(first) - 1):   # and it IS a sort
            a, b = int(first[k]), int(first[k + 1])
            assert np.array_equal(np.asarray(keys[a:b])[want[a:b]], sorted_keys[a:b]) and np.all(np.diff(sorted_keys[a:b]) >= 0)


def check_split(lib, oracle):
    """a long list as K8 sorts it when it does not fit LDS (k_rfa.h): wave_gosort_split down to ranges of `limit`, those with the depth left (lh_sort.h)"""
    rng = np.random.default_rng(44)
    for n, hi, limit in ((3000, 1 << 40, 1024), (3000, 1 << 40, 64), (5000, 40, 1024), (5000, 3, 64), (2500, 1, 64), (4097, 900, 256), (700, 50, 64)):
        keys = rng.integers(0, hi, size=n)
        if hi > 1 << 30:   # mostly distinct, a few equal pairs: a contig's positions
            for _ in range(30): keys[int(rng.integers(0, n))] = keys[int(rng.integers(0, n))]
        want, _ = oracle_perm(oracle, np.array([0, n], dtype=np.int32), keys)
        assert np.array_equal(lib.diag_gosort_split(keys, limit), want), (n, hi, limit)
    n = 3000   # sorted, reversed, sawtooth: towards the depth limit (a range whose depth is used up before it is short stays whole: heap sort)
    for keys in (np.arange(n), np.arange(n)[::-1], np.arange(n) % 7, np.minimum(np.arange(n), n - np.arange(n)) // 3):
        want, _ = oracle_perm(oracle, np.array([0, n], dtype=np.int32), keys)
        assert np.array_equal(lib.diag_gosort_split(keys, 64), want)


def test_emu_gosort_split(emu, oracle):
    check_split(emu, oracle)


def test_emu_gosort_serial_and_wave(emu, oracle):
    check(emu, oracle)


def check_bitonic(lib):
    """K8's network for lists without equal keys (lh_sort.h wave_bitonic_u64, wave_bitonic_u64_blocks): any length, in one LDS block, block by block, all in memory"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 200, 1000, 1023, 1024, 1025, 1153, 2047, 2048, 2049, 3000, 4097, 9000):
        keys = rng.permutation(n * 3)[:n].astype(np.uint64) << np.uint64(20) | np.arange(n, dtype=np.uint64)
        want = np.sort(keys)
        for block in (64, 1024, 0):
            assert np.array_equal(lib.diag_bitonic(keys, block), want), (n, block)


def test_emu_bitonic_network(emu):
    check_bitonic(emu)


# ---- klib's ks_introsort (the region sorts of mem_sort_dedup_patch, reached from gobwa.go:244,253,291,315): K5 / K6 run it by one lane (dev_introsort) and, when a rescue
# replay has to run a call as written, by the whole wave (wave_introsort_i64: the partitions from ballots, the closing insertion sort as a stable ranking).  Equal keys must
# end up where klib leaves them: both against the oracle's restatement (oracle/ksort_impl.h).
def oracle_introsort(oracle, first, keys):
    first = np.ascontiguousarray(first, dtype=np.int32)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    perm = np.zeros(int(first[-1]), dtype=np.int32)
    oracle.L.lo_ks_introsort.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    oracle.L.lo_ks_introsort.restype = None
    oracle.L.lo_ks_introsort(len(first) - 1, first.ctypes.data_as(C.POINTER(C.c_int32)), k.ctypes.data_as(C.POINTER(C.c_int64)), perm.ctypes.data_as(C.POINTER(C.c_int32)))
    return perm


def introsort_cases():
    rng = np.random.default_rng(21)
    out = []
    for sizes, hi in (([0, 1, 2, 3, 5, 16, 17, 18, 33, 34, 40, 64, 65, 100, 127, 128, 129], 4), ([200] * 40, 8), ([300] * 30, 1 << 40), ([1024, 1000, 513], 3),
                      (list(rng.integers(17, 400, size=120)), 30), (list(rng.integers(17, 400, size=120)), 2), ([1024] * 4, 1)):
        first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        out.append((first, rng.integers(0, hi, size=int(first[-1]))))
    # end positions of a region list: mostly distinct, a few equal pairs (the case the pipeline meets)
    sizes = list(rng.integers(30, 320, size=150))
    first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    keys = rng.integers(0, 1 << 33, size=int(first[-1]))
    for k in range(len(sizes)):
        a, b = int(first[k]), int(first[k + 1])
        for _ in range(int(rng.integers(1, 4))):
            keys[a + int(rng.integers(0, b - a))] = keys[a + int(rng.integers(0, b - a))]
    out.append((first, keys))
    n = 1000   # sorted, reversed, sawtooth, organ pipe: the patterns that drive the partitions towards the depth limit
    out.append((np.array([0, n, 2 * n, 3 * n, 4 * n], dtype=np.int32), np.concatenate([np.arange(n), np.arange(n)[::-1], np.arange(n) % 7, np.minimum(np.arange(n), n - np.arange(n))])))
    return out


def check_introsort(lib, oracle):
    for first, keys in introsort_cases():
        want = oracle_introsort(oracle, first, keys)
        ps, pw = lib.diag_introsort(first, keys)
        assert np.array_equal(ps, want), ("one-lane introsort", np.nonzero(ps != want)[0][:8])
        assert np.array_equal(pw, want), ("wave introsort", np.nonzero(pw != want)[0][:8])
        for k in range(len(first) - 1):   # and it IS a sort
            a, b = int(first[k]), int(first[k + 1])
            assert np.all(np.diff(np.asarray(keys[a:b])[want[a:b]]) >= 0)


def test_emu_introsort_one_lane_and_wave(emu, oracle):
    check_introsort(emu, oracle)
