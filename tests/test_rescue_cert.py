"""K6's certificate (csrc/k_rescue3.h: k_resc_cert) against the oracle's ksw_align2 (mem_matesw's call, gobwa.go:286-325 -> bwamem_pair.c), on caller-supplied
windows through lh_diag_rescue_sw — the launches of the pipeline's rescue_dir after its emitting pass.

  * random windows of every kind the probe saw (no hit, a diverged copy, a gapped copy, tandem copies that tie, a copy beside a shorter one): the certificate
    path and the whole-window path both give the oracle's (score, te, qe, tb, qb) wherever a region comes of the attempt, and agree that none does elsewhere;
  * crafted windows, one per term of the certificate: with the term in place the result is the oracle's; with that ONE term switched off (`weaken`) it is not —
    every bound is there because a window exists that needs it.

The emulator runs them in `-m "not gpu"`, the device in `-m gpu`."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
from lariat_amd import capi

MINSEED = 19
EMU = os.environ.get("LH_EMU_LIB") or os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    return capi.Library(EMU)


@pytest.fixture(scope="module")
def hip():
    L = capi.load_library()
    assert L.device_count() >= 1
    return L


def _oracle_sw(oracle, q, t):
    oracle.L.lo_ksw_align2.argtypes = [C.c_int32, C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]
    out = (C.c_int32 * 7)()
    q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
    oracle.L.lo_ksw_align2(len(q), q.ctypes.data_as(C.POINTER(C.c_uint8)), len(t), t.ctypes.data_as(C.POINTER(C.c_uint8)), out)
    return tuple(int(out[i]) for i in range(5))


def _check(lib, oracle, queries, windows, full=False, weaken=0):
    """the device's answers against the oracle's; returns (number of cases that differ, the device's array)"""
    got = lib.diag_rescue_sw(queries, windows, full=full, weaken=weaken)
    bad = 0
    for c, (q, t) in enumerate(zip(queries, windows)):
        want = _oracle_sw(oracle, q, t)
        hit_w = want[0] >= MINSEED and want[4] >= 0
        hit_g = got[c, 0] >= MINSEED and got[c, 4] >= 0
        if hit_w != hit_g or (hit_w and tuple(int(x) for x in got[c, :5]) != want):
            bad += 1
        elif hit_w:   # rows2 is what the reference's reverse pass ran (rescue_cells is counted from it)
            assert got[c, 7] == want[1] - want[3] + 1, (c, got[c], want)
    return bad, got


def _random_cases(rng, n):
    def mutate(x, sub, n_indel):
        y = x.copy()
        m = rng.random(len(y)) < sub
        y[m] = (y[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        for _ in range(n_indel):
            at = int(rng.integers(10, len(y) - 10))
            ln = int(rng.integers(1, 9))
            y = np.concatenate([y[:at], y[at + ln:]]) if rng.random() < 0.5 else np.concatenate([y[:at], rng.integers(0, 4, size=ln).astype(np.uint8), y[at:]])
        return y
    Q, T = [], []
    for case in range(n):
        qlen = int(rng.integers(40, 151))
        q = rng.integers(0, 4, size=qlen).astype(np.uint8)
        tlen = int(rng.integers(300, 800))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        kind = case % 8
        if kind <= 2:      # one diverged copy, ungapped or with indels
            c = mutate(q, [0.01, 0.05, 0.12][kind], [0, 1, 2][kind])
            at = int(rng.integers(0, tlen - len(c)))
            t[at:at + len(c)] = c
        elif kind == 3:    # two copies in tandem, equally good: the FIRST row that reaches the score decides
            c = mutate(q, 0.03, 0)
            at = int(rng.integers(0, tlen - 2 * len(c) - 5))
            t[at:at + len(c)] = c
            t[at + len(c) + 3:at + 2 * len(c) + 3] = c
        elif kind == 4:    # a copy whose best path has a gap, beside a shorter ungapped one
            c = mutate(q, 0.02, 1)
            at = int(rng.integers(0, max(1, tlen - len(c) - 90)))
            t[at:at + len(c)] = c
            t[tlen - 80:tlen - 20] = q[:min(60, qlen)][:60] if qlen >= 60 else t[tlen - 80:tlen - 20]
        elif kind == 5:    # a copy cut off by the window's edge
            c = mutate(q, 0.02, 0)
            cut = int(rng.integers(20, len(c) - 10))
            if rng.random() < 0.5: t[:len(c) - cut] = c[cut:]
            else: t[tlen - cut:] = c[:cut]
        elif kind == 6:    # a low-complexity mate and window: the 5-mers hit everywhere
            q = np.tile(np.array([0, 1], dtype=np.uint8), qlen)[:qlen]
            t[100:100 + 200] = np.tile(np.array([0, 1], dtype=np.uint8), 100)
        # kind 7: nothing planted
        Q.append(q); T.append(t)
    return Q, T


def _crafted(rng):
    """(name, query, window, the weaken bits each of which must break it)"""
    def rnd(n):
        return rng.integers(0, 4, size=n).astype(np.uint8)

    def no_repeat(n):   # neighbouring bases differ, so that a sequence shifted by one base mismatches itself everywhere
        x = np.zeros(n, dtype=np.uint8)
        x[0] = rng.integers(0, 4)
        for i in range(1, n):
            x[i] = (x[i - 1] + rng.integers(1, 4)) & 3
        return x
    def fix_no_repeat(x, lo, hi):   # x[lo:hi]: every base differs from the one before it
        for i in range(max(lo, 1), hi):
            if x[i] == x[i - 1]:
                x[i] = (x[i - 1] + 1 + (i & 1)) & 3
    cases = []
    # 1. paths off d0: diagonal A (earlier rows: it is d0 on a tie of hits) holds runs of 30 | 3 | 30 around single mismatches — 52 hits, best segment 55; diagonal B,
    #    185 rows later, one run of 56 — 52 hits, segment 56: the result is B's.  Off: class A settles on A.
    q = rnd(70)
    a = q[:65].copy(); a[30] = (a[30] + 1) & 3; a[34] = (a[34] + 2) & 3
    t = rnd(500)
    t[50:115] = a
    t[49] = (t[49] + 0) & 3; t[115] = (q[65] + 1) & 3            # (A ends where it was planted)
    t[235 + 5:235 + 61] = q[5:61]                                # B: columns 5..60 on diagonal 235
    t[235 + 4] = (q[4] + 1) & 3; t[235 + 61] = (q[61] + 1) & 3
    cases.append(("a stronger segment on another diagonal", q, t, (1,)))
    # 2. two pieces of d0 around a bad stretch that a detour through d0 + 1 skips: 40 | 5 mismatches | 40 on d0 (60), 40 + 4 - 7 - 7 + 40 = 70 by the detour.
    q = rnd(85)
    fix_no_repeat(q, 39, 47)
    t = rnd(400)
    at = 150
    t[at:at + 40] = q[:40]
    t[at + 40] = [x for x in range(4) if x != q[40] and x != q[39]][0]
    t[at + 41:at + 45] = q[40:44]
    t[at + 45:at + 85] = q[45:85]
    t[at - 1] = (t[at - 1] + 0) & 3; t[at + 85] = (t[at + 85] + 0) & 3
    cases.append(("a detour around a bad stretch of d0", q, t, (2,)))
    # 3. the distance condition: 100 matches on d0, then 12 more on d0 + 1 (one deleted window base between): 100 + 12 - 7 = 105.
    q = rnd(112)
    fix_no_repeat(q, 99, 112)
    t = rnd(500)
    at = 200
    t[at:at + 100] = q[:100]
    t[at + 100] = [x for x in range(4) if x != q[100] and x != q[99]][0]
    t[at + 101:at + 113] = q[100:112]
    cases.append(("a short piece on the neighbouring diagonal", q, t, (4, 8)))   # (without the distance condition it is settled on d0; with it, the rows around d0 must still hold the piece)
    # 4. the strip's width: 100 matches on d0 and 50 more on d0 + 10: 100 + 50 - 16 = 134; the last rows of the second piece lie below d0 + qlen.
    q = rnd(150)
    t = rnd(600)
    at = 250
    t[at:at + 100] = q[:100]
    t[at + 110:at + 160] = q[100:150]
    cases.append(("a second piece ten diagonals off", q, t, (4, 8)))
    return cases


def _run_random(lib, oracle, n, seed):
    rng = np.random.default_rng(seed)
    Q, T = _random_cases(rng, n)
    bad, got = _check(lib, oracle, Q, T)
    assert bad == 0, bad
    settled = int((got[:, 5] < 0).sum())
    none = int(((got[:, 5] == 0) & (got[:, 6] == 0)).sum())
    tl = np.array([len(t) for t in T])
    part = int(((got[:, 6] > 0) & (got[:, 6] < tl)).sum())
    whole = int((got[:, 6] == tl).sum())
    assert settled > n // 8 and none > n // 16 and part > n // 16 and whole > n // 16, (settled, none, part, whole)   # every class of the certificate is exercised
    bad_full, got_full = _check(lib, oracle, Q, T, full=True)
    assert bad_full == 0
    assert (got_full[:, 6] == tl).all()
    return settled, none, part, whole


def _run_crafted(lib, oracle, seeds):
    """every instance: the certificate path and the whole-window path give the oracle's answer (always).  Per crafted case: the terms it was made for are needed and the
    others are not — in nearly every instance (the random background now and then adds a chance hit that moves an instance to another class: it is then
    right for another reason)."""
    as_designed, n = {}, 0
    for seed in seeds:
        rng = np.random.default_rng(seed)
        n += 1
        for name, q, t, bits in _crafted(rng):
            bad, got = _check(lib, oracle, [q], [t])
            assert bad == 0, (seed, name, got)
            bad_full, _ = _check(lib, oracle, [q], [t], full=True)
            assert bad_full == 0, (seed, name)
            ok = all(_check(lib, oracle, [q], [t], weaken=bit)[0] == (1 if bit in bits else 0) for bit in (1, 2, 4, 8))
            as_designed[name] = as_designed.get(name, 0) + int(ok)
    assert len(as_designed) == 4
    for name, k in as_designed.items():
        assert k >= 0.85 * n, (name, k, n)


def test_emu_rescue_certificate_random_windows(emu, oracle):
    _run_random(emu, oracle, 240, 11)


def test_emu_rescue_certificate_crafted_windows(emu, oracle):
    _run_crafted(emu, oracle, range(3, 33))


@pytest.mark.gpu
def test_rescue_certificate_random_windows(hip, oracle):
    for seed in (21, 22, 23):
        _run_random(hip, oracle, 1600, seed)


@pytest.mark.gpu
def test_rescue_certificate_crafted_windows(hip, oracle):
    _run_crafted(hip, oracle, range(3, 203))
