"""K6's exact shortcut under adversarial input (k_rescue2.h: resc_dedup_incremental).  After every mem_matesw attempt the reference calls
mem_sort_dedup_patch on the mate's whole region list (gobwa.go:291,315 -> mem_matesw); the pipeline replaces every call but the first by a function
of the one region that was added.  lh_diag_rescue_dedup runs both on the same lists: region lists crowded into a few kilobases so that most pairs
overlap, with the added region redundant with several entries, better than some and worse than others, on either side of them, on another
contig in between, at the edge of max_chain_gap.  CPU: the kernel sources under the emulator; -m gpu: the product."""
import os
import subprocess

import numpy as np
import pytest

import helpers
from lariat_amd import capi

EMU = os.environ.get("LH_EMU_LIB") or os.path.join(helpers.ROOT, "tests", "_build", "liblariat_emu.so")


PHYSICAL = [0, 1]


def cases(seed, n_cases, gap):
    rng = np.random.default_rng(seed)
    first, regs, added = [0], [], []

    def region(center, spread, rid_choices):
        ln = int(rng.integers(40, 160))
        rb = int(center + rng.integers(-spread, spread + 1))
        dl = int(rng.integers(-3, 4)) if rng.random() < 0.5 else 0
        qb = int(rng.integers(0, 150 - min(ln, 149)))
        rid = int(rng.choice(rid_choices))
        if rid_choices is PHYSICAL:   # two contigs as they are: an interval of coordinates each
            rid = int(rb + ln + dl > 5200)
        return [rb, rb + ln + dl, qb, min(150, qb + ln), int(rng.integers(19, 151)), rid]

    for c in range(n_cases):
        n = int(rng.integers(0, 40)) if c % 7 else int(rng.integers(40, 200))
        style = c % 5
        rids = [0] if style < 3 else (PHYSICAL if style == 3 else [0, 0, 0, 1])   # style 4: contigs that interleave (no real list does: every call as written)
        centers = [5000 + int(rng.integers(0, 400)) * (1 if style != 2 else 40) for _ in range(int(rng.integers(1, 5)))]
        spread = [30, 200, 60, 150, gap + 200][style]
        for _ in range(n):
            regs.append(region(int(rng.choice(centers)), spread, rids))
        first.append(len(regs))
        b = region(int(rng.choice(centers)), spread if style != 4 else 50, PHYSICAL if style == 3 else [0])
        if n and rng.random() < 0.5:   # b as a near-copy of an entry: a few bases longer or shorter, a slightly different score
            src = regs[first[-2] + int(rng.integers(0, n))]
            b = [src[0] + int(rng.integers(-4, 5)), src[1] + int(rng.integers(-4, 5)), src[2], src[3], max(19, src[4] + int(rng.integers(-20, 21))), src[5] if style == 3 else 0]
            if b[1] <= b[0]:
                b[1] = b[0] + 20
            if rng.random() < 0.3:   # b IS an entry (the window of a later anchor holds a region the mate already has): which of the two the call keeps is the introsort's to say
                b = list(src)
        added.append(b)
    return np.array(first, dtype=np.int32), np.array(regs, dtype=np.int64).reshape(-1, 6), np.array(added, dtype=np.int64)


def check(lib):
    tot = {0: 0, 1: 0, 2: 0, 3: 0}
    shrunk = grew = same = 0
    for seed, gap in ((1, 10000), (2, 10000), (3, 300), (4, 50), (5, 10000), (6, 120)):
        first, regs, added = cases(seed, 1500, gap)
        v, n = lib.diag_rescue_dedup(first, regs, added, max_chain_gap=gap)
        assert not (v == 2).any(), ("incremental dedup differs from mem_sort_dedup_patch", seed, np.nonzero(v == 2)[0][:10])
        assert not (v == 4).any(), ("the call as written on the list in LDS differs from the one in memory", seed, np.nonzero(v == 4)[0][:10])
        for k in range(4):
            tot[k] += int((v == k).sum())
        ok = v == 0
        n_full, n_clean = n & 0xffff, n >> 16
        shrunk += int((ok & (n_full < n_clean)).sum())     # b knocked out more than one entry
        same += int((ok & (n_full == n_clean)).sum())      # b lost (or replaced exactly one entry)
        grew += int((ok & (n_full == n_clean + 1)).sum())  # b joined, nobody left
    print("rescue dedup property test:", tot, "lists that shrank / kept their length / grew:", shrunk, same, grew)
    assert tot[0] > 4500 and tot[1] > 1000 and tot[3] > 1000 and shrunk > 100 and same > 300 and grew > 300


def test_emu_incremental_dedup_equals_the_full_call():
    subprocess.check_call(["make", "-s", "-C", os.path.join(helpers.ROOT, "tests", "hipemu")])
    check(capi.Library(EMU))


@pytest.mark.gpu
def test_incremental_dedup_equals_the_full_call():
    check(capi.load_library())
