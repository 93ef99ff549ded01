"""N4 — accuracy accounting for simulated reads (host side, telemetry only; nothing here feeds back into the alignments).

Two reports, both keyed on the read-name convention `mol:<bc>:<chrom>:<mol start>:<mol end>:<pos1>:<pos2>` (7 colon fields):

* `SimulatedStats` — the counters lariat keeps under `-simulated` (inference/lariat.go:517-542): over the ACTIVE alignments of
  barcodes that went through RFA, `total`, `total_mapq10`, `correct` (|true pos − pos| < 600, contig not compared — as the
  reference does) and `correct_mapq10`.
* `check_report` — the report of go/check.py:41-105 over BAM records: fractions unmapped / proper pair / mapq = 0 / < 30 / >= 30
  and, per MAPQ bin (nearest of 5, 15, 30, 45), the median MAPQ, the empirical MAPQ −10·log10(1 − fraction correct) and the
  bin size.  check.py calls a record correct when its contig equals field 2 and |pos − field 5| < 200 for BOTH mates
  (field 5 is read 1's position); `mate_aware=True` compares read 2 with field 6 instead.
"""
import math

import numpy as np

MAPQ_BINS = (5, 15, 30, 45)


def truth_of(name):
    """(chrom, pos1, pos2) of a simulated read name, or None when the name does not follow the convention"""
    if isinstance(name, bytes):
        name = name.decode("ascii", "replace")
    p = name.split(":")
    if len(p) != 7 or p[0] != "mol":
        return None
    try:
        return p[2], float(p[5]), float(p[6].strip())
    except ValueError:
        return None


class SimulatedStats:
    """lariat.go:517-542 (`stats.total / total_mapq10 / correct / correct_mapq10`), accumulated over batches"""

    def __init__(self):
        self.total = self.total_mapq10 = self.correct = self.correct_mapq10 = 0
        self.placeholders = 0   # not a reference counter: active placeholders (reads without a hit, pos −1) — the reference counts them in total / total_mapq10

    def add(self, result, names, bc_pair_off, bc_do_rfa):
        """`result`: capi.Result of the batch; `names`: one read name per pair; barcodes with bc_do_rfa == 0 are skipped
        (the reference only counts inside the RFA branch of DoRFAForOneBarcode)"""
        act = np.flatnonzero(result.active != 0)
        if act.size == 0:
            return
        read_of = np.searchsorted(result.cand_off, act, side="right") - 1     # read_id of every active candidate
        pair_of = read_of >> 1
        bc_of = np.searchsorted(np.asarray(bc_pair_off), pair_of, side="right") - 1
        keep = np.asarray(bc_do_rfa)[bc_of] != 0
        truth = {}
        for c, r, p in zip(act[keep], read_of[keep], pair_of[keep]):
            t = truth.get(p)
            if t is None:
                t = truth[p] = truth_of(names[p]) or ()
            if not t:
                continue
            self.total += 1
            self.placeholders += int(result.pos[c] < 0)
            mq10 = result.mapq[c] >= 10
            self.total_mapq10 += int(mq10)
            pos = t[1] if (r & 1) == 0 else t[2]
            if abs(pos - float(result.pos[c])) < 600:
                self.correct += 1
                self.correct_mapq10 += int(mq10)

    def as_dict(self):
        d = {"total": self.total, "total_mapq10": self.total_mapq10, "correct": self.correct, "correct_mapq10": self.correct_mapq10}
        d["placeholders"] = self.placeholders
        d["frac_correct"] = self.correct / self.total if self.total else None
        d["frac_correct_mapq10"] = self.correct_mapq10 / self.total_mapq10 if self.total_mapq10 else None
        return d


def check_report(sam_lines, mate_aware=False, include_secondary=True):
    """go/check.py:41-105 over record text (the columns of `lh_records_text`, SAM order: QNAME FLAG RNAME POS MAPQ ...).
    POS in the text is 1-based; check.py compares pysam's 0-based `r.pos`."""
    n = unmapped = proper = mq0 = mq_lt30 = mq_ge30 = 0
    obs = []
    for ln in sam_lines:
        if not ln or ln[0] == "@":
            continue
        f = ln.split("\t", 6)
        flag, mapq = int(f[1]), int(f[4])
        if not include_secondary and flag & 0x900:
            continue
        n += 1
        unmapped += bool(flag & 4)
        proper += bool(flag & 2)
        mq0 += mapq == 0
        mq_lt30 += mapq < 30
        mq_ge30 += mapq >= 30
        t = truth_of(f[0])
        if t is not None:
            true_pos = t[2] if (mate_aware and flag & 0x80) else t[1]
            ok = (not flag & 4) and f[2] == t[0] and abs((int(f[3]) - 1) - int(true_pos)) < 200
            b = min(MAPQ_BINS, key=lambda o: (abs(mapq - o), o))   # np.argmin: first of equally near bins
            obs.append((b, mapq, ok))
    rep = {"records": n}
    if n:
        rep.update({"Unmapped": unmapped / n, "Proper pair": proper / n, "mapq = 0": mq0 / n, "mapq < 30": mq_lt30 / n, "mapq >= 30": mq_ge30 / n})
    bins = []
    for b in MAPQ_BINS:
        v = [(m, ok) for bb, m, ok in obs if bb == b]
        if not v:
            continue
        frac = sum(ok for _, ok in v) / len(v)
        bins.append({"bin": b, "med_map": float(np.median([m for m, _ in v])), "emp_mapq": (-10.0 * math.log10(1.0 - frac)) if frac < 1.0 else float("inf"),
                     "frac_correct": frac, "n": len(v)})
    rep["mapq_bins"] = bins
    return rep


def format_report(rep):
    out = ["records        : %d" % rep["records"]]
    for k in ("Unmapped", "Proper pair", "mapq = 0", "mapq < 30", "mapq >= 30"):
        if k in rep:
            out.append("{0:15}: {1:3f}".format(k, rep[k]))
    for b in rep.get("mapq_bins", []):
        out.append("mapq bin %2d: n %8d  median mapq %5.1f  empirical mapq %5.1f  (%.5f correct)" % (b["bin"], b["n"], b["med_map"], b["emp_mapq"], b["frac_correct"]))
    return "\n".join(out)
