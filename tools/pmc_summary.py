#!/usr/bin/env python3
"""Merge rocprofv3 --pmc counter_collection CSVs (one pass per counter group: FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950) into one per-kernel JSON that bench.py's `roofline.traffic` reads:

   python tools/pmc_summary.py --out profiles/r02_pmc_bench.json --commit $(git rev-parse --short HEAD) --command "bench.py --steps 3" \
          pass1/..._counter_collection.csv pass2/..._counter_collection.csv

Per kernel: the SUM over its dispatches of every counter (FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them), and the
number of dispatches.  Without --out: a table on stdout."""
import argparse
import collections
import csv
import json


def norm(name):
    name = name.split("(")[0].strip()
    return name[5:] if name.startswith("void ") else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--out")
    ap.add_argument("--commit", default=None)
    ap.add_argument("--command", default=None)
    ap.add_argument("--note", default=None)
    ap.add_argument("--ceiling", type=float, default=None, help="measured random-read ceiling (G accesses/s) at the table's real size")
    a = ap.parse_args()
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.Counter())
    for path in a.csv:
        for row in csv.DictReader(open(path)):
            k = norm(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
    kernels = {}
    for k, v in acc.items():
        e = {"calls": max(calls[k].values())}
        for c, x in v.items():
            key = {"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB", "TCC_MISS_sum": "TCC_MISS", "TCC_HIT_sum": "TCC_HIT"}.get(c, c)
            e[key] = x
        kernels[k] = e
    if a.out:
        import hashlib, os
        cs = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lariat_amd", "csrc")
        # every device-side source (kernels, lh_dev.h, the host files that set launch geometry and streams): bench.py reports traffic only while
        # these still match (bench.pmc_is_stale uses the same list)
        srcs = {f: hashlib.sha256(open(os.path.join(cs, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(cs)) if f.endswith((".h", ".inc", ".hip"))}
        doc = {"commit": a.commit, "command": a.command, "note": a.note, "kernel_sources": srcs,
               "units": "FETCH_SIZE_KB / WRITE_SIZE_KB: KB summed over the kernel's dispatches; calls = dispatches",
               "random_read_ceiling_Gaccess_per_s": a.ceiling, "kernels": kernels}
        json.dump(doc, open(a.out, "w"), indent=1, sort_keys=True)
    else:
        for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB", 0)):
            print(k, e)


if __name__ == "__main__":
    main()
