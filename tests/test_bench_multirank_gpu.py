"""bench.py's N > 1 path on the one device of the test box: two ranks spawned by bench.py itself through
torch.distributed.run, sharing GPU 0 (LH_BENCH_SHARE_GPU=1: control collectives over gloo), a small genome, two steps."""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, LH_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mb", "64", "--barcodes", "2000", "--no-cpu-baseline", "--no-extras"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 2 and d["scaling"] == "weak"
    assert len(d["per_rank"]) == 2 and all(r["pairs_per_s"] > 0 and r["index_build_s"] > 0 for r in d["per_rank"])
    # weak scaling: the job aligned 2 ranks x 2 steps x 200,000 pairs
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * 2 / (2 * 2 * 200000) - 1) < 0.05


def test_bench_config4_legs_two_ranks_on_one_device():
    """`bench.py --gpus 2 --repeats`: the configs[4] legs under the launcher — every rank builds the repeat-family genome and its index, aligns its own
    batches; the line carries the whole job's value (both ranks' pairs over the slower rank's time) and every rank's time"""
    env = dict(os.environ, LH_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--repeats", "--genome-mb", "1200", "--barcodes", "1000"]   # (a genome large enough to have unique sequence between the repeat windows: the mixed leg draws 95 % of its reads there)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for leg, pairs in (("repeats", 200 * 100), ("mixed", 1000 * 100)):
        L = d[leg]
        assert L["n_gpus"] == 2 and L["value"] > 0 and L["candidates_per_read"]["mean"] > (10 if leg == "repeats" else 1.5)
        assert len(L["per_rank_timed_s"]) == 2 and all(t > 0 for t in L["per_rank_timed_s"])
        # whole-job value: both ranks' pairs of a step over the slower rank's step
        assert abs(L["value"] * L["ms_per_step"] * 1e-3 / (2 * pairs) - 1) < 0.05


def run_bench(args, env=None, launcher=False):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29611"]
    cmd += [os.path.join(helpers.ROOT, "bench.py")] + args
    p = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", **(env or {})), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_under_the_launcher_equals_plain_run():
    """the driver starts N > 1 runs as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: the same path with one rank (RCCL
    process group on the one device, barrier + all_reduce around the timed region) must report what the plain `python bench.py` reports"""
    args = ["--gpus", "1", "--steps", "30", "--warmup", "3", "--genome-mb", "400", "--barcodes", "5000", "--no-cpu-baseline", "--no-extras"]
    plain = run_bench(args)
    launched = run_bench(args, launcher=True)
    assert plain["n_gpus"] == launched["n_gpus"] == 1 and launched["steps"] == 30
    assert plain["work_per_step"] == launched["work_per_step"]   # the same batches, the same work
    rel = abs(launched["value"] / plain["value"] - 1)
    print("bench.py plain %.0f pairs/s, under torch.distributed.run %.0f pairs/s (%.1f %% apart)" % (plain["value"], launched["value"], 100 * rel))
    assert rel < 0.03
