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
