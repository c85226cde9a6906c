"""GPU: `python bench.py --gpus 2` (no torchrun environment) starts two ranks itself and reports n_gpus = 2.  Rehearsed on the
one-GPU box with both ranks sharing the card and gloo for the barrier/max-over-ranks (MDF_BENCH_SHARE_GPU, MDF_BENCH_DIST_BACKEND),
at a reduced size so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks():
    env = dict(os.environ, MDF_BENCH_SHARE_GPU="1", MDF_BENCH_DIST_BACKEND="gloo", MDF_BENCH_SIZE="320x256x5")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-profile"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert "2 rank(s)" in rec["config"]["parallelism"]
