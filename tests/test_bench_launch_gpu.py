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
                        "--no-cpu-baseline", "--no-profile", "--train-steps", "2", "--blocks", "1"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert "2 rank(s)" in rec["config"]["parallelism"]
    # the N > 1 training leg: every rank ran the cfg3 step with the flat-bucket exchange (gloo stands in for RCCL on this box)
    tr = rec["training"]
    assert tr["n_gpus"] == 2 and tr["backend"] == "gloo" and tr["steps"] == 2
    for leg in ("allreduce", "no_exchange"):
        assert tr[leg]["ms_per_step"] > 0 and tr[leg]["samples_per_s"] > 0, tr[leg]
    assert tr["allreduce"]["exchange_ms_in_step"] > 0 and tr["allreduce"]["collective_alone_ms"] > 0
    assert tr["samples_per_s"] == tr["allreduce"]["samples_per_s"]
    assert "direct" in tr and ("ms_per_step" in tr["direct"] or "error" in tr["direct"])
    assert "cfg4" not in rec and "cfg5_scan" not in rec            # N = 1 blocks only


def test_bench_extra_config_blocks_at_reduced_size():
    """cfg4 / cfg5_scan blocks of the N = 1 line (BASELINE configs[3] and [4]) run and carry their keys; a 7-view scan at a
    reduced size so that it takes seconds (the driver's run uses 49 views at 1600x1184)."""
    import torch
    sys.path[:0] = [ROOT]
    os.environ["MDF_BENCH_SIZE"] = "320x256x5"
    import importlib
    import bench
    try:
        importlib.reload(bench)
        dev = torch.device("cuda", 0)
        r = bench.cfg5_scan_block(dev, 2, nviews=12)
        assert r["model"]["views_per_s"] > 0 and r["model"]["without_feature_cache_views_per_s"] > 0
        assert r["filter"]["kernel_us_per_view"] > 0 and r["filter"]["nsrc"] == 10 and r["scan_views_per_s"] > 0
        intr, extr, dr, pairs = bench.scan_cameras(320, 256, 12)
        assert all(len(p) == 10 and len(set(p)) == 10 and i not in p for i, p in enumerate(pairs))
        assert pairs[5][:4] == [4, 6, 3, 7]
        c4 = bench.cfg4_block(dev, 2, items=2)                       # full 1920x1056 (the block has no size knob), 2 items per figure
        assert c4["7_views"]["views_per_s"] > 0 and c4["11_views"]["views_per_s"] > 0
    finally:
        os.environ.pop("MDF_BENCH_SIZE", None)
        importlib.reload(bench)        # back to the full-size constants for any later importer in this process
