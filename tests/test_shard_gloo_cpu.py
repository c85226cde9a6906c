"""CPU, 2 processes over gloo: the N>1 eval path — items sharded across ranks with no data-path collective, every
item written exactly once, timing = max over ranks.  (The model is a stand-in: the HIP model needs a GPU.)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, time
sys.path[:0] = [%(pkg)r]
import torch
from mdfnet_hip import shard
from load.dtueval import LoadDataset
import importlib.util
spec = importlib.util.spec_from_file_location("mdf_eval", os.path.join(%(pkg)r, "eval.py"))
ev = importlib.util.module_from_spec(spec); spec.loader.exec_module(ev)

class Fake(torch.nn.Module):            # same output contract as CoreNet.forward in eval mode
    def forward(self, imgs, extr, intr, dr):
        d = 425.0 + imgs[:, 0].mean(1) * 100.0
        return {"depth": d.float(), "confidence": torch.ones_like(d).float()}

rank, world, local = shard.init("gloo")
assert world == 2
root = %(root)r
ds = LoadDataset(root, os.path.join(root, "pair.txt"), [1, 4], nviews=3)
n, busy = ev.run_eval(Fake(), ds, torch.device("cpu"), %(out)r, rank, world, nworks=0, log=lambda *a: None)
tot = shard.sum_over_ranks(n); slow = shard.max_over_ranks(busy + rank)   # rank 1 pretends to be 1 s slower
shard.barrier()
if rank == 0:
    print(json.dumps({"total": tot, "slowest": slow, "mine": n}))
'''


def test_two_rank_gloo_eval_sharding(tmp_path):
    sys.path[:0] = [os.path.join(ROOT, "mdf-net_amd")]
    from load import synthetic
    from tools import data_io
    from mdfnet_hip import shard
    assert shard.shard_items(10, 0, 2) == [0, 2, 4, 6, 8] and shard.shard_items(10, 1, 2) == [1, 3, 5, 7, 9]
    assert shard.shard_items(3, 2, 8) == [2] and shard.shard_items(3, 5, 8) == []
    root = synthetic.write_dtu_eval_set(str(tmp_path / "dtu"), scans=(1, 4), nviews_total=5, width=64, height=32)
    out = str(tmp_path / "out")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"pkg": os.path.join(ROOT, "mdf-net_amd"), "root": root, "out": out})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["total"] == 10 and rec["mine"] == 5 and rec["slowest"] >= 1.0
    for scan in (1, 4):
        for v in range(5):
            f = os.path.join(out, "scan%d" % scan, "depth_est", "%08d.pfm" % v)
            assert os.path.exists(f) and os.path.exists(f.replace(".pfm", ".png"))
            assert os.path.exists(os.path.join(out, "scan%d" % scan, "confidence", "%08d.pfm" % v))
            d, _ = data_io.read_pfm(f)
            assert d.shape == (32, 64) and np.isfinite(d).all()
