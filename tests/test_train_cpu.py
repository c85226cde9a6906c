"""CPU: the REHEARSAL backend (mdf-net_amd/rehearsal: the slots' training mode on stock ops, selected explicitly) against the
reference's training golden, the one-process-per-GPU data parallelism (flat gradient bucket, one all-reduce) over gloo on it, and the
product's own dispatch refusing CPU tensors in eval AND training mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mdfnet_hip import synth
from modelutil import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def test_training_forward_backward_vs_reference_golden(golden, seeded_sd, rehearsal_backend):
    from net.loss import Loss
    g = golden("train_tiny.npz")
    m = build_model()
    m.load_state_dict(seeded_sd)
    m.train()
    imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
    out = m(imgs, extr, intr, dr)
    assert isinstance(out["depth"], list) and len(out["depth"]) == 4            # core.py:72-73
    for i, d in enumerate(out["depth"]):
        np.testing.assert_allclose(d.detach().numpy(), g[f"depth{i}"], rtol=0, atol=1e-3)
    gt = {k: T(g["gt" + k]) for k in ("3", "2", "1", "0")}
    loss = Loss()(out, gt, dr)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-6)
    loss.backward()
    params = dict(m.named_parameters())
    assert all(p.grad is not None for p in params.values())                       # every parameter trains
    for k in g:
        if k.startswith("grad:"):
            ref = g[k]
            np.testing.assert_allclose(params[k[5:]].grad.numpy(), ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max()))


def test_eval_and_training_mode_refuse_cpu_tensors(seeded_sd):
    """no second backend behind the slots -- without `rehearsal.enable()` CPU tensors raise in both modes."""
    import rehearsal
    assert not rehearsal.enabled()
    m = build_model()
    m.load_state_dict(seeded_sd)
    imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=1, seed=1)
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(imgs, extr, intr, dr)
    m.train()
    with pytest.raises(RuntimeError, match="rehearsal.enable"):
        m(imgs, extr, intr, dr)


WORKER = r'''
import os, sys, json
sys.path[:0] = [%(pkg)r, %(tests)r]
import numpy as np, torch, torch.distributed as dist
from mdfnet_hip import ddp, shard, synth
import rehearsal; rehearsal.enable()      # CPU ranks: the stock-op backend, selected explicitly
from modelutil import build_model
from net.loss import Loss
torch.set_num_threads(2)
rank, world, _ = shard.init("gloo")
m = build_model()
sd = synth.seeded_state_dict(m.state_dict(), seed=1 + rank)       # ranks start DIFFERENT: broadcast must fix that
m.load_state_dict(sd); m.train()
bucket = ddp.FlatBucket(m)
bucket.broadcast_parameters(0)
imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
rng = np.random.RandomState(7)
gt = {k: torch.from_numpy((425 + 510 * rng.rand(2, 64 // s, 96 // s)).astype(np.float32)) for k, s in (("3", 8), ("2", 4), ("1", 2), ("0", 1))}
sl = slice(rank, rank + 1)                                         # batch split on dim 0: one sample per rank
out = m(imgs[sl], extr[sl], intr[sl], dr[sl])
loss = Loss()(out, {k: v[sl] for k, v in gt.items()}, dr[sl])
bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients()
if rank == 0:
    P = dict(m.named_parameters())
    keys = ["Backbone.conv01.0.conv.weight", "Homoaggre.0.depth_weight.0.conv.weight", "Regular.2.prob.weight", "Refine.conv2.2.weight"]
    np.savez(%(out)r, flat=bucket.flat.numpy(), **{k: P[k].grad.numpy() for k in keys})
    print(json.dumps({"n": int(bucket.flat.numel()), "ntensors": len(bucket.params), "loss": float(loss)}))
dist.barrier()
'''


def test_two_rank_gloo_flat_bucket_matches_mean_of_replica_gradients(tmp_path, seeded_sd, rehearsal_backend):
    """DDP semantics = DataParallel's: each replica normalises with its own batch statistics; the synchronised gradient is
    the mean of the per-replica gradients.  Checked against two single-process backward passes."""
    from net.loss import Loss
    out = str(tmp_path / "grads.npz")
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"pkg": os.path.join(ROOT, "mdf-net_amd"), "tests": os.path.join(ROOT, "tests"), "out": out})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29641", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n"] == 1206380 and rec["ntensors"] == 158          # one flat 4.83 MB bucket
    got = np.load(out)
    # reference computation in this process: rank-0 weights, one backward per sample, mean of the gradients.
    # Same intra-op thread count as the workers: CPU reduction orders depend on it, and the stage-1 curve fit
    # amplifies 1e-7 differences in the probability volume to ~1e-3 in the gradients.
    nthreads = torch.get_num_threads()
    torch.set_num_threads(2)
    imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
    rng = np.random.RandomState(7)
    gt = {k: T((425 + 510 * rng.rand(2, 64 // s, 96 // s)).astype(np.float32)) for k, s in (("3", 8), ("2", 4), ("1", 2), ("0", 1))}
    acc = None
    for i in range(2):
        m = build_model()
        m.load_state_dict(seeded_sd)                                # seed=1 == rank 0's weights
        m.train()
        sl = slice(i, i + 1)
        o = m(imgs[sl], extr[sl], intr[sl], dr[sl])
        Loss()(o, {k: v[sl] for k, v in gt.items()}, dr[sl]).backward()
        flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
        acc = flat if acc is None else acc + flat
    torch.set_num_threads(nthreads)
    exp = (acc / 2).numpy()
    np.testing.assert_allclose(got["flat"], exp, rtol=1e-4, atol=1e-6 * float(np.abs(exp).max()))


EXCH_WORKER = r'''
import os, sys, json
sys.path[:0] = [%(pkg)r]
import torch, torch.distributed as dist
from mdfnet_hip import ddp, shard
rank, world, _ = shard.init("gloo")
torch.manual_seed(0)
lin = torch.nn.Sequential(torch.nn.Linear(37, 11), torch.nn.Linear(11, 3))     # 37*11+11+11*3+3 = 454 elements: not a multiple of 3
res = {}
for mode in ("allreduce", "direct"):
    b = ddp.FlatBucket(lin)
    g = torch.Generator().manual_seed(100 + rank)
    b.flat.copy_(torch.randn(b.flat.numel(), generator=g))
    b.allreduce_gradients(mode)
    res[mode] = b.flat.clone()
exp = sum(torch.randn(res["direct"].numel(), generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
ok = bool(torch.allclose(res["direct"], exp, rtol=1e-6, atol=1e-7)) and bool(torch.allclose(res["allreduce"], exp, rtol=1e-6, atol=1e-7))
views = all(p.grad.data_ptr() >= b.flat.data_ptr() for p in lin.parameters())
gathered = [None] * world
dist.all_gather_object(gathered, [ok, views])
if rank == 0:
    print(json.dumps({"world": world, "ok": all(g[0] for g in gathered), "views": all(g[1] for g in gathered), "n": int(res["direct"].numel())}))
'''


def test_three_rank_gloo_direct_gradient_exchange_equals_allreduce(tmp_path):
    """SURVEY section 5's exchange for a fully connected xGMI node (all-to-all of shards, local sum, all-gather) gives the mean of
    the replica gradients, like the one all-reduce; 3 ranks and a buffer whose length is not a multiple of the world size."""
    script = tmp_path / "x.py"
    script.write_text(EXCH_WORKER % {"pkg": os.path.join(ROOT, "mdf-net_amd")})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
                        "127.0.0.1", "--master-port", "29643", str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec == {"world": 3, "ok": True, "views": True, "n": 454}


def test_flat_adam_cpu_rehearsal_matches_torch_adam():
    """The CPU leg of mdfnet_hip.optim.FlatAdam (used by the gloo rehearsals of train.py) applies torch.optim.Adam's update."""
    import torch
    from mdfnet_hip import ddp, optim as moptim
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b.load_state_dict(a.state_dict())
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    bucket = ddp.FlatBucket(b)
    ob = moptim.FlatAdam(bucket, lr=1e-2)
    x = torch.randn(16, 5)
    for _ in range(5):
        oa.zero_grad(); bucket.zero_grad()
        a(x).square().mean().backward(); b(x).square().mean().backward()
        bucket.allreduce_gradients()
        oa.step(); ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8)
