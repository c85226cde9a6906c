"""The recorded training step (eight graphs, stage chains side by side: mdfnet_hip/graphstep.py) against the eager step over MANY steps
on the cfg3 shape, from the same weights and on the same sample: a race between the graphs (a missing join, two graphs sharing a
block of a memory pool) would show as a non-finite or drifting loss.  dev tool
    python scripts/diag_graph_long.py [steps]"""
import copy, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ddp
from mdfnet_hip.graphstep import GraphedTrainStep
from mdfnet_hip.optim import FlatAdam
from net import loss as loss_mod
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, V = 768, 576, 5
lr = 1e-4
me = bench.build(dev).train()
mg = copy.deepcopy(me)
be, bg = ddp.FlatBucket(me), ddp.FlatBucket(mg)
oe, og = FlatAdam(be, lr=lr), FlatAdam(bg, lr=lr)
crit = loss_mod.Loss().to(dev)
imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(W, H, V, batch=1, rot_deg=2.0, seed=3))
gt = {str(k): (torch.rand(1, H >> k, W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}


def eager():
    out = me(imgs, extr, intr, dr)
    loss = crit(out, gt, dr)
    be.zero_grad(); loss.backward(); be.allreduce_gradients(); oe.step()
    return float(loss)


for _ in range(2):
    eager()                                       # the recording's two warm-up steps, eagerly on the other model
step = GraphedTrainStep(mg, crit, bg, og, (imgs, extr, intr, dr, gt), warmup=2)
worst, bad = 0.0, None
t0 = time.time()
for it in range(steps):
    le = eager()
    lg = float(step(imgs, extr, intr, dr, gt))
    rel = abs(lg - le) / abs(le)
    worst = max(worst, rel)
    fin = bool(torch.isfinite(bg.flat).all()) and lg == lg
    if it < 5 or it % 25 == 0 or not fin:
        print(f"step {it:4d} loss eager {le:10.4f} recorded {lg:10.4f} rel {rel:.2e} finite {fin} t={time.time() - t0:.1f}s", flush=True)
    if not fin:
        bad = it
        break
pd = max(float((p - q).abs().max()) for p, q in zip(mg.parameters(), me.parameters()))
print(f"done: {it + 1} steps, worst relative loss difference {worst:.2e}, max parameter difference {pd:.2e} (Adam moves a weight by <= lr = {lr} per step), "
      f"first non-finite step: {bad}", flush=True)
