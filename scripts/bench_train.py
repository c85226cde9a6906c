"""cfg3-shaped training step on one GPU (stock-op training path): 768x576, 5 views, batch 1, forward + loss + backward +
flat-bucket all-reduce (1 rank) + Adam.  Informational (the headline metric is inference).  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ddp
from net import loss as loss_mod
dev = torch.device('cuda', 0)
W, H, V = (int(x) for x in os.environ.get("MDF_TRAIN_SHAPE", "768,576,5").split(","))
model = bench.build(dev).train()
bucket = ddp.FlatBucket(model)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = loss_mod.Loss().to(dev)
imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(W, H, V, batch=1, rot_deg=2.0, seed=3))
gt = {str(k): (torch.rand(1, H >> k, W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}   # dtutrain.py:55-58 key order
def step():
    out = model(imgs, extr, intr, dr)
    loss = crit(out, gt, dr)
    bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients(); opt.step()
    return float(loss.detach())
for _ in range(2): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"train step {W}x{H}x{V} B=1: {dt*1e3:.1f} ms  ({1/dt:.2f} samples/s), loss {l:.3f}, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
