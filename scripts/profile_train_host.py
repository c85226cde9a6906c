"""Host-side cost of one training step (python + launch overhead): issue time without synchronisation vs wall, and a cProfile
of the issuing thread.  dev tool"""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ddp
from net import loss as loss_mod
dev = torch.device('cuda', 0)
W, H, V = 768, 576, 5
model = bench.build(dev).train()
bucket = ddp.FlatBucket(model)
if os.environ.get("MDF_TRAIN_STOCK") == "1":
    import rehearsal
    rehearsal.enable(on_gpu=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
else:
    from mdfnet_hip.optim import FlatAdam
    opt = FlatAdam(bucket, lr=1e-3)
crit = loss_mod.Loss().to(dev)
imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(W, H, V, batch=1, rot_deg=2.0, seed=3))
gt = {str(k): (torch.rand(1, H >> k, W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}
def step():
    out = model(imgs, extr, intr, dr)
    loss = crit(out, gt, dr)
    bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t0) / n
print(f"issue {t_issue*1e3:.2f} ms/step, wall {t_wall*1e3:.2f} ms/step")
seg = [0.0] * 5
for _ in range(n):
    t = [time.perf_counter()]
    out = model(imgs, extr, intr, dr); t.append(time.perf_counter())
    loss = crit(out, gt, dr); t.append(time.perf_counter())
    bucket.zero_grad(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    bucket.allreduce_gradients(); opt.step(); t.append(time.perf_counter())
    for i in range(5): seg[i] += t[i + 1] - t[i]
torch.cuda.synchronize()
print("issue time per step (ms): forward %.2f, loss %.2f, zero_grad %.2f, backward %.2f, allreduce+adam %.2f" % tuple(1e3 * x / n for x in seg))
import threading
pr = cProfile.Profile()
for _ in range(5):
    pr.enable()
    out = model(imgs, extr, intr, dr)
    pr.disable()
    loss = crit(out, gt, dr)
    bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients(); opt.step()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
