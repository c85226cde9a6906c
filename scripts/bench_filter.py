"""Fused consistency filter at DTU size (1600x1200, 10 source views): kernel time vs HBM roofline.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/mdf-net_amd']
import numpy as np, torch
from mdfnet_hip import ops, synth
h, w, n = 1200, 1600, 10
dev = "cuda:0"
intr, extr, _ = synth.make_cameras(w, h, n + 1, batch=1, rot_deg=3.0, seed=3)
yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
rng = np.random.RandomState(0)
depths = [torch.from_numpy((650 + 0.02 * xx - 0.01 * yy + rng.normal(0, 1.0, (h, w))).astype(np.float32)).to(dev) for _ in range(n + 1)]
conf = torch.rand(h, w, device=dev)
args = (depths[0], conf, intr[0, 0], extr[0, 0], depths[1:], [intr[0, v] for v in range(1, n + 1)], [extr[0, v] for v in range(1, n + 1)])
for _ in range(3): ops.consistency_fuse(*args)
torch.cuda.synchronize()
ops.profile_begin()
for _ in range(10): r = ops.consistency_fuse(*args)
rec = [x for x in ops.profile_end() if x[0] == "mdf_consistency_fuse_fwd"]
ms = sum(x[2] for x in rec) / len(rec); by = rec[0][3]["bytes"]
t0 = time.perf_counter()
for _ in range(10): ops.consistency_fuse(*args)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
print(f"consistency_fuse_kernel {w}x{h} nsrc={n}: {ms*1e3:.1f} us per reference view, algorithmic {by/1e6:.1f} MB -> {by/ms/1e6:.0f} GB/s "
      f"({by/ms/1e6/8000:.3f} of 8 TB/s); host+device wall per call {wall*1e3:.2f} ms; final mask keeps {float(r['final_mask'].float().mean()):.3f}")
