"""BASELINE config 4 shape (Tanks&Temples-like 1920x1056, 7 views, metric depth range) -- informational.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth
from mdfnet_hip.pipeline import InFlight
dev = torch.device('cuda', 0)
model = bench.build(dev)
imgs = synth.make_images(1920, 1056, 7, batch=1, seed=5)
intr, extr, dr = synth.make_cameras(1920, 1056, 7, batch=1, rot_deg=2.0, seed=6, depth_range=(0.5, 10.0), baseline=0.25)
inputs = tuple(t.to(dev) for t in (imgs, extr, intr, dr))
with torch.no_grad():
    for n in (1, 3):
        pipe = InFlight(dev, n)
        for _ in range(2 * n + 1): pipe.submit(lambda: model(inputs[0], inputs[1].clone(), inputs[2].clone(), inputs[3].clone()))
        pipe.drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 12
        for _ in range(K): pipe.submit(lambda: model(inputs[0], inputs[1].clone(), inputs[2].clone(), inputs[3].clone()))
        pipe.drain(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"cfg4 1920x1056x7, {n} in flight: {K/dt:6.1f} views/s ({1e3*dt/K:5.2f} ms per view), peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
