"""Throughput vs items per forward (batch) at cfg2 -- informational; bench.py's headline stays batch 1 like eval.py.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth
dev = torch.device('cuda', 0)
model = bench.build(dev)
for B in (1, 2, 4):
    inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=B, rot_deg=3.0, seed=100))
    with torch.no_grad():
        for _ in range(3): model(*inputs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 12
        for _ in range(n): model(inputs[0], inputs[1].clone(), inputs[2].clone(), inputs[3].clone())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {B}: {B*n/dt:7.1f} views/s  ({1e3*dt/n:6.2f} ms per forward)", flush=True)
