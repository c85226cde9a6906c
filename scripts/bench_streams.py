"""Throughput with n items in flight on n HIP streams (batch 1 each) -- the eval driver's pipelining.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth
dev = torch.device('cuda', 0)
model = bench.build(dev)
inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100))
with torch.no_grad():
    for _ in range(3): ref = model(*inputs)
    torch.cuda.synchronize()
    for n in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(n)]
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        outs = []
        for k in range(2 * n):                      # warm the per-stream allocator pools
            with torch.cuda.stream(streams[k % n]): outs.append(model(inputs[0], inputs[1].clone(), inputs[2].clone(), inputs[3].clone()))
        torch.cuda.synchronize()
        K = 24
        t0 = time.perf_counter()
        for k in range(K):
            with torch.cuda.stream(streams[k % n]): out = model(inputs[0], inputs[1].clone(), inputs[2].clone(), inputs[3].clone())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        same = all(torch.equal(o["depth"], ref["depth"]) for o in outs) and torch.equal(out["depth"], ref["depth"])
        print(f"{n} stream(s): {K/dt:7.1f} views/s  ({1e3*dt/K:5.2f} ms per view), outputs identical to the 1-stream result: {same}", flush=True)
