"""Host issue time of one eval forward at cfg2 against its GPU time (dev tool): the loop below enqueues N forwards without
synchronising; while the host is faster than the GPU its time per call is pure issue time."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/mdf-net_amd']
import torch, bench
from mdfnet_hip import synth, hostmirror
dev = torch.device('cuda', 0)
model = bench.build(dev)
inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100))
cams_cpu = tuple(t.cpu() for t in inputs[1:])
def fresh():
    host = tuple(t.clone() for t in cams_cpu)
    devs = tuple(t.to(dev, non_blocking=True) for t in host)
    for d_, h_ in zip(devs, host): hostmirror.put(d_, h_)
    return devs
with torch.no_grad():
    for _ in range(5): model(inputs[0], *fresh())
    torch.cuda.synchronize()
    for n in (4, 8, 16):
        t0 = time.perf_counter()
        for _ in range(n): model(inputs[0], *fresh())
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{n:2d} forwards: host issue {1e3 * (t1 - t0) / n:.2f} ms per forward, wall {1e3 * (t2 - t0) / n:.2f} ms per forward", flush=True)
