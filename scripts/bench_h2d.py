"""PCIe-inclusive rate: every step uploads its 5 images (113.7 MB, pinned host memory) inside the timed region.
Not the headline (`value` has inputs resident) -- DESIGN.md section 5 quotes this next to it.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, hostmirror
from mdfnet_hip.pipeline import InFlight
dev = torch.device('cuda', 0)
model = bench.build(dev)
imgs, extr, intr, dr = synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100)
host_imgs = [imgs.clone().pin_memory() for _ in range(4)]
for n in (1, 3):
    pipe = InFlight(dev, n)
    def step(k):
        x = host_imgs[k % 4].to(dev, non_blocking=True)           # on the caller's stream; the item's stream waits for it
        cams = [t.to(dev, non_blocking=True) for t in (extr, intr, dr)]
        for g, c in zip(cams, (extr, intr, dr)): hostmirror.put(g, c)
        pipe.submit(lambda x=x, c=cams: model(x, *c), keep=(x, cams))
    with torch.no_grad():
        for k in range(6): step(k)
        pipe.drain(); torch.cuda.synchronize()
        K = 24; t0 = time.perf_counter()
        for k in range(K): step(k)
        pipe.drain(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} in flight, images uploaded every step: {K/dt:7.1f} views/s ({1e3*dt/K:5.2f} ms per view; H2D {imgs.numel()*4/1e6:.1f} MB per view)", flush=True)
