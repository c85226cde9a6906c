"""Per-launch breakdown of one forward at cfg2 (HIP events on the launch stream).  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch, bench
from mdfnet_hip import synth, ops
dev = torch.device('cuda', 0)
model = bench.build(dev)
inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100))
with torch.no_grad():
    for _ in range(3): model(*inputs)
    acc = {}
    N = 5
    for _ in range(N):
        ops.profile_begin(); model(*inputs)
        for i, (name, tag, ms, work) in enumerate(ops.profile_end()):
            k = (i, name.replace('mdf_', '').replace('_fwd', ''), tag)
            a = acc.setdefault(k, [0.0, work]); a[0] += ms
tot = 0
for (i, name, tag), (ms, work) in sorted(acc.items()):
    ms /= N; tot += ms
    extra = ''
    if work.get('flops'): extra += f" {work['flops']/ms/1e9:7.1f} TF/s"
    if work.get('bytes'): extra += f" {work['bytes']/ms/1e6:7.0f} GB/s(alg)"
    print(f"{i:3d} {name:26s} {tag:34s} {ms*1e3:8.1f} us{extra}")
print(f"total {tot:.3f} ms")
