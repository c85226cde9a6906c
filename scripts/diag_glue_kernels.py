"""Which torch (non hand-written) GPU kernels run in a forward, and from which Python lines?  dev tool (GPU box)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from mdfnet_hip import synth, hostmirror
dev = torch.device('cuda', 0)
model = bench.build(dev)
cpu_in = synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100)
imgs = cpu_in[0].to(dev)
def fresh():
    cams = [t.clone() for t in cpu_in[1:]]
    dv = [t.to(dev) for t in cams]
    for g, c in zip(dv, cams): hostmirror.put(g, c)
    return (imgs, *dv)
with torch.no_grad():
    for _ in range(3): model(*fresh())
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model(*fresh())
        torch.cuda.synchronize()
ev = prof.events()
rows = []
for e in ev:
    if e.device_time_total > 0 and e.cpu_parent is None or True:
        pass
# group GPU time by (op name, innermost repo stack frame)
agg = {}
for e in ev:
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0: continue
    if e.name.startswith("mdf_") : continue
    site = "?"
    for fr in (e.stack or []):
        if "/mdf-net_amd/" in fr or "bench.py" in fr or "/scripts/" in fr:
            site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-90:]
            break
    k = (e.name[:60], site[:110], str(e.input_shapes)[:60])
    a = agg.setdefault(k, [0.0, 0]); a[0] += dt; a[1] += 1
tot = sum(v[0] for v in agg.values())
print("torch-op GPU time in one forward: %.1f us" % tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us x%-3d %-45s %-60s %s" % (v[0], v[1], k[0], k[1], k[2]))
