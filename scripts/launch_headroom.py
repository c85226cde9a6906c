"""Every C-ABI launch of one eval forward with its algorithmic work, bracketed time and the time it would take at the roof that
bounds it (fp32 MFMA 157.3 TFLOP/s / HBM 8 TB/s): sorted by the ABSOLUTE time above the roof.  dev tool."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ops
dev = torch.device('cuda', 0)
model = bench.build(dev)
inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100))
REP = 5
with torch.no_grad():
    for _ in range(3):
        model(*inputs)
    torch.cuda.synchronize()
    runs = []
    for _ in range(REP):
        ops.profile_begin()
        model(*inputs)
        runs.append(ops.profile_end())
rows = []
for i, (n, tag, ms, work) in enumerate(runs[0]):
    ms = min(r[i][2] for r in runs)
    fl, by = float(work.get("flops", 0)), float(work.get("bytes", 0))
    roof = max(fl / 157.3e12, by / 8e12) * 1e3
    rows.append((ms - roof, ms, roof, fl, by, work.get("kernel", "").split("(")[0][-60:], n, tag))
tot = sum(r[1] for r in rows)
print(f"{len(rows)} launches, {tot:.3f} ms (event-bracketed, min of {REP})")
for ex, ms, roof, fl, by, k, n, tag in sorted(rows, key=lambda r: -r[0]):
    print(f"{1e3*ms:7.1f}us roof {1e3*roof:6.1f}us excess {1e3*ex:6.1f}us  {fl/1e9:6.2f} GF {by/1e6:7.1f} MB  {fl/ms/1e9/157.3 if ms else 0:5.2f}  {k:45s} {tag}")
