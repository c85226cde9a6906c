import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/mdf-net_amd']
from mdfnet_hip import ops
h, w, n = 1200, 1600, 3
yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
depth = torch.from_numpy((600 + 0.05 * xx - 0.03 * yy + 5 * np.sin(xx / 90.0)).astype(np.float32)).cuda()
conf = torch.rand(h, w).cuda()
K = torch.tensor([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1]]); E = torch.eye(4)
r = ops.consistency_fuse(depth, conf, K, E, [depth] * n, [K] * n, [E] * n, per_view=True)
d = (r["depth_avg"] - depth).abs()
print('geo all', bool(r['geo_mask'].all()), 'view masks all', bool(r['view_masks'].all()), 'max diff', float(d.max()))
i = int(d.argmax()); y, x = i // w, i % w
print('worst at', y, x, float(depth[y, x]), float(r['depth_avg'][y, x]), 'rep', [float(r['rep'][v, y, x]) for v in range(n)], 'neighbors', depth[y, x-1:x+2].tolist())
