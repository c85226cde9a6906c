"""Microbenchmark of single 2-D conv layers (dev tool).  MDF_AB=<env> MDF_AB_VALS=a,b,c for in-process A/B."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
L = [("refine 8->8", 8, 8, 3, 1, 1, 592, 800), ("refine 1->8", 1, 8, 3, 1, 1, 592, 800), ("refine 8->1", 8, 1, 3, 1, 1, 1184, 1600),
     ("bb 8->8", 8, 8, 3, 1, 5, 1184, 1600), ("bb 3->8", 3, 8, 3, 1, 5, 1184, 1600), ("head 16->16 k1", 16, 16, 1, 1, 5, 592, 800),
     ("bb 8->16 k5s2", 8, 16, 5, 2, 5, 1184, 1600), ("bb 16->16", 16, 16, 3, 1, 5, 592, 800), ("bb 16->32 k5s2", 16, 32, 5, 2, 5, 592, 800),
     ("bb 32->32", 32, 32, 3, 1, 5, 296, 400), ("bb 32->64 k5s2", 32, 64, 5, 2, 5, 296, 400), ("bb 64->64", 64, 64, 3, 1, 5, 148, 200),
     ("head 64->64 k1", 64, 64, 1, 1, 5, 148, 200), ("head 64->32 k1", 64, 32, 1, 1, 5, 148, 200), ("head 32->32 k1", 32, 32, 1, 1, 5, 296, 400),
     ("head 64->16 k1", 64, 16, 1, 1, 5, 148, 200), ("head 32->16 k1", 32, 16, 1, 1, 5, 296, 400)]
if len(sys.argv) > 1:
    L = [l for l in L if sys.argv[1] in l[0]]
dev = "cuda:0"
ab = os.environ.get("MDF_AB"); vals = os.environ.get("MDF_AB_VALS", "0").split(",")
for name, ci, co, k, st, b, h, w in L:
    x = torch.randn(b, h, w, ci, device=dev)
    wp = ops.pack_conv2d_weight(torch.randn(co, ci, k, k, device=dev) * 0.1)
    def timeit(n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ops.conv2d_nhwc(x, wp, ci, co, k, st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    timeit(3)
    res = []
    for v in vals:
        if ab: os.environ[ab] = v
        res.append(min(timeit() for _ in range(3)))
    print(f"{name:16s} {ci}->{co} k{k} {h}x{w}x{b}: " + "  ".join(f"{ab}={v}: {t:7.1f} us" for v, t in zip(vals, res)))
