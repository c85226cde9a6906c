"""The stride-1 3x3x3 layers that wino3d.hip serves, at their cfg2 shapes (dev; MDF_HIP_LIB selects a variant build)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
dev = "cuda:0"
L = [(32, 16, 48, 148, 200), (16, 16, 48, 148, 200), (16, 16, 12, 148, 200), (16, 16, 4, 296, 400), (32, 32, 24, 74, 100), (32, 32, 6, 74, 100)]
tag = os.environ.get("MDF_HIP_LIB", "default").split("libmdfnet_hip")[-1]
tot = 0.0
for ci, co, D, H, W in L:
    x = torch.randn(1, D, H, W, ci, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(co, ci, 3, 3, 3, device=dev) / (27 * ci) ** 0.5)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    for _ in range(3): ops.conv3d_ndhwc(x, wp, ci, co, 1, False, al, be, True)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv3d_ndhwc(x, wp, ci, co, 1, False, al, be, True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    tot += best
    print(f"[{tag}] {ci}->{co} {D}x{H}x{W}: {best*1e3:7.1f} us  {2*27*ci*co*D*H*W/best/1e9:6.1f} TFLOP/s", flush=True)
print(f"[{tag}] sum {tot*1e3:.1f} us")
