"""The stride-2 3-D layers of cfg2's regularisers: us per launch (MDF_CONV_LDS_S2_MIN_VOXELS=-1: conv3d.hip's kernel).  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops, lib
dev = 'cuda:0'
for ci, co, d, h, w in [(8, 16, 8, 592, 800), (8, 16, 24, 296, 400), (16, 32, 48, 148, 200), (16, 32, 4, 296, 400), (16, 32, 12, 148, 200)]:
    x = torch.randn(1, d, h, w, ci, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05, False)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    for _ in range(3): ops.conv3d_ndhwc(x, wp, ci, co, 2, False, al, be, True)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv3d_ndhwc(x, wp, ci, co, 2, False, al, be, True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{ci}->{co} s2 {d}x{h}x{w}: {best:7.1f} us  [{lib().mdf_last_launch().decode()[:40]}]")
