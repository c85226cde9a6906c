"""The three 5x5 stride-2 layers of the pyramid at cfg2's sizes: us per launch under the dev switches given in the environment.  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops, lib
dev = 'cuda:0'
for ci, co, h, w in [(8, 16, 1184, 1600), (16, 32, 592, 800), (32, 64, 296, 400)]:
    x = torch.randn(5, h, w, ci, device=dev)
    wp = ops.pack_conv2d_weight(torch.randn(co, ci, 5, 5, device=dev) * 0.05)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    for _ in range(3): ops.conv2d_nhwc(x, wp, ci, co, 5, 2, al, be, True)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_nhwc(x, wp, ci, co, 5, 2, al, be, True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{ci}->{co} k5s2 {h}x{w}x5: {best:7.1f} us  [{lib().mdf_last_launch().decode()[:40]}]")
