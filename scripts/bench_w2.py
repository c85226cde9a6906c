"""The 2-D Winograd layers of the pyramid at cfg2's sizes (3x3 stride 1 and the 5x5 stride-2 layers over parity images): us per launch.  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops, lib
dev = 'cuda:0'
for ci, co, k, s, h, w in [(8, 16, 5, 2, 1184, 1600), (16, 32, 5, 2, 592, 800), (32, 64, 5, 2, 296, 400), (16, 16, 3, 1, 592, 800), (32, 32, 3, 1, 296, 400), (64, 64, 3, 1, 148, 200)]:
    x = torch.randn(5, h, w, ci, device=dev)
    wp = ops.pack_conv2d_weight(torch.randn(co, ci, k, k, device=dev) * 0.05)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    for _ in range(3): ops.conv2d_nhwc(x, wp, ci, co, k, s, al, be, True)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_nhwc(x, wp, ci, co, k, s, al, be, True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{ci}->{co} k{k}s{s} {h}x{w}x5: {best:7.1f} us  [{lib().mdf_last_launch().decode()[:40]}]")
