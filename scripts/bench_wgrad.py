"""Weight-gradient kernels on the layer shapes of one cfg3 training step (or MDF_WGRAD_ONLY=<substring of the tag>), HIP-event
timed.  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from mdfnet_hip import train_ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
only = os.environ.get("MDF_WGRAD_ONLY", "")
reps = int(os.environ.get("MDF_WGRAD_REPS", "20"))
cases3d = [(16, 32, 1, (48, 72, 96)), (16, 16, 1, (48, 72, 96)), (32, 32, 1, (24, 36, 48)), (64, 64, 1, (12, 18, 24)), (8, 8, 1, (8, 288, 384)),
           (8, 16, 1, (24, 144, 192)), (32, 16, 2, (24, 36, 48)), (16, 8, 2, (4, 144, 192)), (1, 8, 1, (8, 288, 384))]
cases2d = [(8, 8, 3, 1, (576, 768, 5)), (8, 4, 3, 1, (576, 768, 5)), (16, 8, 5, 2, (288, 384, 5)), (16, 16, 3, 1, (288, 384, 5)), (32, 16, 5, 2, (144, 192, 5)),
           (32, 32, 3, 1, (144, 192, 5)), (64, 32, 5, 2, (72, 96, 5)), (64, 64, 3, 1, (72, 96, 5)), (16, 64, 1, 1, (288, 384, 5))]
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.0
for a, b, s, (d, h, w) in cases3d:
    tag = f"3d {a}x{b} s{s} {d}x{h}x{w}"
    if only and only not in tag: continue
    small = torch.randn(1, d, h, w, a, device=dev); big = torch.randn(1, d * s, h * s, w * s, b, device=dev)
    us = timeit(lambda: train_ops.conv3d_wgrad(small, big, s, (a, b, 3, 3, 3)))
    fl = 2.0 * 27 * a * b * d * h * w
    tot += us
    print(f"{tag:34s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
for a, b, k, s, (h, w, n) in cases2d:
    tag = f"2d {a}x{b} k{k}s{s} {h}x{w}x{n}"
    if only and only not in tag: continue
    small = torch.randn(n, h, w, a, device=dev); big = torch.randn(n, h * s, w * s, b, device=dev)
    us = timeit(lambda: train_ops.conv2d_wgrad(small, big, k, s, (a, b, k, k)))
    fl = 2.0 * k * k * a * b * n * h * w
    tot += us
    print(f"{tag:34s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
print(f"total {tot:.0f} us")
