"""A Res block of the refinement net at cfg2 size (592x800, 1 image): one launch (res_pair.hip) vs the two conv launches.  dev tool
MDF_RES_PAIR_BLOCKS is read once per process: run once per value."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
dev = "cuda:0"
n, h, w = (int(v) for v in os.environ.get("MDF_SHAPE", "1,592,800").split(","))
x = torch.randn(n, h, w, 8, device=dev)
wa = ops.pack_conv2d_weight(torch.randn(8, 8, 3, 3, device=dev) / 72 ** 0.5)
wb = ops.pack_conv2d_weight(torch.randn(8, 8, 3, 3, device=dev) / 72 ** 0.5)
def two():
    t = ops.conv2d_nhwc(x, wa, 8, 8, 3, 1, None, None, True)
    return ops.conv2d_nhwc(t, wb, 8, 8, 3, 1, None, None, False, x, 0.1)
def one():
    return ops.conv2d_res_pair(x, wa, wb, 0.1)
assert torch.equal(one(), two())
for name, fn in (("two launches", two), ("one launch", one)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us  (MDF_RES_PAIR_BLOCKS={os.environ.get('MDF_RES_PAIR_BLOCKS', 'default')})", flush=True)
