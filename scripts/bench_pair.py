"""conv01 of the feature pyramid at cfg2 (5 x 1184 x 1600): the fused pair launch vs the two single-layer launches. dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
n, h, w = 5, 1184, 1600
x = torch.rand(n, 3, h, w, device=dev)
w1 = torch.randn(8, 3, 3, 3, device=dev) / 27 ** 0.5; w2 = torch.randn(8, 8, 3, 3, device=dev) / 72 ** 0.5
a1, b1, a2, b2 = (torch.rand(8, device=dev) + 0.5 for _ in range(4))
wp1, wp2 = ops.pack_conv2d_weight(w1), ops.pack_conv2d_weight(w2)
def two():
    t1 = ops.conv2d_nhwc(x, wp1, 3, 8, 3, 1, a1, b1, True, planar_in=True)
    return ops.conv2d_nhwc(t1, wp2, 8, 8, 3, 1, a2, b2, True)
def one():
    return ops.conv2d_pair_planar(x, wp1, a1, b1, wp2, a2, b2)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("max abs diff one vs two:", float((one() - two()).abs().max()))
for _ in range(2):
    print(f"MDF_PAIR_BLOCKS={os.environ.get('MDF_PAIR_BLOCKS', '-')}: two launches {timeit(two):7.1f} us   one fused launch {timeit(one):7.1f} us", flush=True)
