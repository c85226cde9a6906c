import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/mdf-net_amd']
import torch
from mdfnet_hip import ops, train_ops
dev = torch.device('cuda', 0)
x = torch.randn(1, 4, 8, 16, 16, device=dev)
conv = torch.nn.Conv3d(16, 16, 3, padding=1, bias=False).to(dev)
wp = ops.pack_conv3d_weight(conv.weight)
def t(fn, n=2000):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return dt * 1e6
print("conv3d_ndhwc host us/call", t(lambda: ops.conv3d_ndhwc(x, wp, 16, 16, 1, False, None, None, False, None)))
print("bn_stats", t(lambda: train_ops.bn_stats(x, 4 * 8 * 16, 16)))
print("pack_conv3d_weight", t(lambda: ops.pack_conv3d_weight(conv.weight)))
print("torch.empty", t(lambda: torch.empty((1, 4, 8, 16, 16), device=dev)))
print("torch.zeros(32,f64)", t(lambda: torch.zeros(32, device=dev, dtype=torch.float64)))
print("torch add", t(lambda: x + x))
big = torch.randn(4096, 4096, device=dev)
h = torch.randn(12)
def busy_then(fn):
    def g():
        y = big @ big
        fn()
    return g
print("matmul alone", t(lambda: big @ big, 50))
print("matmul + .to(dev) pageable", t(busy_then(lambda: h.to(dev, non_blocking=True)), 50))
hp = h.pin_memory()
print("matmul + .to(dev) pinned", t(busy_then(lambda: hp.to(dev, non_blocking=True)), 50))
print("matmul + torch.tensor(list, device)", t(busy_then(lambda: torch.tensor([1, 2, 3], device=dev)), 50))
print("matmul + torch.full", t(busy_then(lambda: torch.full((1,), 0.5, device=dev)), 50))
