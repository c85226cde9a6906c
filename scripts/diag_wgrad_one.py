"""One weight-gradient launch (partial tiles only, no slab sum), HIP-event timed over back-to-back launches. dev tool
usage: python scripts/diag_wgrad_one.py A B stride D H W   (env: MDF_WGRAD_TH, MDF_WGRAD_BLOCKS)"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from mdfnet_hip import lib, ops
a, b, s, d, h, w = (int(v) for v in sys.argv[1:7])
dev = torch.device('cuda', 0)
small = torch.randn(1, d, h, w, a, device=dev); big = torch.randn(1, d * s, h * s, w * s, b, device=dev)
L = lib()
n = L.mdf_conv3d_wgrad_workspace(1, d, h, w, a, b)
work = torch.empty(n, device=dev); dw = torch.empty(a * b * 27, device=dev)
ns = ctypes.c_int(0)
st = ops._stream(dw)
def fn():
    rc = L.mdf_conv3d_wgrad_partial(small.data_ptr(), big.data_ptr(), dw.data_ptr(), work.data_ptr(), 1, d, h, w, a, b, s, ctypes.byref(ns), st)
    assert rc == 0
for _ in range(5): fn()
torch.cuda.synchronize()
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * 27 * a * b * d * h * w
print(f"{a}x{b} s{s} {d}x{h}x{w} TH={os.environ.get('MDF_WGRAD_TH','-')} BLOCKS={os.environ.get('MDF_WGRAD_BLOCKS','-')} slabs={ns.value}: {us:7.1f} us  {fl/us/1e6:6.1f} TFLOP/s", flush=True)
