"""Diagnostic build with in-kernel cycle stamps: where does a conv_lds block spend its time?
   usage (on the GPU box): python scripts/diag_conv_stamps.py     (builds a separate .so; the product library is untouched)"""
import ctypes, os, subprocess, sys
os.environ.setdefault('MDF_CONV_WINO3D', '0'); os.environ.setdefault('MDF_CONV_WINO2D', '0')   # (the stamps live in conv_lds_kernel)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
csrc = R + '/mdf-net_amd/csrc'
so = '/tmp/libmdfnet_stamps.so'
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.cpp'))]
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '--offload-arch=gfx950', '-DMDF_STAMPS',
       '-I', R + '/include', '-I', csrc, '-x', 'hip'] + srcs + ['-o', so]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
import torch
import mdfnet_hip
mdfnet_hip.LIB_PATH = so
from mdfnet_hip import ops
lib = mdfnet_hip.lib()
lib.mdf_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = 'cuda:0'
for name, ci, co, D, H, W in ([] if os.environ.get("STAMPS_2D_ONLY") else [("32->16", 32, 16, 48, 148, 200), ("16->16", 16, 16, 48, 148, 200)]):
    x = torch.randn(1, D, H, W, ci, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
    for _ in range(2): ops.conv3d_ndhwc(x, wp, ci, co)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.mdf_debug_read_stamps(buf, 1)
    n = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.conv3d_ndhwc(x, wp, ci, co)
    e1.record(); torch.cuda.synchronize()
    lib.mdf_debug_read_stamps(buf, 1)
    sched, pro, comp, fill, tot, blocks, items, dsteps = [float(v) for v in buf]
    print(f"{name:7s} {e0.elapsed_time(e1)/n*1e3:7.1f} us/launch | per block-launch: total {tot/blocks:9.0f} cyc = sched {sched/tot:5.1%} prologue {pro/tot:5.1%} "
          f"compute {comp/tot:5.1%} refill {fill/tot:5.1%} | items/block {items/blocks:5.1f}, d-steps/item {dsteps/max(items-blocks,1):4.1f}, "
          f"cyc/d-step compute {comp/dsteps:7.0f} refill {fill/dsteps:6.0f}, cyc/item sched {sched/items:6.0f} prologue {pro/max(items-blocks,1):6.0f}")

print("2-D layers (per tile: compute = MFMA step incl. epilogue stores, refill = wait for the next tile's loads + LDS stores + barrier)")
for name, ci, co, k, st, b, h, w in [("refine 8->8", 8, 8, 3, 1, 1, 592, 800), ("refine 8->32", 8, 32, 3, 1, 1, 592, 800), ("bb 8->8", 8, 8, 3, 1, 5, 1184, 1600), ("bb 8->16 k5s2", 8, 16, 5, 2, 5, 1184, 1600), ("bb 16->16", 16, 16, 3, 1, 5, 592, 800), ("bb 16->32 k5s2", 16, 32, 5, 2, 5, 592, 800),
                                     ("bb 32->32", 32, 32, 3, 1, 5, 296, 400), ("bb 32->64 k5s2", 32, 64, 5, 2, 5, 296, 400), ("bb 64->64", 64, 64, 3, 1, 5, 148, 200)]:
    x = torch.randn(b, h, w, ci, device=dev)
    wp = ops.pack_conv2d_weight(torch.randn(co, ci, k, k, device=dev) * 0.1)
    for ev in ("0",):
        os.environ["MDF_CONV2D_PREFETCH_EARLY"] = ev
        for _ in range(2): ops.conv2d_nhwc(x, wp, ci, co, k, st)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.mdf_debug_read_stamps(buf, 1)
        n = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): ops.conv2d_nhwc(x, wp, ci, co, k, st)
        e1.record(); torch.cuda.synchronize()
        lib.mdf_debug_read_stamps(buf, 1)
        sched, pro, comp, fill, tot, blocks, items, tiles = [float(v) for v in buf]
        print(f"{name:15s} early={ev} {e0.elapsed_time(e1)/n*1e3:7.1f} us | blocks/launch {blocks/n:6.0f} total {tot/blocks:9.0f} cyc: prologue {pro/tot:5.1%} compute {comp/tot:5.1%} "
              f"refill {fill/tot:5.1%} | tiles/block {tiles/blocks:6.1f}, cyc/tile compute {comp/tiles:6.0f} refill {fill/tiles:6.0f}")
