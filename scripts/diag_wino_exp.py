"""Dev experiment: build conv_lds variants with -DMDF_WG_EXP=n (1: no transform, 2: one LDS read instead of 16, 3: 1/16 of the
weight loads; results are WRONG by design) and time the Winograd layers -- which part of the step limits it?"""
import ctypes, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
csrc = R + '/mdf-net_amd/csrc'
import torch
import mdfnet_hip
from mdfnet_hip import ops
srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.cpp'))]
exp = int(sys.argv[1])
so = '/tmp/libmdfnet_exp%d.so' % exp
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '--offload-arch=gfx950', '-DMDF_WG_EXP=%d' % exp,
                '-I', R + '/include', '-I', csrc, '-x', 'hip'] + srcs + ['-o', so], check=True, stderr=subprocess.DEVNULL)
mdfnet_hip.LIB_PATH = so
dev = 'cuda:0'
for name, ci, co, D, H, W in [("32->16", 32, 16, 48, 148, 200), ("16->16", 16, 16, 48, 148, 200), ("32->32", 32, 32, 24, 74, 100)]:
    x = torch.randn(1, D, H, W, ci, device=dev)
    wp = ops.pack_conv3d_weight(torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05)
    for _ in range(3): ops.conv3d_ndhwc(x, wp, ci, co)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv3d_ndhwc(x, wp, ci, co)
    e1.record(); torch.cuda.synchronize()
    print(f"exp {exp}  {name}: {e0.elapsed_time(e1)/10*1e3:7.1f} us", flush=True)
