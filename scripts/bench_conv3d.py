"""Per-layer microbenchmark of the regulariser conv layers at BASELINE config-2 shapes (dev tool).
usage: python scripts/bench_conv3d.py [--check]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops

# (name, cin, cout, mode, D, H, W) at cfg2; mode s1|s2|tr ; dims are INPUT dims
L = [("s0.conv01.0", 32, 16, "s1", 48, 148, 200), ("s0.conv01.1", 16, 16, "s1", 48, 148, 200),
     ("s0.conv12.0", 16, 32, "s2", 48, 148, 200), ("s0.conv12.1", 32, 32, "s1", 24, 74, 100),
     ("s0.conv232.0", 32, 64, "s2", 24, 74, 100), ("s0.conv232.1", 64, 64, "s1", 12, 37, 50),
     ("s0.conv232.3T", 64, 32, "tr", 12, 37, 50), ("s0.conv10T", 32, 16, "tr", 24, 74, 100),
     ("s1.conv01", 16, 8, "s1", 24, 296, 400), ("s1.conv12.0", 8, 16, "s2", 24, 296, 400),
     ("s1.conv12.1", 16, 16, "s1", 12, 148, 200), ("s1.conv23.1", 32, 32, "s1", 6, 74, 100),
     ("s1.trconv21T", 16, 8, "tr", 12, 148, 200),
     ("s2.conv01", 8, 8, "s1", 8, 592, 800), ("s2.conv12.0", 8, 16, "s2", 8, 592, 800), ("s2.conv12.1", 16, 16, "s1", 4, 296, 400),
     ("s2.trconv21T", 16, 8, "tr", 4, 296, 400)]
if "--tr" in sys.argv:
    L = [l for l in L if l[3] == "tr"]
if "--small" in sys.argv:     # the small-volume levels of the three U-Nets (split-K / one-tile kernels)
    L = [("s0.conv232.1", 64, 64, "s1", 12, 37, 50), ("s0.conv232.0", 32, 64, "s2", 24, 74, 100), ("s1.conv23.1", 32, 32, "s1", 6, 74, 100),
         ("s1.conv34.0", 32, 64, "s2", 6, 74, 100), ("s1.conv34.1", 64, 64, "s1", 3, 37, 50), ("s2.conv23.1", 32, 32, "s1", 2, 148, 200),
         ("s2.conv34.0", 32, 64, "s2", 2, 148, 200), ("s2.conv34.1", 64, 64, "s1", 1, 74, 100),
         ("s0.conv12.0", 16, 32, "s2", 48, 148, 200), ("s1.conv23.0", 16, 32, "s2", 12, 148, 200), ("s2.conv23.0", 16, 32, "s2", 4, 296, 400)]
if "--smalltr" in sys.argv:   # the shallow transposed layers (stage 2's innermost levels)
    L = [("s2.tr43", 64, 32, "tr", 1, 74, 100), ("s2.tr32", 32, 16, "tr", 2, 148, 200), ("s1.tr43", 64, 32, "tr", 3, 37, 50),
         ("s1.tr32", 32, 16, "tr", 6, 74, 100), ("s0.tr32", 64, 32, "tr", 12, 37, 50)]
if "--s2" in sys.argv:
    L = [l for l in L if l[3] == "s2"]
dev = "cuda:0"
check = "--check" in sys.argv
tot_ms = tot_fl = 0.0
for name, ci, co, mode, D, H, W in L:
    tr = mode == "tr"
    x = torch.randn(1, D, H, W, ci, device=dev)
    wt = torch.randn(*((ci, co) if tr else (co, ci)), 3, 3, 3, device=dev) / (27 * ci) ** 0.5
    if "--zeros" in sys.argv:      # clock/power experiment: same instruction stream on all-zero operands
        x.zero_(); wt.zero_()
    wp = ops.pack_conv3d_weight(wt, tr)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    st = 1 if mode == "s1" else 2
    # --res: the transposed layers with their skip tensor, as in the regularisers (x1 + relu(bn(convT(x2))), regular.py:67)
    skip = torch.randn(1, 2 * D, 2 * H, 2 * W, co, device=dev) if (tr and "--res" in sys.argv) else None
    y = ops.conv3d_ndhwc(x, wp, ci, co, st, tr, al, be, True, skip)
    if check:
        import torch.nn.functional as F
        xc = x.permute(0, 4, 1, 2, 3)
        ref = F.conv_transpose3d(xc, wt, None, 2, 1, 1) if tr else F.conv3d(xc, wt, None, st, 1)
        ref = F.relu(ref * al.view(1, -1, 1, 1, 1) + be.view(1, -1, 1, 1, 1)).permute(0, 2, 3, 4, 1)
        if skip is not None:
            ref = ref + skip
        err = (y - ref).abs().max().item()
    torch.cuda.synchronize()
    n = 10
    def timeit():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.conv3d_ndhwc(x, wp, ci, co, st, tr, al, be, True, skip)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ab = os.environ.get("MDF_AB")   # e.g. MDF_AB=MDF_CONV_ITEMS_PER_BLOCK MDF_AB_VALS=2,3,6 : interleaved in ONE process
    if ab:
        vals = os.environ.get("MDF_AB_VALS", "0,1").split(",")
        res = {v: [] for v in vals}
        for _ in range(4):
            for v in vals:
                os.environ[ab] = v; res[v].append(timeit())
        ms = min(res[vals[0]])
        print(f"   A/B {ab}: " + "  ".join(f"{v} -> {min(t)*1e3:.1f} us" for v, t in res.items()))
    else:
        ms = timeit()
    nvox = D * H * W if tr else y.shape[1] * y.shape[2] * y.shape[3]
    fl = 2.0 * 27 * ci * co * nvox
    tot_ms += ms; tot_fl += fl
    print(f"{name:14s} {ci:2d}->{co:2d} {mode} {D:3d}x{H:3d}x{W:3d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TFLOP/s" + (f"  maxerr {err:.2e}" if check else ""))
print(f"sum {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.2f} TFLOP/s")
