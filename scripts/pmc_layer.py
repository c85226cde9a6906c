"""One regulariser conv layer, a few launches: the target of `rocprofv3 --pmc ... -- python3 scripts/pmc_layer.py CIN COUT D H W [s1|s2|tr]`
(dev tool; transposed layers run with their skip tensor and folded BatchNorm + ReLU, as in the regularisers)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
ci, co, D, H, W = (int(a) for a in sys.argv[1:6])
x = torch.randn(1, D, H, W, ci, device="cuda:0")
mode = sys.argv[6] if len(sys.argv) > 6 else "s1"
tr = mode == "tr"
wp = ops.pack_conv3d_weight(torch.randn(*((ci, co) if tr else (co, ci)), 3, 3, 3, device="cuda:0") * 0.05, tr)
al, be = torch.rand(co, device="cuda:0") + 0.5, torch.randn(co, device="cuda:0") * 0.1
skip = torch.randn(1, 2 * D, 2 * H, 2 * W, co, device="cuda:0") if tr else None
for _ in range(5):
    ops.conv3d_ndhwc(x, wp, ci, co, 1 if mode == "s1" else 2, tr, al, be, True, skip)
torch.cuda.synchronize()
