"""One regulariser conv layer, a few launches: the target of `rocprofv3 --pmc ... -- python3 scripts/pmc_layer.py CIN COUT D H W` (dev tool)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
ci, co, D, H, W = (int(a) for a in sys.argv[1:6])
x = torch.randn(1, D, H, W, ci, device="cuda:0")
wp = ops.pack_conv3d_weight(torch.randn(co, ci, 3, 3, 3, device="cuda:0") * 0.05)
for _ in range(5):
    ops.conv3d_ndhwc(x, wp, ci, co)
torch.cuda.synchronize()
