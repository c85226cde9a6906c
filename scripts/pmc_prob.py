"""prob head at the stage-1/2 sizes of cfg2, a few launches (target of scripts/pmc_kernels.sh prob_fused scripts/pmc_prob.py).  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops
for (c, d, h, w) in ((8, 8, 592, 800), (8, 24, 296, 400)):
    x = torch.randn(1, d, h, w, c, device="cuda:0")
    wt = torch.randn(1, c, 3, 3, 3, device="cuda:0") * 0.2
    hyp = torch.rand(1, d, h, w, device="cuda:0") * 500 + 400
    wp = ops.pack_prob_weight(wt)
    for _ in range(5):
        ops.prob_head(x, wt, hyp, wpack=wp)
torch.cuda.synchronize()
