"""Regulariser stage s, training forward + backward: fused BatchNorm sums vs separate passes vs float64 CPU autograd. dev tool"""
import os, sys, copy
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd', R + '/tests']
import torch
from mdfnet_hip import train_ops, ops
from modelutil import build_model
DEV = 'cuda:0'
def l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))
for stage in (0, 1, 2):
    torch.manual_seed(21 + stage)
    m = build_model()
    reg_ref = m.Regular[stage].train()
    reg64 = copy.deepcopy(reg_ref).double().train()
    reg = copy.deepcopy(reg_ref).to(DEV)
    g, d, h, w = (((32, 48, 12, 20), (16, 24, 24, 40), (8, 8, 48, 56)) if not os.environ.get('BIG') else ((32, 48, 24, 40), (16, 24, 48, 80), (8, 8, 96, 112)))[stage]
    torch.manual_seed(stage)
    cost = torch.rand(2, g, d, h, w)
    hyp = (425 + 510 * torch.rand(2, 1, h, w)) + torch.linspace(-20, 20, d).reshape(1, d, 1, 1)
    dd = torch.randn(2, h, w)
    c64 = cost.double().requires_grad_(True)
    p64, d64 = reg64(c64, hyp.double())
    d64.backward(dd.double())
    for fused in (True, False):
        train_ops.FUSE_BN_SUMS = fused
        c = cost.to(DEV).requires_grad_(True)
        prob, depth = reg(c, hyp.to(DEV))
        depth.backward(dd.to(DEV))
        worst = max(l2(p.grad, q.grad) for p, q in zip(reg.parameters(), reg64.parameters()))
        print(f"stage {stage} fused={fused}: prob {l2(prob, p64):.2e} depth {l2(depth, d64):.2e} dcost {l2(c.grad, c64.grad):.2e} worst param grad {worst:.2e}", flush=True)
        reg.zero_grad()
