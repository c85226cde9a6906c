"""The prob heads of cfg2's three stages: us per launch under the dev switches in the environment.  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops, lib
dev = 'cuda:0'
for c, d, h, w in [(8, 8, 592, 800), (8, 24, 296, 400), (16, 48, 148, 200)]:
    x = torch.randn(1, d, h, w, c, device=dev)
    wt = torch.randn(1, c, 3, 3, 3, device=dev) * 0.3
    hyp = (torch.rand(1, d, h, w, device=dev) * 500 + 400) if c == 8 else (torch.rand(1, d, 1, 1, device=dev) * 500 + 400)
    wp = ops.pack_prob_weight(wt)
    for _ in range(3): ops.prob_head(x, wt, hyp, wpack=wp)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.prob_head(x, wt, hyp, wpack=wp)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{c}->1 {d}x{h}x{w}: {best:7.1f} us  [{lib().mdf_last_launch().decode()[:40]}]")
