"""Aggregation kernels at the three cfg2 stage shapes (and cfg4's), HIP-event timed, synthetic DTU-like cameras.
MDF_WARP_WINDOW=0|1 selects the plain / LDS-window kernel (read once per process).  dev tool"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from mdfnet_hip import ops, synth
from net.unit.scale import scale_cam
dev = torch.device('cuda', 0)
torch.manual_seed(0)
W, H, V = (int(x) for x in os.environ.get("MDF_SHAPE", "1600,1184,5").split(","))
intr, extr, dr = synth.make_cameras(W, H, V, batch=1, rot_deg=float(os.environ.get("MDF_ROT", "3.0")), seed=101)
tot = 0.0
for stage, (c, g, d) in enumerate(((64, 32, 48), (32, 16, 24), (16, 8, 8))):
    h, w = H >> (3 - stage), W >> (3 - stage)
    rp, sps = scale_cam(intr, extr, stage)
    proj = ops.relative_projections(rp, list(sps)).to(dev)
    feats = [torch.randn(1, c, h, w, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(V)]
    if stage == 0:
        hyp = torch.linspace(425, 935, d, device=dev).reshape(1, d, 1, 1)
    else:
        span = 40.0 if stage == 1 else 6.0      # per-pixel hypotheses around a smooth depth map, as the cascade produces
        base = 600 + 80 * torch.sin(torch.linspace(0, 6, w, device=dev)).reshape(1, 1, 1, w) + torch.zeros(1, 1, h, w, device=dev)
        hyp = (base + torch.linspace(-span, span, d, device=dev).reshape(1, d, 1, 1)).contiguous()
    wpar = torch.randn(g + 4, device=dev)
    for _ in range(3):
        cost = ops.warp_aggregate_vec(feats, proj, hyp, wpar, g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        cost = ops.warp_aggregate_vec(feats, proj, hyp, wpar, g)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = 4.0 * (V * c * h * w + hyp.numel() + g * d * h * w)
    tot += ms
    print(f"stage {stage} C{c} D{d} {w}x{h}: {ms*1e3:8.1f} us  {by/ms/1e6:7.0f} GB/s  checksum {float(cost.double().sum()):.6f}", flush=True)
print(f"total {tot*1e3:.1f} us per view (window={os.environ.get('MDF_WARP_WINDOW', '1')})")
