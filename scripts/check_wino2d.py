"""wino2d.hip against conv_lds.hip's 2-D Winograd form (bit for bit), against torch, and timed at the cfg2 shapes (dev check)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
import torch.nn.functional as F
from mdfnet_hip import ops
dev = "cuda:0"
shapes = [(16, 16, 1, 130, 201), (32, 32, 2, 125, 131), (16, 16, 1, 8, 32), (32, 32, 1, 3, 5), (16, 16, 3, 17, 40), (32, 32, 1, 61, 700), (16, 16, 5, 592, 800), (32, 32, 5, 296, 400),
          (64, 64, 1, 37, 50), (64, 64, 2, 9, 131), (64, 64, 5, 148, 200), (64, 64, 1, 3, 5)]
ok = True
for ci, co, b, h, w in shapes:
    g = torch.Generator().manual_seed(ci * 100 + co + w)
    x = torch.randn(b, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5
    al, be = torch.rand(co, generator=g) + 0.5, torch.rand(co, generator=g) * 0.4 - 0.2
    xd = ops.to_nhwc(x.to(dev)); wp = ops.pack_conv2d_weight(wt.to(dev))
    res = torch.randn(b, h, w, co, device=dev)
    outs = {}
    for v in ("0", "1"):
        os.environ["MDF_CONV_WINO2D"] = v
        outs[v] = (ops.conv2d_nhwc(xd, wp, ci, co, 3, 1, al.to(dev), be.to(dev), True, res, 0.1), ops.conv2d_nhwc(xd, wp, ci, co, 3, 1))
    torch.cuda.synchronize()
    eq = torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    ref = F.conv2d(x, wt, None, 1, 1).permute(0, 2, 3, 1)
    err = (outs["1"][1].cpu() - ref).abs().max().item()
    print(f"{ci}->{co} {b}x{h}x{w}: equal to the conv_lds form: {eq} (max diff {(outs['0'][0] - outs['1'][0]).abs().max().item():.3e}); max err vs torch {err:.3e}", flush=True)
    ok &= eq and err < 5e-5
for ci, co, b, h, w in [(16, 16, 5, 592, 800), (32, 32, 5, 296, 400), (64, 64, 5, 148, 200)]:
    x = torch.randn(b, h, w, ci, device=dev)
    wp = ops.pack_conv2d_weight(torch.randn(co, ci, 3, 3, device=dev) / (9 * ci) ** 0.5)
    al, be = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev) * 0.1
    for v in ("0", "1", "0", "1"):
        os.environ["MDF_CONV_WINO2D"] = v
        for _ in range(3): ops.conv2d_nhwc(x, wp, ci, co, 3, 1, al, be, True)
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.conv2d_nhwc(x, wp, ci, co, 3, 1, al, be, True)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{ci}->{co} {b}x{h}x{w} MDF_CONV_WINO2D={v}: {best*1e3:7.1f} us", flush=True)
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
