"""wino3d.hip against conv_lds.hip's Winograd form (bit for bit) and against torch (dev check; tests/test_regular_gpu.py holds the asserts)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/mdf-net_amd']
import torch
import torch.nn.functional as F
from mdfnet_hip import ops
dev = "cuda:0"
shapes = [(32, 16, 1, 6, 130, 201), (16, 16, 2, 5, 125, 131), (16, 16, 1, 1, 400, 400), (32, 16, 1, 2, 3, 25001), (16, 16, 1, 3, 260, 197),
          (32, 16, 1, 48, 148, 200), (16, 16, 1, 12, 148, 200), (16, 16, 1, 4, 296, 400), (32, 16, 1, 90, 5, 401), (16, 16, 1, 7, 9, 33)]
ok = True
for ci, co, b, d, h, w in shapes:
    g = torch.Generator().manual_seed(ci * 100 + co + w)
    x = torch.randn(b, ci, d, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, 3, generator=g) / (27 * ci) ** 0.5
    al, be = torch.rand(co, generator=g) + 0.5, torch.rand(co, generator=g) * 0.4 - 0.2
    xd = ops.to_ndhwc(x.to(dev)); wp = ops.pack_conv3d_weight(wt.to(dev))
    res = torch.randn(b, d, h, w, co, device=dev)
    os.environ["MDF_CONV_LDS_MIN_VOXELS"] = "0"
    outs = {}
    for v in ("0", "1"):
        os.environ["MDF_CONV_WINO3D"] = v
        outs[v] = (ops.conv3d_ndhwc(xd, wp, ci, co, 1, False, al.to(dev), be.to(dev), True, res), ops.conv3d_ndhwc(xd, wp, ci, co, 1, False))
    torch.cuda.synchronize()
    eq = torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    ref = F.conv3d(x, wt, None, 1, 1).permute(0, 2, 3, 4, 1)
    err = (outs["1"][1].cpu() - ref).abs().max().item()
    d01 = (outs["0"][0] - outs["1"][0]).abs().max().item()
    nan = torch.isnan(outs["1"][0]).any().item()
    print(f"{ci}->{co} {b}x{d}x{h}x{w}: equal to the conv_lds form: {eq} (max diff {d01:.3e}, nan {nan}); max err vs torch {err:.3e}")
    ok &= eq and err < 5e-5
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
