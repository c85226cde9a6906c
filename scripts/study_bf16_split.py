"""Error budget for fp32-exact GEMMs on the bf16 matrix cores: emulate the split-bf16 products in the CPU
oracle's conv layers and measure the end-to-end depth deviation from the plain fp32 oracle.  CPU only (runs anywhere).
a = a1 + a2 + a3 (bf16 each); variant xN keeps the N leading cross products a_i*b_j (i+j smallest first)."""
import os, sys, itertools
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import numpy as np, torch
import torch.nn.functional as F
from oracle import mvs_oracle as O
from mdfnet_hip import synth
sys.path.insert(0, R + "/tests")
from modelutil import build_model

def split3(x):
    x1 = x.bfloat16().float(); r = x - x1
    x2 = r.bfloat16().float(); x3 = (r - x2).bfloat16().float()
    return [x1, x2, x3]

ORDER = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0), (1, 2), (2, 1), (2, 2)]
real = {"conv2d": F.conv2d, "conv3d": F.conv3d, "conv_transpose3d": F.conv_transpose3d}
state = {"n": 0, "which": ("conv3d", "conv_transpose3d", "conv2d")}

def make(name):
    fn = real[name]
    def emu(x, w, bias=None, *a, **k):
        if state["n"] == 0 or name not in state["which"] or w.shape[0] * w.shape[1] < 64:   # leave the tiny 1-channel convs alone
            return fn(x, w, bias, *a, **k)
        xs, ws = split3(x), split3(w)
        out = None
        for i, j in ORDER[:state["n"]]:
            t = fn(xs[i], ws[j], None, *a, **k)
            out = t if out is None else out + t
        if bias is not None:
            out = out + bias.view(1, -1, *([1] * (out.dim() - 2)))
        return out
    return emu

for n in real: setattr(O.F, n, make(n))
torch.manual_seed(0)
m = build_model()
sd = synth.seeded_state_dict(m.state_dict(), seed=1)
for (w, h, v) in ((160, 128, 3), (320, 256, 5)):
    scene = synth.make_scene(w, h, v, rot_deg=2.0, seed=7)
    state["n"] = 0
    ref = O.core_forward(sd, *scene)["depth"].numpy()
    for which in (("conv3d", "conv_transpose3d"), ("conv3d", "conv_transpose3d", "conv2d")):
        for n in (3, 4, 6):
            state["n"], state["which"] = n, which
            d = O.core_forward(sd, *scene)["depth"].numpy()
            e = np.abs(d - ref)
            print(f"{w}x{h}x{v}  split-bf16 x{n} in {'+'.join(which):34s}: mean|d depth| {e.mean():.3e} mm  max {e.max():.3e}  p99 {np.quantile(e, 0.99):.3e}", flush=True)
