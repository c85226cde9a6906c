"""Would F(4x4, 3x3) in fp32 (2.25 multiplies per output instead of 4 for F(2x2,3x3), 9 direct) stay inside the error budget?
CPU emulation in the oracle: F(4,3) in (h,w), direct in d, on the 3-D stride-1 layers.  CPU only."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd', R + '/tests']
import numpy as np, torch
import torch.nn.functional as F
from oracle import mvs_oracle as O
from mdfnet_hip import synth
from modelutil import build_model

# Lavin & Gray F(4,3): points 0, +-1, +-2, inf
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)

def wino43(x, w):          # x [b,c,d,h,w], w [o,c,3,3,3]: F(4,3) in h,w per kd, summed over kd
    b, c, d, h, wd = x.shape
    ph, pw = (-h) % 4, (-wd) % 4
    xp = F.pad(x, (1, 1 + pw, 1, 1 + ph, 1, 1))
    out = None
    for kd in range(3):
        xs = xp[:, :, kd:kd + d]
        t = xs.unfold(3, 6, 4).unfold(4, 6, 4)                                  # [b,c,d,nh,nw,6,6]
        V = torch.einsum('ax,by,ncdhwxy->ncdhwab', BT, BT, t)
        U = torch.einsum('ax,by,oixy->oiab', G, G, w[:, :, kd])
        M = torch.einsum('ncdhwab,ocab->nodhwab', V, U)
        Y = torch.einsum('pa,qb,nodhwab->nodhwpq', AT, AT, M)                    # [b,o,d,nh,nw,4,4]
        nh, nw = Y.shape[3:5]
        Y = Y.permute(0, 1, 2, 3, 5, 4, 6).reshape(b, w.shape[0], d, 4 * nh, 4 * nw)[:, :, :, :h, :wd]
        out = Y if out is None else out + Y
    return out

real = F.conv3d
state = {"on": False, "n": 0}
def emu(x, w, bias=None, stride=1, padding=0, *a, **k):
    st = stride if isinstance(stride, int) else stride[0]
    pa = padding if isinstance(padding, int) else padding[0]
    if state["on"] and st == 1 and pa == 1 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[0] >= 16 and w.shape[1] >= 16:
        state["n"] += 1
        y = wino43(x, w)
        return y if bias is None else y + bias.view(1, -1, 1, 1, 1)
    return real(x, w, bias, stride, padding, *a, **k)
O.F.conv3d = emu
torch.manual_seed(0)
x = torch.randn(1, 32, 5, 13, 10); w = torch.randn(16, 32, 3, 3, 3) / np.sqrt(27 * 32)
ref64 = real(x.double(), w.double(), None, 1, 1)
print(f"single layer 32->16: max abs error vs fp64  direct {(real(x, w, None, 1, 1).double() - ref64).abs().max():.2e}   F(4,3) {(wino43(x, w).double() - ref64).abs().max():.2e}")
m = build_model()
sd = synth.seeded_state_dict(m.state_dict(), seed=1)
for (wd, h, v) in ((160, 128, 3), (320, 256, 5)):
    scene = synth.make_scene(wd, h, v, rot_deg=2.0, seed=7)
    state["on"] = False
    ref = O.core_forward(sd, *scene)["depth"].numpy()
    state.update(on=True, n=0)
    d = O.core_forward(sd, *scene)["depth"].numpy()
    e = np.abs(d - ref)
    print(f"{wd}x{h}x{v}  F(4,3) on the Cin,Cout >= 16 stride-1 3-D layers [{state['n']} convs]: mean|d depth| {e.mean():.3e} mm  max {e.max():.3e}  p99 {np.quantile(e, 0.99):.3e}", flush=True)
