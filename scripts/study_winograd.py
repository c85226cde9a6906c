"""Error budget for Winograd F(2x2x2, 3x3x3) in fp32 on the regulariser's stride-1 3-D convs: 8 instead of
27 multiplies per output and channel pair (3.4x fewer MFMA flops).  Emulated in the CPU oracle, end-to-end depth deviation
from the direct-conv oracle.  CPU only."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd', R + '/tests']
import numpy as np, torch
import torch.nn.functional as F
from oracle import mvs_oracle as O
from mdfnet_hip import synth
from modelutil import build_model

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

def wino3d(x, w):
    b, c, d, h, wd = x.shape
    pd, ph, pw = d % 2, h % 2, wd % 2
    xp = F.pad(x, (1, 1 + pw, 1, 1 + ph, 1, 1 + pd))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)                       # [b,c,nd,nh,nw,4,4,4]
    V = torch.einsum('ax,by,ez,ncdhwxyz->ncdhwabe', BT, BT, BT, t)
    U = torch.einsum('ax,by,ez,oixyz->oiabe', G, G, G, w)
    M = torch.einsum('ncdhwabe,ocabe->nodhwabe', V, U)
    Y = torch.einsum('pa,qb,re,nodhwabe->nodhwpqr', AT, AT, AT, M)              # [b,o,nd,nh,nw,2,2,2]
    nd, nh, nw = Y.shape[2:5]
    Y = Y.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(b, w.shape[0], 2 * nd, 2 * nh, 2 * nw)
    return Y[:, :, :d, :h, :wd]

real = F.conv3d
state = {"on": False, "n": 0}
def emu(x, w, bias=None, stride=1, padding=0, *a, **k):
    st = stride if isinstance(stride, int) else stride[0]
    pa = padding if isinstance(padding, int) else padding[0]
    if state["on"] and st == 1 and pa == 1 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[0] >= state["min_cout"]:
        state["n"] += 1
        y = wino3d(x, w)
        return y if bias is None else y + bias.view(1, -1, 1, 1, 1)
    return real(x, w, bias, stride, padding, *a, **k)
O.F.conv3d = emu

# single-layer check
torch.manual_seed(0)
x = torch.randn(1, 16, 9, 13, 10); w = torch.randn(16, 16, 3, 3, 3) / np.sqrt(27 * 16)
ref64 = real(x.double(), w.double(), None, 1, 1)
e_dir = (real(x, w, None, 1, 1).double() - ref64).abs().max().item(); e_win = (wino3d(x, w).double() - ref64).abs().max().item()
print(f"single layer 16->16: max abs error vs fp64  direct {e_dir:.2e}   winograd {e_win:.2e}  (|y| ~ {ref64.abs().mean():.2f})")

m = build_model()
sd = synth.seeded_state_dict(m.state_dict(), seed=1)
for (wd, h, v) in ((160, 128, 3), (320, 256, 5)):
    scene = synth.make_scene(wd, h, v, rot_deg=2.0, seed=7)
    state["on"] = False
    ref = O.core_forward(sd, *scene)["depth"].numpy()
    for min_cout, label in ((8, "all stride-1 3x3x3 layers (incl. Cout 8), prob conv excluded"), (1, "all incl. the prob conv")):
        state.update(on=True, n=0, min_cout=min_cout)
        d = O.core_forward(sd, *scene)["depth"].numpy()
        e = np.abs(d - ref)
        print(f"{wd}x{h}x{v}  Winograd F(2,3)^3 on {label} [{state['n']} convs]: mean|d depth| {e.mean():.3e} mm  max {e.max():.3e}  p99 {np.quantile(e, 0.99):.3e}", flush=True)
