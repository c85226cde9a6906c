"""How far is the reference algorithm from ITSELF across x86 hosts?  Runs the CPU oracle (same ATen calls as the
reference) on this host and compares with the goldens produced by the real reference in the build container."""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd', R + '/tests']
from mdfnet_hip import synth
from oracle import mvs_oracle as O
T = torch.from_numpy
meta = np.load(R + '/tests/golden/state_dict_meta.npz')
shapes = {}
for k, s, dt in zip(meta['keys'], meta['shapes'], meta['dtypes']):
    shape = tuple(int(x) for x in s.strip('[]').split(',') if x.strip())
    shapes[str(k)] = torch.empty(shape, dtype=torch.int64 if 'int64' in str(dt) else torch.float32)
sd = synth.seeded_state_dict(shapes, seed=1)
os.system("lscpu | grep -i 'model name'")
for name in ('e2e_tiny.npz', 'e2e_cfg1.npz', 'e2e_5view.npz'):
    g = dict(np.load(R + '/tests/golden/' + name))
    w, h, v, b, rot, seed = g['cfg']
    imgs, extr, intr, dr = synth.make_scene(int(w), int(h), int(v), batch=int(b), rot_deg=float(rot), seed=int(seed))
    out, tr = O.core_forward(sd, imgs, extr, intr, dr, keep=True)
    e = (out['depth'] - T(g['depth'])).abs()
    msg = f"{name}: oracle(this host) vs reference(build host): mean|d depth| {e.mean():.3e} max {e.max():.3e}"
    if 'hypos1' in g:
        msg += f" | hyp1 mean {(tr['hypos1']-T(g['hypos1'])).abs().mean():.3e} hyp2 mean {(tr['hypos2']-T(g['hypos2'])).abs().mean():.3e}"
    print(msg)
g = dict(np.load(R + '/tests/golden/ops.npz'))
dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=4.0, seed=5)[3]
hyp0 = O.uniform_hypos(dr, 48)
s = O.gauss1_fit(T(g['reg0_prob']), hyp0)
print('gauss1_fit on golden prob: mismatches', int((s != T(g['hyp1_s'])).sum()), 'of', s.numel(), 'max rel', float(((s - T(g['hyp1_s'])).abs() / T(g['hyp1_s'])).max()))
