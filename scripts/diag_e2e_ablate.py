"""Ablation (dev diagnostic): inject reference-exact tensors at successive points of the product forward and
watch the stage-1 hypothesis error / final depth error.  Usage: python scripts/diag_e2e_ablate.py"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd', R + '/tests']
from mdfnet_hip import synth, ops
from oracle import mvs_oracle as O
from modelutil import build_model
T = torch.from_numpy
g = dict(np.load(R + '/tests/golden/e2e_tiny.npz'))
meta = np.load(R + '/tests/golden/state_dict_meta.npz')
shapes = {}
for k, s, dt in zip(meta['keys'], meta['shapes'], meta['dtypes']):
    shape = tuple(int(x) for x in s.strip('[]').split(',') if x.strip())
    shapes[str(k)] = torch.empty(shape, dtype=torch.int64 if 'int64' in str(dt) else torch.float32)
sd = synth.seeded_state_dict(shapes, seed=1)
m = build_model(); m.load_state_dict(sd); m.eval().cuda()
w, h, v, b, rot, seed = g['cfg']
imgs, extr, intr, dr = synth.make_scene(int(w), int(h), int(v), batch=int(b), rot_deg=float(rot), seed=int(seed))
bb = {k[9:]: v_ for k, v_ in sd.items() if k.startswith('Backbone.')}
cpu_feats = [O.fpn_4scales(imgs[:, i], bb) for i in range(int(v))]

def run(inject):
    hooks = []
    tr = {}
    if 'feat' in inject:
        it = iter(cpu_feats)
        hooks.append(m.Backbone.register_forward_hook(lambda mod, i, o: tuple(t.cuda() for t in next(it))))
    for st in range(3):
        if f'cost{st}' in inject:
            hooks.append(m.Homoaggre[st].register_forward_hook(lambda mod, i, o, st=st: ops.from_ndhwc(ops.to_ndhwc(T(g[f'cost{st}']).cuda()))))
        if f'prob{st}' in inject:
            hooks.append(m.Regular[st].register_forward_hook(lambda mod, i, o, st=st: T(g[f'prob{st}']).cuda()))
        hooks.append(m.Depth_hypos[st].register_forward_hook(lambda mod, i, o, st=st: tr.__setitem__(st, o)))
    with torch.no_grad():
        out = m(imgs.cuda(), extr.cuda(), intr.cuda(), dr.cuda())
    for hk in hooks: hk.remove()
    e1 = (tr[1].cpu() - T(g['hypos1'])).abs().mean().item()
    e2 = (tr[2].cpu() - T(g['hypos2'])).abs().mean().item()
    ef = (out['depth'].cpu() - T(g['depth'])).abs().mean().item()
    print(f"inject={sorted(inject)!s:50s} hyp1 err {e1:.2e}  hyp2 err {e2:.2e}  final {ef:.2e}")

run(set())
run({'feat'})
run({'feat', 'cost0'})
run({'prob0'})
run({'prob0', 'cost1'})
run({'prob0', 'prob1'})
run({'prob0', 'prob1', 'cost2'})
run({'prob0', 'prob1', 'prob2'})
