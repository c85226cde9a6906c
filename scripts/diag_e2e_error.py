import sys, os, numpy as np, torch
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path[:0]=[R, R+'/mdf-net_amd', R+'/tests']
from mdfnet_hip import synth, ops
from oracle import mvs_oracle as O
from modelutil import build_model
import conftest
g=dict(np.load(R+'/tests/golden/e2e_tiny.npz'))
meta=np.load(R+'/tests/golden/state_dict_meta.npz')
shapes={}
for k,s,dt in zip(meta['keys'],meta['shapes'],meta['dtypes']):
    shape=tuple(int(x) for x in s.strip('[]').split(',') if x.strip())
    shapes[str(k)]=torch.empty(shape,dtype=torch.int64 if 'int64' in str(dt) else torch.float32)
sd=synth.seeded_state_dict(shapes,seed=1)
m=build_model(); m.load_state_dict(sd); m.eval().cuda()
w,h,v,b,rot,seed=g['cfg']
imgs,extr,intr,dr=synth.make_scene(int(w),int(h),int(v),batch=int(b),rot_deg=float(rot),seed=int(seed))
T=torch.from_numpy
# golden-derived stage depths
gd=[O.depth_regression(T(g[f'prob{s}']),T(g[f'hypos{s}'])) for s in range(3)]
tr={}
hk=[m.Regular[s].register_forward_hook(lambda mod,i,o,s=s: tr.__setitem__(s,o)) for s in range(3)]
hh=[m.Depth_hypos[s].register_forward_hook(lambda mod,i,o,s=s: tr.__setitem__(('h',s),o)) for s in range(3)]
with torch.no_grad():
    out=m(imgs.cuda(),extr.cuda(),intr.cuda(),dr.cuda())
    for s in range(3):
        d=ops.depth_regress(tr[s],tr[('h',s)]).cpu()
        print('stage',s,'depth err mean',(d-gd[s]).abs().mean().item(),'max',(d-gd[s]).abs().max().item())
    # refine in isolation: same input (golden depth2) on GPU (MIOpen) vs CPU oracle
    rg=m.Refine(gd[2].cuda(),dr.cuda()).cpu()
    rc=O.refine_net2(gd[2],dr,{k[7:]:v for k,v in sd.items() if k.startswith('Refine.')})
    print('refine GPU-vs-CPU same input: mean',(rg-rc).abs().mean().item(),'max',(rg-rc).abs().max().item())
    print('cpu refine vs golden final', (rc-T(g['depth'])).abs().mean().item())
    # backbone GPU vs CPU
    f_g=m.Backbone(imgs[:,0].cuda())
    f_c=O.fpn_4scales(imgs[:,0],{k[9:]:v for k,v in sd.items() if k.startswith('Backbone.')})
    for a,c in zip(f_g,f_c): print('backbone err max',(a.cpu()-c).abs().max().item(),'scale',c.abs().max().item())
print('final err mean',(out['depth'].cpu()-T(g['depth'])).abs().mean().item())
