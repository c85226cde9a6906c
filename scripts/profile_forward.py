"""Host-side issue time per phase of the forward (no syncs inside), then a py-spy-like wall profile.  dev tool"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ops, hostmirror
dev = torch.device('cuda', 0)
model = bench.build(dev)
inputs = tuple(t.to(dev) for t in synth.make_scene(bench.WIDTH, bench.HEIGHT, bench.VIEWS, batch=1, rot_deg=3.0, seed=100))
marks = []
def hook(name):
    def pre(m, i): marks.append((name + ':in', time.perf_counter()))
    def post(m, i, o): marks.append((name + ':out', time.perf_counter()))
    return pre, post
for name, mod in [('Backbone', model.Backbone), ('Refine', model.Refine)] + [(f'Hypos{s}', model.Depth_hypos[s]) for s in range(3)] + \
        [(f'Aggre{s}', model.Homoaggre[s]) for s in range(3)] + [(f'Regular{s}', model.Regular[s]) for s in range(3)]:
    pre, post = hook(name)
    mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
_orig = ops._abi
abi_time = [0.0, 0]
def timed_abi(name, args, tag="", work=None):
    t = time.perf_counter(); _orig(name, args, tag, work); abi_time[0] += time.perf_counter() - t; abi_time[1] += 1
ops._abi = timed_abi
with torch.no_grad():
    for _ in range(3): model(*inputs)
    torch.cuda.synchronize()
    for rep in range(2):
        marks.clear(); abi_time[0] = 0; abi_time[1] = 0
        t0 = time.perf_counter()
        out = model(*inputs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"--- forward issue {1e3*(t1-t0):.2f} ms, +sync {1e3*(t2-t1):.2f} ms; C-ABI calls: {abi_time[1]} taking {1e3*abi_time[0]:.2f} ms on the host")
        prev = t0
        for name, t in marks:
            print(f"   {name:14s} +{1e3*(t-prev):7.2f} ms")
            prev = t
        print(f"   end            +{1e3*(t1-prev):7.2f} ms")
