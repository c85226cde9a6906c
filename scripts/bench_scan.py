"""Whole-scan throughput: 49 reference views x 5-view items, with and without the feature cache.
Items/s over the model only (inputs resident, no file IO).  python scripts/bench_scan.py [--views 49] [--w 1600 --h 1184]"""
import argparse, os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "mdf-net_amd"))
from config import build_model
from mdfnet_hip import synth, hostmirror

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=49); ap.add_argument("--w", type=int, default=1600); ap.add_argument("--h", type=int, default=1184)
a = ap.parse_args()
dev = torch.device("cuda", 0)
m = build_model(); m.load_state_dict(synth.seeded_state_dict(m.state_dict(), 29)); m.eval().to(dev)
V = a.views
imgs_all, ext_all, intr_all, dr = synth.make_scene(a.w, a.h, V, seed=5)
imgs_all = imgs_all.to(dev)
rng = np.random.default_rng(0)
items = [[r] + [int(x) for x in rng.permutation([v for v in range(V) if v != r])[:4]] for r in range(V)]

def run(cache):
    fc = {} if cache else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in items:
        e = ext_all[:, it].clone(); k = intr_all[:, it].clone(); d = dr.clone()
        args = (imgs_all[:, it], e.to(dev), k.to(dev), d.to(dev))
        for g, c in zip(args[1:], (e, k, d)): hostmirror.put(g, c)
        with torch.no_grad():
            if cache: m(*args, feature_cache=fc, view_keys=it)
            else: m(*args)
    torch.cuda.synchronize()
    return len(items) / (time.perf_counter() - t0)

run(False)
for c in (False, True, False, True):
    print("cache=%d  %.2f reference views (items)/s" % (c, run(c)), flush=True)
