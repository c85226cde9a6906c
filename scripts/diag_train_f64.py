"""dev: where does the HIP training step's distance from float64 come from?  Tiny training golden step (96x64x3, batch 2):
per-tensor L2 error vs the float64 oracle for (a) the full HIP path, (b) the HIP path with ONE slot family routed through torch
autograd (stock ops on the GPU), next to the fp32 CPU oracle's own error.  usage (GPU box): python scripts/diag_train_f64.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "mdf-net_amd"), os.path.join(R, "tests")]
import numpy as np, torch
from mdfnet_hip import layers, synth
from modelutil import build_model
from net.loss import Loss
from oracle import gen_golden, mvs_oracle as O

DEV = "cuda:0"
g = dict(np.load(os.path.join(R, "tests/golden/train_tiny.npz")))
m0 = build_model()
sd = synth.seeded_state_dict(m0.state_dict(), seed=1)
gtn = {k: g["gt" + k] for k in ("3", "2", "1", "0")}
_, d64, g64 = gen_golden.train_f64(sd, gtn)
imgs, extr, intr, dr = synth.make_scene(96, 64, 3, batch=2, rot_deg=3.0, seed=31)
sdo = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in sd.items()}
out_ref = O.core_forward(sdo, imgs, extr, intr, dr, training=True)
O.mvs_loss(out_ref["depth"], {k: torch.from_numpy(v) for k, v in gtn.items()}, dr).backward()


def l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


ref_err = {k: l2(sdo[k].grad.numpy(), g64[k]) for k in g64}
orig = layers.hip_train


def run(stock_classes):
    def patched(mod, *ts):
        if mod is not None and type(mod).__name__ in stock_classes:
            return False
        return orig(mod, *ts)
    layers.hip_train = patched
    try:
        m = build_model(); m.load_state_dict(sd); m.train().to(DEV)
        out = m(imgs.to(DEV), extr.to(DEV), intr.to(DEV), dr.to(DEV))
        loss = Loss()(out, {k: torch.from_numpy(v).to(DEV) for k, v in gtn.items()}, dr.to(DEV))
        loss.backward()
    finally:
        layers.hip_train = orig
    de = [float(np.abs(o.detach().cpu().numpy() - d).mean()) for o, d in zip(out["depth"], d64)]
    return de, {k: l2(p.grad.cpu().numpy(), g64[k]) for k, p in m.named_parameters()}


classes = sorted({type(mod).__name__ for mod in m0.modules()})
print("module classes:", classes)
groups = ["Backbone", "Homoaggre.0", "Homoaggre.1", "Homoaggre.2", "Regular.0", "Regular.1", "Regular.2", "Refine"]
dref = [float(np.abs(o.detach().numpy() - d).mean()) for o, d in zip(out_ref["depth"], d64)]
print("fp32 oracle depth err vs f64:", ["%.2e" % e for e in dref])
variants = [("full HIP", set())] + [(f"stock {c}", {c}) for c in sys.argv[1:]]
for name, st in variants:
    de, ge = run(st)
    print(f"\n== {name}: depth err vs f64 {['%.2e' % e for e in de]} (ratio {['%.2f' % (a / b) for a, b in zip(de, dref)]})")
    for grp in groups:
        ks = [k for k in ge if k.startswith(grp)]
        rs = sorted(((ge[k] / max(ref_err[k], 1e-30), ge[k], ref_err[k], k) for k in ks), reverse=True)
        med = float(np.median([r[0] for r in rs]))
        print(f"  {grp:12s} n={len(ks):3d} median ratio {med:5.2f}  worst {rs[0][0]:5.2f} ({rs[0][1]:.1e} vs {rs[0][2]:.1e}) {rs[0][3]}")
    if name == "full HIP":
        for grp in ("Regular.2", "Homoaggre.0", "Homoaggre.2"):
            for k in [k for k in ge if k.startswith(grp)]:
                print(f"     {k:44s} HIP {ge[k]:.2e}  fp32 oracle {ref_err[k]:.2e}  ratio {ge[k] / max(ref_err[k], 1e-30):.2f}")
