"""Which python lines of ONE eval forward (cfg2) run stock ATen ops on GPU tensors / host<->device copies?  dev tool"""
import collections, os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from mdfnet_hip import synth
dev = torch.device('cuda', 0)
model = bench.build(dev).eval()
W, H, V = 1600, 1184, 5
imgs, extr, intr, dr = synth.make_scene(W, H, V, batch=1, rot_deg=2.0, seed=3)
imgs = imgs.to(dev)
VIEW_OPS = {"view", "permute", "transpose", "reshape", "slice", "select", "detach", "alias", "expand", "unsqueeze", "squeeze", "as_strided",
            "t", "unbind", "split", "_unsafe_view", "empty", "empty_like", "empty_strided", "new_empty", "is_same_size", "sym_size", "sym_stride",
            "sym_numel", "unfold", "_reshape_alias", "view_as", "split_with_sizes", "narrow", "lift_fresh", "_local_scalar_dense"}
counts = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
        for a in args:
            if isinstance(a, (list, tuple)):
                flat += [t for t in a if isinstance(t, torch.Tensor)]
        base = name.replace("aten.", "").split(".")[0]
        dev_kw = (kwargs or {}).get("device")
        on_gpu = any(t.is_cuda for t in flat) or (dev_kw is not None and "cuda" in str(dev_kw))
        if on_gpu and base not in VIEW_OPS:
            frame = "?"
            for fs in reversed(traceback.extract_stack()):
                fn = fs.filename
                if fn.startswith(R) and "scripts/" not in fn:
                    frame = f"{fn[len(R) + 1:]}:{fs.lineno} {fs.name}"
                    break
            shapes = ",".join(str(tuple(t.shape)) for t in flat[:2])
            counts[(name.replace("aten.", ""), frame, shapes)] += 1
        return func(*args, **(kwargs or {}))


def fwd():
    with torch.no_grad():
        return model(imgs, extr.clone().to(dev, non_blocking=True), intr.clone().to(dev, non_blocking=True), dr.clone().to(dev, non_blocking=True))


for _ in range(3):
    fwd()
torch.cuda.synchronize()
with Rec():
    fwd()
torch.cuda.synchronize()
tot = 0
for (name, frame, shapes), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    tot += n
    print(f"{n:4d}  {name:30s} {frame:60s} {shapes}")
print("total:", tot)
