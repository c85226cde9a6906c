"""Which python lines of the training step run stock ATen ops on the GPU (copies, fills, adds, ...)?  A TorchDispatchMode
records every ATen call touching a CUDA tensor with the innermost frame inside this repo.  dev tool"""
import collections, os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from mdfnet_hip import synth, ddp
from net import loss as loss_mod
dev = torch.device('cuda', 0)
W, H, V = 768, 576, 5
model = bench.build(dev).train()
bucket = ddp.FlatBucket(model)
if os.environ.get("MDF_TRAIN_STOCK") == "1":
    import rehearsal
    rehearsal.enable(on_gpu=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
else:
    from mdfnet_hip.optim import FlatAdam
    opt = FlatAdam(bucket, lr=1e-3)
crit = loss_mod.Loss().to(dev)
imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(W, H, V, batch=1, rot_deg=2.0, seed=3))
gt = {str(k): (torch.rand(1, H >> k, W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}


def step():
    out = model(imgs, extr, intr, dr)
    loss = crit(out, gt, dr)
    bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients(); opt.step()


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
            counts[(name.replace("aten.", ""), frame)] += 1
        return func(*args, **(kwargs or {}))


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 2
with Rec():
    for _ in range(N):
        step()
torch.cuda.synchronize()
tot = 0
for (name, frame), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    tot += n
    print(f"{n / N:7.1f} /step  {name:36s} {frame}")
print("total stock ATen calls on GPU tensors per step:", tot / N)
