"""cfg3-shaped training step on one GPU: 768x576, 5 views, batch 1, forward + loss + backward + flat-bucket all-reduce
(1 rank) + Adam, on the HIP training path (MDF_TRAIN_STOCK=1: the PyTorch-ROCm autograd path, for comparison).
Prints samples/s and the per-kernel-family time of one step (HIP events on the launch stream).  dev tool"""
import collections, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/mdf-net_amd']
import torch
import bench
from mdfnet_hip import synth, ddp, ops
from net import loss as loss_mod
dev = torch.device('cuda', 0)
W, H, V = (int(x) for x in os.environ.get("MDF_TRAIN_SHAPE", "768,576,5").split(","))
NB = int(os.environ.get("MDF_TRAIN_BATCH", "1"))          # dev: a larger batch makes the eager step GPU-bound (stream A/Bs)
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
imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(W, H, V, batch=NB, rot_deg=2.0, seed=3))
gt = {str(k): (torch.rand(NB, H >> k, W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}   # dtutrain.py:55-58 key order
def step():
    out = model(imgs, extr, intr, dr)
    loss = crit(out, gt, dr)
    bucket.zero_grad(); loss.backward(); bucket.allreduce_gradients(); opt.step()
    return loss.detach()
for _ in range(3): l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("MDF_TRAIN_STEPS", "10"))
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
if os.environ.get("MDF_TRAIN_GRAPH") == "1":       # the same step recorded once and replayed (mdfnet_hip/graphstep.py)
    from mdfnet_hip.graphstep import GraphedTrainStep
    t0 = time.perf_counter()
    gstep = GraphedTrainStep(model, crit, bucket, opt, (imgs, extr, intr, dr, gt), warmup=1)
    torch.cuda.synchronize(); trec = time.perf_counter() - t0
    for _ in range(3): lg = gstep(imgs, extr, intr, dr, gt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): lg = gstep(imgs, extr, intr, dr, gt)
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / n
    print(f"train step {W}x{H}x{V} B=1 [one hipGraph replay per step]: {dtg*1e3:.2f} ms ({1/dtg:.2f} samples/s; host {t_issue*1e3:.2f} ms per step), "
          f"loss {float(lg):.3f}, recording {trec:.2f} s, eager {dt*1e3:.2f} ms", flush=True)
    if gstep.graph_c is not None and os.environ.get("MDF_TRAIN_PIECES") == "1":
        # the five graphs of the split recording, each replayed ALONE (synchronised on both sides): what runs beside what
        def timed(g, st=None):
            ts = []
            for _ in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if st is None: g.replay()
                else:
                    with torch.cuda.stream(st): g.replay()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3
        gstep._upload(*[__import__("mdfnet_hip").hostmirror.get(t) for t in (extr, intr, dr)])
        parts = [("F", timed(gstep.graph_a))] + [(f"S{i}", timed(g, st)) for i, (g, st) in enumerate(zip(gstep.graph_s, gstep.side))] + ([("R", timed(gstep.graph_r))] if gstep.graph_r is not None else []) + [("C", timed(gstep.graph_c))] + ([("W", timed(gstep.graph_w, gstep.side[0]))] if gstep.graph_w is not None else []) + [("D", timed(gstep.graph_d))]
        def timed_group(pairs):          # [(graph, stream or None)] replayed side by side
            ts = []
            cur = torch.cuda.current_stream(dev)
            for _ in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for g, st in pairs:
                    if st is None: g.replay()
                    else:
                        st.wait_stream(cur)
                        with torch.cuda.stream(st): g.replay()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3
        sreg = timed_group([(g, st) for g, st in zip(gstep.graph_s, gstep.side)] + ([(gstep.graph_r, None)] if gstep.graph_r is not None else []))
        cw = timed_group(([(gstep.graph_w, gstep.side[0])] if gstep.graph_w is not None else []) + [(gstep.graph_c, None)])
        print(f"side by side (ms): S0 | S1 | S2 | R {sreg:.2f}; C | W {cw:.2f}", flush=True)
        print("pieces alone (ms): " + ", ".join(f"{k} {v:.2f}" for k, v in parts) + f"; sum {sum(v for _, v in parts):.2f}", flush=True)
mode = "stock PyTorch-ROCm autograd" if os.environ.get("MDF_TRAIN_STOCK") == "1" else "HIP training kernels"
print(f"train step {W}x{H}x{V} B=1 [{mode}]: {dt*1e3:.1f} ms  ({1/dt:.2f} samples/s), loss {float(l):.3f}, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
if os.environ.get("MDF_TRAIN_STOCK") != "1" and not os.environ.get("MDF_TRAIN_NOPROFILE"):
    ops.profile_begin()
    step()
    fam = collections.OrderedDict()
    for name, tag, ms, work in ops.profile_end():
        key = name.replace("mdf_", "") + ((" " + tag.split()[0] + " " + tag.split()[1]) if name == "mdf_warp_aggregate_vec_train" else "")
        f = fam.setdefault(key, [0.0, 0, 0.0, 0.0])
        f[0] += ms; f[1] += 1; f[2] += work.get("flops", 0.0); f[3] += work.get("bytes", 0.0)
    tot = sum(f[0] for f in fam.values())
    print(f"hand-written kernels in one step: {tot:.2f} ms over {sum(f[1] for f in fam.values())} launches")
    for k, (ms, cnt, fl, by) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        extra = f"{fl / ms / 1e9:7.1f} TFLOP/s" if fl else (f"{by / ms / 1e6:7.0f} GB/s" if by else "")
        print(f"  {k:42s} {ms:8.3f} ms {cnt:4d} launches {extra}")
    if os.environ.get("MDF_TRAIN_VERBOSE"):
        ops.profile_begin(); step()
        for name, tag, ms, work in ops.profile_end():
            print(f"    {name:34s} {tag:40s} {ms*1e3:9.1f} us")
