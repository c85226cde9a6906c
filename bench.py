#!/usr/bin/env python3
"""Headline benchmark: views/sec of the 4-scale MVS eval forward at DTU 1600x1184 (cropped 1600x1200), 5 views,
hypotheses (48,24,8), batch 1 per rank (BASELINE.json configs[1]); synthetic DTU-shaped tensors, seeded weights.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one reference-view inference (one CoreNet.forward, B=1) with the inputs already resident in HBM.
Views shard across ranks with no data-path collective (eval items are independent): weak scaling.
Rank 0 prints ONE JSON line; it also carries
  roofline      the dominant hand-written kernel family (by time), measured live with HIP events on the launch
                stream in a separate profile pass: algorithmic flops (or bytes) / summed launch time vs CDNA4 peak
  kernels       the same for every hand-written kernel family + the stock (MIOpen) remainder
  cpu_baseline  the oracle (CPU restatement of the reference, kind "port") timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "mdf-net_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)

WIDTH, HEIGHT, VIEWS = 1600, 1184, 5
if os.environ.get("MDF_BENCH_SIZE"):            # rehearsal knob of tests/test_bench_launch_gpu.py (never set by the driver)
    WIDTH, HEIGHT, VIEWS = (int(v) for v in os.environ["MDF_BENCH_SIZE"].split("x"))


def build(device):
    import contextlib
    import io
    from mdfnet_hip import synth
    with contextlib.redirect_stdout(io.StringIO()):
        import config
        model = config.build_model()
    model.load_state_dict(synth.seeded_state_dict(model.state_dict(), seed=1))
    return model.eval().to(device)


def family(name, tag):
    if name == "mdf_conv3d_fwd":
        return "conv3d (regulariser): conv_lds_kernel / conv3d_kernel, fp32 MFMA implicit GEMM"
    if name == "mdf_conv2d_fwd":
        return "conv2d (feature pyramid + refine + prob-head partial sums): conv_lds_kernel, fp32 MFMA implicit GEMM"
    if name == "mdf_warp_aggregate_vec_fwd":
        return "warp_kernel<kVec> (fused warp+aggregate)"
    if name == "mdf_prob_softmax_regress_fwd":
        return "prob_head_kernel"
    if name == "mdf_prob_from_partials_fwd":
        return "prob_from_partials_kernel (combine + softmax(D) + soft-argmin)"
    return name.replace("mdf_", "").replace("_fwd", "") + "_kernel"


def profile_pass(model, inputs, steps=3):
    """Per-launch HIP-event timing of every hand-written kernel (events recorded on the launch stream)."""
    from mdfnet_hip import ops
    agg = {}
    with torch.no_grad():
        for _ in range(steps):
            ops.profile_begin()
            model(*inputs)
            for name, tag, ms, work in ops.profile_end():
                if name == "mdf_conv3d_pack_weights":
                    continue
                f = agg.setdefault(family(name, tag), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0,
                                                       "bound": work.get("bound", "hbm"), "top": {}})
                f["ms"] += ms
                f["flops"] += work.get("flops", 0.0)
                f["bytes"] += work.get("bytes", 0.0)
                f["launches"] += 1
                t = f["top"].setdefault(tag, [0.0, 0.0, 0.0, 0])
                t[0] += ms; t[1] += work.get("flops", 0.0); t[2] += work.get("bytes", 0.0); t[3] += 1
    out = []
    for fam, f in agg.items():
        ms = f["ms"] / steps
        rec = {"kernel": fam, "launches_per_step": f["launches"] // steps, "ms_per_step": round(ms, 4), "bound": f["bound"]}
        if f["bound"] == "mfma" and f["flops"]:
            ach = f["flops"] / steps / (ms * 1e-3) / 1e12
            rec.update(achieved=round(ach, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                       algorithmic_gflop_per_step=round(f["flops"] / steps / 1e9, 2))
        elif f["bytes"]:
            ach = f["bytes"] / steps / (ms * 1e-3) / 1e9
            rec.update(achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4),
                       algorithmic_mb_per_step=round(f["bytes"] / steps / 1e6, 1))
        if f["bound"] == "mfma":
            # layers whose arithmetic intensity is below the fp32 ridge (157.3 TF/s / 8 TB/s = 19.7 flop/B) are
            # HBM-bound by their algorithmic bytes and are reported against the HBM peak instead
            ridge = PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            hi = [v for v in f["top"].values() if v[2] and v[1] / v[2] >= ridge]
            lo = {k: v for k, v in f["top"].items() if v[2] and v[1] / v[2] < ridge}
            if hi:
                hms = sum(v[0] for v in hi) / steps
                hfl = sum(v[1] for v in hi) / steps
                rec["layers_above_ridge"] = {"ms_per_step": round(hms, 4), "achieved": round(hfl / (hms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                             "frac": round(hfl / (hms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            if lo:
                lms = sum(v[0] for v in lo.values()) / steps
                lby = sum(v[2] for v in lo.values()) / steps
                rec["layers_below_ridge"] = {"ms_per_step": round(lms, 4), "achieved": round(lby / (lms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                             "frac": round(lby / (lms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "bound": "hbm",
                                             "shapes": sorted(lo.keys())}
        heavy = max(f["top"].items(), key=lambda kv: kv[1][0])
        rec["heaviest_launch"] = {"shape": heavy[0], "ms": round(heavy[1][0] / heavy[1][3], 4)}
        if f["bound"] == "mfma" and heavy[1][1]:
            rec["heaviest_launch"]["tflops"] = round(heavy[1][1] / heavy[1][3] / (heavy[1][0] / heavy[1][3] * 1e-3) / 1e12, 2)
        elif heavy[1][2]:
            rec["heaviest_launch"]["gbs"] = round(heavy[1][2] / heavy[1][3] / (heavy[1][0] / heavy[1][3] * 1e-3) / 1e9, 1)
        out.append(rec)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def measured_traffic():
    """HBM bytes per forward per kernel family, measured offline with rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in separate
    passes, gfx950 corrections applied; scripts/summarize_traffic.py) and committed under profiles/."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get("families", {})


def cpu_baseline(timed_views=3):
    """Oracle (CPU port of the reference algorithm) on the host cores; bounded sample of the same workload: BASELINE.md
    section 3's protocol -- 1 full-size warm-up view + 3 timed full-size views, median."""
    import platform
    import statistics
    from mdfnet_hip import synth
    from oracle import mvs_oracle as O
    model_sd = synth.seeded_state_dict(build("cpu").state_dict(), seed=1)
    # the GPU box hands one GPU's job a 16-core share of the host (256 logical cores are visible; using them
    # all oversubscribes and is 15x slower): use the affinity mask, capped at that share
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MDF_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    cpu_model = platform.processor() or "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass
    times = []
    with torch.no_grad():
        scene = synth.make_scene(WIDTH, HEIGHT, VIEWS, seed=0)
        O.core_forward(model_sd, *scene)          # full-size warm-up (oneDNN primitives, thread pool, page faults)
        for _ in range(timed_views):
            t0 = time.time()
            O.core_forward(model_sd, *scene)
            times.append(time.time() - t0)
    med = statistics.median(times)
    return {"value": round(1.0 / med, 4), "unit": "views/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model,
            "sample": f"1 warm-up + {timed_views} timed full-size views ({WIDTH}x{HEIGHT}x{VIEWS}, hypotheses 48/24/8) through "
                      f"oracle.core_forward (torch {torch.__version__} CPU, {cores} threads on {cpu_model}); median {med:.2f} s/view, "
                      f"all {[round(t, 2) for t in times]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # ~0.9 s of timed region at 4.4 ms/step
    ap.add_argument("--warmup", type=int, default=10)
    from mdfnet_hip.pipeline import DEFAULT_IN_FLIGHT
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MDF_BENCH_IN_FLIGHT", DEFAULT_IN_FLIGHT)),
                    help="items in flight on that many HIP streams (the eval driver's pipelining); 1 = strictly one at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before this process touches the GPU (no HIP call
        # has happened yet; the children are fresh subprocesses, nothing is exec'ed over an initialised process)
        from mdfnet_hip import shard
        raise SystemExit(shard.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); the product path has no CPU fallback")
    # rehearsal knobs for a 1-GPU box (never set by the driver): several ranks on one card, gloo instead of RCCL
    backend = os.environ.get("MDF_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("MDF_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from mdfnet_hip import synth

    model = build(dev)
    inputs = tuple(t.to(dev) for t in synth.make_scene(WIDTH, HEIGHT, VIEWS, batch=1, rot_deg=3.0, seed=100 + rank))

    def barrier():
        if world > 1:
            dist.barrier()

    from mdfnet_hip.pipeline import InFlight
    last = {}
    pipe = InFlight(dev, args.in_flight, done=lambda tag, o: last.__setitem__("out", o))

    from mdfnet_hip import hostmirror
    cams_cpu = tuple(t.cpu() for t in inputs[1:])

    def fresh_cameras():
        """The (tiny) camera / depth-range tensors are fresh objects every step, handed over the way eval.py:run_eval hands the
        loader's batch over: host tensors copied to the device, the host copies registered as their mirrors.  The images stay
        resident in HBM; the control-plane work (host prelude, small H2D copies) is part of every timed step."""
        host = tuple(t.clone() for t in cams_cpu)
        devs = tuple(t.to(dev, non_blocking=True) for t in host)
        for d_, h_ in zip(devs, host):
            hostmirror.put(d_, h_)
        return devs

    def one_step():
        cams = fresh_cameras()
        pipe.submit(lambda c=cams: model(inputs[0], *c), keep=cams)

    with torch.no_grad():
        for _ in range(max(args.warmup, 2 * args.in_flight if args.in_flight > 1 else 0)):   # also warms each stream's allocator pool
            one_step()
        pipe.drain()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        pipe.drain()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = last["out"]
        # the same K steps strictly one at a time (latency view of the same work), rank 0 only, outside the timed region
        dt_serial = None
        if rank == 0 and args.in_flight > 1:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.steps):
                model(inputs[0], *fresh_cameras())
            torch.cuda.synchronize()
            dt_serial = time.perf_counter() - ts
    assert torch.isfinite(out["depth"]).all()
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kernels, cpu = None, None
    if rank == 0 and not args.no_profile:
        kernels = profile_pass(model, inputs)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        views_per_s = world * args.steps / dt
        rec = {"metric": "views/sec at DTU 1600x1200x5-view x4-scale", "value": round(views_per_s, 3), "unit": "views/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "items_in_flight": args.in_flight,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"DTU eval {WIDTH}x{HEIGHT} (1600x1200 cropped as load/dtueval.py:34), {VIEWS} views, "
                                      "3 cost-volume stages + x2 refine = 4 output scales, hypotheses (48,24,8), batch 1 per rank, "
                                      "seeded random weights (pth/dtu_29.pth is not available offline)",
                          "views_per_rank_per_step": 1, "parallelism": f"views sharded over {world} rank(s), no collective",
                          "pipelining": (f"{args.in_flight} independent views in flight per rank on {args.in_flight} HIP streams, as the "
                                         "eval driver issues them (mdfnet_hip/pipeline.py); every step's work completes inside the "
                                         "timed region" if args.in_flight > 1 else "one view at a time")}}
        if dt_serial is not None:
            rec["one_at_a_time"] = {"value": round(args.steps / dt_serial, 3), "unit": "views/s", "ms_per_view": round(1e3 * dt_serial / args.steps, 3),
                                    "note": "same steps with a single view in flight (rank 0, outside the timed region)"}
        if kernels:
            # dominant kernel = the MFMA implicit-GEMM conv kernel (one template family, conv_lds.hip/conv3d.hip), summed
            # over its 2-D and 3-D launches: algorithmic flops / summed launch time
            traffic = measured_traffic()
            fam_of = {"conv": "mfma_conv", "warp": "warp_aggregate", "prob": "prob_head"}
            for k in kernels:
                key = next((v for pre, v in fam_of.items() if k["kernel"].startswith(pre)), None)
                if key in traffic:
                    k["traffic"] = {"hbm_bytes_per_step": round(traffic[key]["hbm_bytes_per_forward"]),
                                    "read": round(traffic[key]["read_bytes_per_forward"]),
                                    "write": round(traffic[key]["write_bytes_per_forward"]),
                                    "source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 on gfx950)"}
            mf = [k for k in kernels if k["bound"] == "mfma" and "achieved" in k]
            if mf:
                ms = sum(k["ms_per_step"] for k in mf)
                gf = sum(k["algorithmic_gflop_per_step"] for k in mf)
                ach = gf / ms
                rec["roofline"] = {"kernel": "fp32-MFMA implicit-GEMM conv family (conv_lds_kernel + conv3d_kernel)", "bound": "mfma",
                                   "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "traffic": (round(traffic["mfma_conv"]["hbm_bytes_per_forward"]) if "mfma_conv" in traffic else None),
                                   "traffic_unit": "HBM bytes per step over all launches of the family (PMC, offline)",
                                   "mode": "one view at a time on one stream (profile pass after the timed region: per-launch HIP events "
                                           "on the launch stream); compare with one_at_a_time, not with the in-flight headline",
                                   "ms_per_step": round(ms, 3),
                                   "launches_per_step": sum(k["launches_per_step"] for k in mf),
                                   "algorithmic_gflop_per_step": round(gf, 1)}
            rec["kernels"] = kernels
            rec["hip_kernels_ms_per_step"] = round(sum(k["ms_per_step"] for k in kernels), 3)
        if cpu:
            rec["cpu_baseline"] = cpu
        print(json.dumps(rec), flush=True)
    if world > 1:
        barrier()                      # rank 0's profile pass is over: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
