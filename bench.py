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
  training      BASELINE configs[2].  N=1: one GPU's training step (768x576x5, batch 1): `ms_per_step` is the step replayed from one
                hipGraph recording (what train.py runs with MDF_TRAIN_HIPGRAPH=1), `eager` the same step issued launch by launch
                (host-bound on slow host shares), with the per-family kernel table; N>1: EVERY rank
                runs that step data-parallel with the flat-bucket gradient exchange over RCCL -- aggregate samples/s, the
                collective's ms per step, and the direct (all-to-all + all-gather) exchange as a second figure
  cfg4          BASELINE configs[3]: 1920x1056 with 7 and 11 views, views/s (rank 0, N=1)
  cfg5_scan     BASELINE configs[4], one scan: 49 items with the cross-item feature cache + the fused consistency filter (rank 0, N=1)
The headline `value` / `config` are the eval figures for every N; the blocks above are extra keys of the same line.
"""
import argparse
import math
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
    if name in ("mdf_conv2d_fwd", "mdf_conv2d_pair_fwd", "mdf_conv2d_res_pair_fwd", "mdf_conv1x1_heads_fwd", "mdf_refine_tail_fwd", "mdf_prob_fused_fwd"):
        # (the one-launch prob head stays in this family: it is the prob-head partial-sum conv with the softmax behind it)
        return ("conv2d (feature pyramid + refine + prob-head partial sums): conv_lds_kernel / conv_pair_valu_kernel / conv1x1_kernel / "
                "res_pair_kernel / refine_tail_kernel / prob_fused_kernel, fp32 MFMA implicit GEMM")
    if name == "mdf_warp_aggregate_vec_fwd":
        return "warp_kernel<kVec> (fused warp+aggregate)"
    if name == "mdf_prob_softmax_regress_fwd":
        return "prob_head_kernel"
    if name == "mdf_prob_from_partials_fwd":
        return "prob_from_partials_kernel (combine + softmax(D) + soft-argmin)"
    return name.replace("mdf_", "").replace("_fwd", "") + "_kernel"


def _wall_ms(fn, reps=3):
    """GPU time of fn() from ONE HIP event pair on the launch stream (no per-launch instrumentation), mean over reps."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


_SLEEP_CYCLES_PER_MS = None


def _gpu_sleep(ms):
    """Keep the stream busy for ~ms (torch's spin kernel), so that the host runs AHEAD of the GPU in what follows."""
    global _SLEEP_CYCLES_PER_MS
    if ms <= 0:
        return
    if _SLEEP_CYCLES_PER_MS is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(2_000_000)
        e1.record()
        torch.cuda.synchronize()
        _SLEEP_CYCLES_PER_MS = 2_000_000 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms * _SLEEP_CYCLES_PER_MS))


def _empty_bracket_ms(n=64):
    """Cost of the instrumentation itself: an event pair with NOTHING between, on a busy stream (median of n)."""
    _gpu_sleep(2.0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


def bracketed_launches(fn, steps=3):
    """Per-launch HIP-event timing of every hand-written kernel fn() issues (events recorded on the launch stream).
    Two things would make raw brackets over-count (the family sum exceeded the one-at-a-time wall):
      * when the host cannot keep up with two event records per launch, the stream runs dry and a bracket includes the host's
        time between `e0.record()` and the kernel's submission -> every instrumented pass starts behind a GPU sleep long enough
        for the host to stay ahead (measured first: instrumented wall - plain wall);
      * the marker packets themselves -> the cost of an EMPTY bracket on a busy stream is subtracted from every bracket.
    Safety net: if the corrected brackets of a step still add up to more than the plain (un-instrumented) GPU wall of the same
    step, they are scaled down to it (`scaled_by` < 1 in the info).  -> (records [(abi, tag, ms, work)] of all steps, info)"""
    from mdfnet_hip import ops
    w_plain = _wall_ms(fn, steps)

    def instrumented(lead_ms):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        _gpu_sleep(lead_ms)
        e0.record()
        ops.profile_begin()
        fn()
        e1.record()
        r = ops.profile_end()
        return r, e0.elapsed_time(e1)
    _, w_trial = instrumented(0.0)
    lead = max(0.0, 1.5 * (w_trial - w_plain)) + 0.5
    empty = _empty_bracket_ms()
    recs, w_prof = [], 0.0
    for _ in range(steps):
        r, w = instrumented(lead)
        recs += r
        w_prof += w
    w_prof /= steps
    recs = [(n, tag, max(ms - empty, 0.0), work) for n, tag, ms, work in recs]
    total = sum(r[2] for r in recs) / steps
    scale = min(1.0, w_plain / total) if total > 0 else 1.0
    if scale < 1.0:
        recs = [(n, tag, ms * scale, work) for n, tag, ms, work in recs]
    return recs, {"wall_plain_ms": round(w_plain, 4), "wall_instrumented_ms": round(w_prof, 4),
                  "wall_instrumented_without_lead_ms": round(w_trial, 4), "gpu_lead_ms": round(lead, 3),
                  "brackets_per_step": round(len(recs) / steps, 1), "empty_bracket_us": round(empty * 1e3, 2),
                  "bracketed_sum_ms": round(total * scale, 4), "scaled_by": round(scale, 4),
                  "note": "per-launch time = HIP-event bracket on the launch stream minus the empty-bracket cost; instrumented passes run "
                          "behind a GPU sleep so the host stays ahead; the bracketed sum cannot exceed wall_plain_ms"}


def family_table(recs, steps, family_of):
    """records of `steps` steps -> one entry per kernel family: launches, ms, algorithmic work, achieved vs CDNA4 peak."""
    agg = {}
    for name, tag, ms, work in recs:
        fam = family_of(name, tag)
        if fam is None:
            continue
        f = agg.setdefault(fam, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0,
                                 "bound": work.get("bound", "hbm"), "top": {}})
        if work.get("bound") == "mfma":           # (a family that mixes MFMA launches with their small sum / control launches is MFMA-bound)
            f["bound"] = "mfma"
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
        if ms <= 0:
            out.append(rec)
            continue
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
            hi = [v for v in f["top"].values() if v[2] and v[1] / v[2] >= ridge and v[0] > 0]
            lo = {k: v for k, v in f["top"].items() if v[2] and v[1] / v[2] < ridge and v[0] > 0}
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
        if heavy[1][0] > 0:
            rec["heaviest_launch"] = {"shape": heavy[0], "ms": round(heavy[1][0] / heavy[1][3], 4)}
            if f["bound"] == "mfma" and heavy[1][1]:
                rec["heaviest_launch"]["tflops"] = round(heavy[1][1] / heavy[1][3] / (heavy[1][0] / heavy[1][3] * 1e-3) / 1e12, 2)
            elif heavy[1][2]:
                rec["heaviest_launch"]["gbs"] = round(heavy[1][2] / heavy[1][3] / (heavy[1][0] / heavy[1][3] * 1e-3) / 1e9, 1)
        out.append(rec)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def profile_pass(model, inputs, steps=3):
    """Eval forward, one view at a time: per-family table of the hand-written kernels + the instrumentation calibration."""
    def fwd():
        with torch.no_grad():
            model(*inputs)
    recs, info = bracketed_launches(fwd, steps)
    info["per_kernel"] = per_kernel_table(recs, steps)
    return family_table(recs, steps, lambda n, t: None if n == "mdf_conv3d_pack_weights" else family(n, t)), info


def per_kernel_table(recs, steps):
    """Per __global__ function (the kernel each ABI entry enqueued last, mdf_last_launch): launches, bracketed time and the
    ALGORITHMIC bytes / flops of the calls it served, per step -- what scripts/summarize_traffic.py sets beside the PMC bytes of the
    same kernel (wasted traffic attributable per kernel)."""
    tab = {}
    for n, tag, ms, work in recs:
        k = work.get("kernel") or n
        t = tab.setdefault(k, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
        t["launches"] += 1
        t["ms"] += ms
        t["bytes"] += float(work.get("bytes", 0.0))
        t["flops"] += float(work.get("flops", 0.0))
    return {k: {"launches_per_step": round(v["launches"] / steps, 1), "ms_per_step": round(v["ms"] / steps, 4),
                "algorithmic_mb_per_step": round(v["bytes"] / steps / 1e6, 2), "algorithmic_gflop_per_step": round(v["flops"] / steps / 1e9, 2)}
            for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"])}


def _newest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present -> (path, name) or (None, None)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return (c[-1], "profiles/" + os.path.basename(c[-1])) if c else (None, None)


def measured_traffic(kind="eval"):
    """HBM bytes per step per kernel family, measured offline with rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE in separate
    passes, gfx950 corrections applied; scripts/summarize_traffic.py) and committed under profiles/: the newest round's file
    for this workload.  -> (families dict, file name)"""
    path, name = _newest_profile({"eval": "traffic.json", "train": "train_traffic.json"}[kind])
    if path:
        with open(path) as f:
            return json.load(f).get("families", {}), name
    return {}, None


def rocprof_conv_family():
    """The conv family's time per forward by rocprofv3 --kernel-trace --stats of the SAME command (one view at a time), from the
    newest summary committed under profiles/ (scripts/r04_profiles.sh): the cross-check the contract asks the live figure to
    agree with.  The family is every kernel mdfnet_hip/kernel_families.py files under `mfma_conv` (exact function names, with a
    CPU test that no `__global__` of csrc/ is unclassified -- the r03 figure missed `convtr_all_kernel`).  -> dict or None"""
    import csv
    from mdfnet_hip import kernel_families as KF
    path, name = _newest_profile("bench_cfg2_kernel_stats.csv")
    if not path:
        return None
    ns, forwards, launches = 0, 0, 0
    with open(path) as f:
        for r in csv.DictReader(f):
            if KF.function_name(r["Name"]) in ("conv_pair_kernel", "conv_pair_valu_kernel"):
                forwards = int(r["Calls"])                      # exactly one launch per forward
            if KF.family(r["Name"]) == KF.MFMA_CONV:
                ns += int(r["TotalDurationNs"])
                launches += int(r["Calls"])
    if not forwards:
        return None
    return {"ms_per_step": round(ns / forwards / 1e6, 3), "launches_per_step": round(launches / forwards, 1), "forwards": forwards,
            "source": name + " (rocprofv3 --kernel-trace --stats of bench.py --in-flight 1, offline)"}


def attach_traffic(kernels, traffic, source, fam_of):
    for k in kernels:
        key = next((v for pre, v in fam_of if k["kernel"].startswith(pre)), None)
        if key in traffic:
            k["traffic"] = {"hbm_bytes_per_step": round(traffic[key]["hbm_bytes_per_forward"]),
                            "read": round(traffic[key]["read_bytes_per_forward"]),
                            "write": round(traffic[key]["write_bytes_per_forward"]),
                            "source": f"{source} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 on gfx950)"}


# --------------------------------------------------------------------------- BASELINE config 3: the training step
TRAIN_W, TRAIN_H, TRAIN_V = 768, 576, 5


def train_family(name, tag):
    if name in ("mdf_conv3d_fwd", "mdf_conv3d_train_fwd"):
        return "conv3d forward + input gradients (regulariser): conv_lds_kernel / conv3d_kernel, fp32 MFMA"
    if name in ("mdf_conv2d_fwd", "mdf_conv2d_train_fwd"):
        return "conv2d forward + input gradients (feature pyramid, refine, prob partial sums): conv_lds_kernel, fp32 MFMA"
    if name in ("mdf_conv3d_wgrad", "mdf_conv3d_wgrad_partial", "mdf_wgrad_sum_batch", "mdf_wgrad_batch_flush"):
        # (mdf_wgrad_batch_flush: the step's deferred weight gradients, 3-D and 2-D, as one job-table launch per kernel form)
        return "wgrad3d: wgrad_lds_kernel / wgrad_lds_batch_kernel / wgrad_kernel (weight gradients, split-K fp32 MFMA; the batched launch carries the 2-D layers' too)"
    if name in ("mdf_conv2d_wgrad", "mdf_conv2d_wgrad_partial"):
        return "wgrad2d: wgrad_lds_kernel / wgrad2d_kernel (weight gradients, split-K fp32 MFMA)"
    if name.startswith("mdf_bn_"):
        return "batchnorm (apply, backward; the sums ride in the conv epilogues): bn_*_kernel"
    if name == "mdf_warp_aggregate_vec_train":
        return "aggregate scatter (pass 3): warp_bwd_kernel" if "pass3" in tag else "aggregate passes 0-2: warp_train_kernel"
    if name.startswith("mdf_prob_") or name == "mdf_upsample2_bilinear_bwd":
        return "prob head + upsample backward"
    if name.startswith("mdf_masked_smooth_l1"):
        return "loss (masked smooth-L1, 4 scales)"
    if name in ("mdf_pack_batch", "mdf_adam_step"):
        return "weight packing + Adam (one launch each)"
    return "small per-pixel heads and control kernels"


def training_block(dev, steps, blocks, stock_steps):
    """BASELINE.json configs[2] on ONE GPU (rank 0, N=1; reported NEXT to the headline, never as `value`): one training step =
    forward + loss + backward + flat-bucket gather (the all-reduce is the identity at world 1) + Adam at 768x576, 5 views,
    batch 1 (the per-GPU share of batch 8 on 8 GPUs), on the hand-written training kernels (reference: train.py:36-45)."""
    import statistics
    from mdfnet_hip import synth, ddp, layers
    from mdfnet_hip.optim import FlatAdam
    from net import loss as loss_mod
    model = build(dev).train()
    bucket = ddp.FlatBucket(model)
    opt = FlatAdam(bucket, lr=1e-3)
    crit = loss_mod.Loss().to(dev)
    imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(TRAIN_W, TRAIN_H, TRAIN_V, batch=1, rot_deg=2.0, seed=3))
    gt = {str(k): (torch.rand(1, TRAIN_H >> k, TRAIN_W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}   # dtutrain.py:55-58

    def step(optimizer=opt):
        out = model(imgs, extr, intr, dr)
        loss = crit(out, gt, dr)
        bucket.zero_grad()
        loss.backward()
        bucket.allreduce_gradients()
        optimizer.step()
        return loss.detach()

    for _ in range(3):
        first = step()
    torch.cuda.synchronize()
    times = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / steps)
    med = statistics.median(times)
    assert torch.isfinite(last).item()
    # per-launch brackets on ONE stream: with the stages' backward chains on streams of their own (net/core.py) three launches run
    # side by side and every bracket would read the time of all of them
    from net import core as _core
    streams_on, _core._STAGE_STREAMS = _core._STAGE_STREAMS, False
    try:
        recs, info = bracketed_launches(step, 3)
    finally:
        _core._STAGE_STREAMS = streams_on
    kernels = family_table(recs, 3, train_family)
    traffic, source = measured_traffic("train")
    gflop = sum(k.get("algorithmic_gflop_per_step", 0.0) for k in kernels)
    rec = {"workload": f"BlendedMVS-shaped training step {TRAIN_W}x{TRAIN_H}, {TRAIN_V} views, batch 1 on one GPU (BASELINE configs[2] "
                       "per-GPU share): forward + masked smooth-L1 loss + backward + gradient bucket + Adam, hypotheses (48,24,8), "
                       "batch-statistics BatchNorm, synthetic tensors, seeded weights",
           "ms_per_step": round(1e3 * med, 3), "samples_per_s": round(1.0 / med, 2), "steps": steps, "blocks": blocks,
           "ms_per_step_blocks": [round(1e3 * t, 3) for t in times], "dtype": "f32",
           "loss_first": round(float(first), 3), "loss_last": round(float(last), 3),
           "peak_memory_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "mfma_algorithmic_gflop_per_step": round(gflop, 1),
           "whole_step_frac_of_fp32_mfma_peak": round(gflop / (med * 1e3) / PEAK_FP32_MFMA_TFLOPS, 4),
           "hip_kernels_ms_per_step": round(sum(k["ms_per_step"] for k in kernels), 3),
           "hip_launches_per_step": sum(k["launches_per_step"] for k in kernels),
           "instrumentation": info, "kernels": kernels}
    if traffic:
        rec["traffic"] = {"source": f"{source} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 on gfx950; conv = 2-D + 3-D "
                                    "forward and input-gradient launches, wgrad = 2-D + 3-D)",
                          "hbm_bytes_per_step": {k: {"read": round(v["read_bytes_per_forward"]), "write": round(v["write_bytes_per_forward"]),
                                                     "total": round(v["hbm_bytes_per_forward"]), "launches": v["launches_per_forward"]}
                                                 for k, v in traffic.items()}}
    if stock_steps > 0:
        # the same step through PyTorch-ROCm's own autograd (MIOpen convs, ATen grid_sample / BatchNorm): a stated baseline
        import rehearsal
        rehearsal.enable(on_gpu=True)                    # explicit: torch autograd over the stock ops instead of the HIP training kernels
        try:
            sopt = torch.optim.Adam(model.parameters(), lr=1e-3)
            step(sopt)                                   # warm-up (MIOpen find)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(stock_steps):
                step(sopt)
            torch.cuda.synchronize()
            st = (time.perf_counter() - t0) / stock_steps
            rec["stock_pytorch_rocm_baseline"] = {"ms_per_step": round(1e3 * st, 1), "samples_per_s": round(1.0 / st, 3), "steps": stock_steps,
                                                  "note": "same model, inputs and step through torch autograd on this GPU (rehearsal backend, mdf-net_amd/rehearsal)",
                                                  "speedup": round(st / med, 1)}
        finally:
            rehearsal.disable()
    return rec


def primary_training_figure(rec):
    """`training.ms_per_step` / `samples_per_s` = the step as train.py runs it with MDF_TRAIN_HIPGRAPH=1 (one hipGraph replay per step) when
    that variant ran; the launch-by-launch (eager) figures move under `training.eager`.  Both execute the same kernels; the eager step is
    issued at the host's limit (312 launches + the autograd plumbing in ~8.5 ms) and reads 8.5 / 9.1 / 10.3 ms on three boxes of one
    pool depending on the host share, the replayed step 8.44-8.47 ms on all of them -- the replay is the figure that describes the GPU work."""
    g = rec.get("graph_replay") or {}
    rec["eager"] = {k: rec[k] for k in ("ms_per_step", "samples_per_s", "ms_per_step_blocks") if k in rec}
    rec["eager"]["note"] = "the same step issued launch by launch from Python (host-bound on boxes with a slow host share)"
    if "ms_per_step" in g:
        rec["ms_per_step"], rec["samples_per_s"], rec["ms_per_step_blocks"] = g["ms_per_step"], g["samples_per_s"], g["ms_per_step_blocks"]
        rec["mode"] = "replayed from its hipGraph recording (mdfnet_hip/graphstep.py: eight chain-shaped graphs, the stages' backward chains side by side; train.py: MDF_TRAIN_HIPGRAPH=1); launch-by-launch figures under `eager`"
        rec["whole_step_frac_of_fp32_mfma_peak"] = round(rec["mfma_algorithmic_gflop_per_step"] / g["ms_per_step"] / PEAK_FP32_MFMA_TFLOPS, 4)
    else:
        rec["mode"] = "launch by launch (the recorded-step variant did not run: see graph_replay)"
    if "stock_pytorch_rocm_baseline" in rec:      # like with like: the baseline is issued launch by launch, so is `eager`
        sb = rec["stock_pytorch_rocm_baseline"]
        sb["speedup"] = round(sb["ms_per_step"] / rec["eager"]["ms_per_step"], 1)
        sb["speedup_note"] = "stock autograd (launch by launch) / HIP training kernels issued launch by launch (training.eager)"
        if "ms_per_step" in g:
            sb["speedup_vs_recorded_step"] = round(sb["ms_per_step"] / g["ms_per_step"], 1)


def training_graph_child(steps, blocks):
    """The same training step recorded ONCE as a hipGraph and replayed per step (mdfnet_hip/graphstep.py): run as a child process of
    the bench (`--train-graph-child`), prints one JSON line.  A replay costs the host one call; the GPU executes the same launches."""
    import statistics
    from mdfnet_hip import synth, ddp
    from mdfnet_hip.graphstep import GraphedTrainStep
    from mdfnet_hip.optim import FlatAdam
    from net import loss as loss_mod
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = build(dev).train()
    bucket = ddp.FlatBucket(model)
    opt = FlatAdam(bucket, lr=1e-3)
    crit = loss_mod.Loss().to(dev)
    imgs, extr, intr, dr = synth.make_scene(TRAIN_W, TRAIN_H, TRAIN_V, batch=1, rot_deg=2.0, seed=3)
    gt = {str(k): (torch.rand(1, TRAIN_H >> k, TRAIN_W >> k, device=dev) * 400 + 480) for k in (3, 2, 1, 0)}
    t0 = time.perf_counter()
    step = GraphedTrainStep(model, crit, bucket, opt, (imgs.to(dev), extr.to(dev), intr.to(dev), dr.to(dev), gt), warmup=2)
    torch.cuda.synchronize()
    t_rec = time.perf_counter() - t0
    imgs_d = imgs.to(dev)
    for _ in range(3):
        first = step(imgs_d, extr, intr, dr, gt)          # cameras and range as HOST tensors (a loader's): their arithmetic is the host's
    first = float(first)
    torch.cuda.synchronize()
    times, hosts = [], []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step(imgs_d, extr, intr, dr, gt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / steps)
        hosts.append((t1 - t0) / steps)
    med = statistics.median(times)
    assert torch.isfinite(last).item()
    print(json.dumps({"ms_per_step": round(1e3 * med, 3), "samples_per_s": round(1.0 / med, 2), "steps": steps, "blocks": blocks,
                      "ms_per_step_blocks": [round(1e3 * t, 3) for t in times],
                      "host_ms_per_step": round(1e3 * statistics.median(hosts), 3), "recording_s": round(t_rec, 2),
                      "loss_first": round(first, 3), "loss_last": round(float(last), 3),
                      "note": "forward + loss + backward + bucket + Adam recorded once (torch.cuda.graph over the hand-written launches) and "
                              "replayed per step -- as eight chain-shaped graphs, the three stages' backward chains side by side on three streams, their weight gradients beside the trunk's chain "
                              "(mdfnet_hip/graphstep.py); per step the host uploads the packed control plane + Adam's scalars and issues the "
                              "replays (host_ms_per_step), the GPU executes the same launches as the eager step"}), flush=True)


def training_graph(steps, blocks, timeout=240.0):
    """Runs training_graph_child in a fresh process (a fault there must not take the bench line with it) -> dict."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train-graph-child", "--train-steps", str(steps), "--blocks", str(blocks)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"child exited with {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except Exception as e:      # noqa: BLE001  (timeout, spawn failure: reported, never fatal for the bench line)
        return {"error": f"{type(e).__name__}: {e}"}


def _throughput(dev, fn_factory, n_items, in_flight, warm):
    """views/s of `n_items` calls issued through the eval driver's InFlight queue (all work completes inside the timed region)."""
    from mdfnet_hip.pipeline import InFlight
    pipe = InFlight(dev, in_flight)
    for i in range(warm):
        pipe.submit(fn_factory(i))
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_items):
        pipe.submit(fn_factory(i))
    pipe.drain()
    torch.cuda.synchronize()
    return n_items / (time.perf_counter() - t0)


def cfg4_block(dev, in_flight, items=12):
    """BASELINE.json configs[3]: Tanks&Temples-shaped eval, 1920x1056, metric depth range, 7 views (config.py:119 of the reference
    sets nviews per set; load/tankseval.py:36 crops to 1920x1056) and 11 views.  Reported NEXT to the headline (rank 0, N=1)."""
    from mdfnet_hip import synth, hostmirror
    model = build(dev)
    rec = {"workload": "Tanks&Temples-shaped eval 1920x1056, depth range [0.5, 10], hypotheses (48,24,8), batch 1, synthetic tensors, "
                       "seeded weights; fresh camera tensors every item, images resident", "items_timed": items, "dtype": "f32"}
    with torch.no_grad():
        for nv in (7, 11):
            imgs = synth.make_images(1920, 1056, nv, batch=1, seed=5).to(dev)
            intr, extr, dr = synth.make_cameras(1920, 1056, nv, batch=1, rot_deg=2.0, seed=6, depth_range=(0.5, 10.0), baseline=0.25)

            def item(i, imgs=imgs, cams=(extr, intr, dr)):
                host = tuple(t.clone() for t in cams)
                devs = tuple(t.to(dev, non_blocking=True) for t in host)
                for d_, h_ in zip(devs, host):
                    hostmirror.put(d_, h_)
                return lambda: model(imgs, *devs)
            one = _throughput(dev, item, items, 1, 3)
            many = _throughput(dev, item, items, in_flight, 2 * in_flight + 1) if in_flight > 1 else one
            rec[f"{nv}_views"] = {"views_per_s": round(many, 2), "ms_per_view": round(1e3 / many, 3), "items_in_flight": in_flight,
                                  "one_at_a_time_views_per_s": round(one, 2)}
            del imgs
    rec["peak_memory_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    return rec


def scan_cameras(width, height, nviews, seed=7, rot_deg=2.0, step=40.0):
    """One synthetic DTU-like scan: `nviews` cameras on a line, 40 mm apart, small seeded rotations (so an item's 4 sources have the
    +-40 / +-80 mm baselines of the headline scene) -> intrinsics [1,V,3,3], extrinsics [1,V,4,4], range [1,2]; pair lists as
    pair.txt's: the 10 nearest views, nearest first."""
    import numpy as np
    from mdfnet_hip import synth
    intr, extr, dr = synth.make_cameras(width, height, nviews, batch=1, rot_deg=0.0, seed=seed)
    rng = np.random.RandomState(seed)
    for v in range(nviews):
        e = np.eye(4)
        e[0, 3] = step * (v - nviews // 2)
        e[1, 3] = 3.0 * ((v % 5) - 2)
        e[:3, :3] = synth._rot(rng, rot_deg)
        extr[0, v] = torch.from_numpy(e.astype(np.float32))
    pairs = []
    for r in range(nviews):
        order = sorted((v for v in range(nviews) if v != r), key=lambda v: (abs(v - r), v))
        pairs.append(order[:10])
    return intr, extr, dr, pairs


def cfg5_scan_block(dev, in_flight, nviews=49, nsrc_model=4):
    """BASELINE.json configs[4], one scan's share: every view of a 49-view DTU-shaped scan as the reference view of a 5-view item at
    1600x1184 through eval.py's issue pattern (items in flight, cross-item feature cache: each image goes through the feature pyramid
    once -- SURVEY 8(f) N3), then the consistency filter + fusion of the scan's 49 depth maps against 10 source views each, one fused
    launch per reference view (mdf_consistency_fuse_fwd; reference: tools/filter/dynamic_filter_gpu.py:12-164).  File IO excluded,
    images resident, cameras fresh per item.  Reported NEXT to the headline (rank 0, N=1)."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "mdf-net_amd"))
    from eval import FeatureCache
    from mdfnet_hip import synth, hostmirror, ops
    model = build(dev)
    imgs_all = synth.make_images(WIDTH, HEIGHT, nviews, batch=1, seed=9).to(dev)
    intr, extr, dr, pairs = scan_cameras(WIDTH, HEIGHT, nviews)
    depth = [None] * nviews
    conf = [None] * nviews

    def run_scan(cache_on, flight):
        from mdfnet_hip.pipeline import InFlight
        cache = FeatureCache() if cache_on else None

        def done(tag, out):
            depth[tag], conf[tag] = out["depth"][0], out["confidence"][0]
        pipe = InFlight(dev, flight, done=done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for r in range(nviews):
                ids = [r] + pairs[r][:nsrc_model]
                host = (extr[:, ids].clone(), intr[:, ids].clone(), dr.clone())
                devs = tuple(t.to(dev, non_blocking=True) for t in host)
                for d_, h_ in zip(devs, host):
                    hostmirror.put(d_, h_)
                im = imgs_all[:, ids]
                if cache_on:
                    pipe.submit(lambda im=im, c=devs, k=[("scan", v) for v in ids]: model(im, *c, feature_cache=cache, view_keys=k), tag=r, keep=devs)
                else:
                    pipe.submit(lambda im=im, c=devs: model(im, *c), tag=r, keep=devs)
            pipe.drain()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run_scan(True, in_flight)                         # warm-up (allocator pools of every stream, the cache's steady state)
    t_plain = run_scan(False, in_flight)
    t_cache = run_scan(True, in_flight)

    def run_filter():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kept = 0.0
        res = []
        for r in range(nviews):
            src = pairs[r]
            res.append(ops.consistency_fuse(depth[r], conf[r], intr[0, r], extr[0, r], [depth[v] for v in src],
                                            [intr[0, v] for v in src], [extr[0, v] for v in src]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kept = float(torch.stack([x["final_mask"].float().mean() for x in res]).mean())
        return dt, kept
    run_filter()
    ops.profile_begin()
    t_filter, kept = run_filter()
    launches = [x for x in ops.profile_end() if x[0] == "mdf_consistency_fuse_fwd"]
    k_us = 1e3 * sum(x[2] for x in launches) / max(1, len(launches))
    k_bytes = launches[0][3]["bytes"] if launches else 0.0
    return {"workload": f"one DTU-shaped scan: {nviews} reference views x 5-view items at {WIDTH}x{HEIGHT} (hypotheses 48/24/8) + consistency "
                        f"filter of the {nviews} depth maps against 10 source views each; synthetic tensors, seeded weights, no file IO",
            "model": {"views_per_s": round(nviews / t_cache, 2), "ms_per_view": round(1e3 * t_cache / nviews, 3),
                      "feature_cache": True, "items_in_flight": in_flight,
                      "without_feature_cache_views_per_s": round(nviews / t_plain, 2)},
            "filter": {"us_per_view_wall": round(1e6 * t_filter / nviews, 1), "kernel_us_per_view": round(k_us, 1),
                       "kernel_gbs": round(k_bytes / (k_us * 1e-6) / 1e9, 1) if k_us else None,
                       "kernel_frac_of_hbm_peak": round(k_bytes / (k_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if k_us else None,
                       # its real roof (profiles/r05_filter_pmc.md): 267 vector instructions per (pixel, source view) -- seven IEEE divides and a
                       # sqrt that bit-exactness needs -- against the 1024 SIMD-32 units' issue rate at 2.4 GHz (78.6 T lane-ops/s)
                       "kernel_frac_of_vector_issue_peak": round(267.0 * WIDTH * HEIGHT * 10 / (k_us * 1e-6) / (1024 * 32 * 2.4e9), 4) if k_us else None,
                       "algorithmic_bytes_per_view": round(k_bytes), "nsrc": 10, "final_mask_keeps": round(kept, 4),
                       "note": "depth maps are the model's own (random weights: no two views agree, so the masks are nearly empty and "
                               "the gathers incoherent); wall includes the host-side matrix set-up and mask conversions per view"},
            "scan_views_per_s": round(nviews / (t_cache + t_filter), 2), "dtype": "f32"}


def training_ddp_block(dev, world, rank, steps, blocks):
    """BASELINE.json configs[2] over N ranks (called by EVERY rank when WORLD_SIZE > 1): data-parallel training step at 768x576x5,
    batch 1 per rank = global batch N, the gradients of all replicas averaged by ONE collective over the flat 4.83-MB bucket per step
    (mdfnet_hip/ddp.py; RCCL over xGMI when the backend is nccl) -- the data-parallel exchange of the reference's
    nn.DataParallel (train.py:24-26).  Each block = EXACTLY `steps` steps between barrier + synchronize brackets, MAX over ranks.
    Reports the aggregate samples/s, the collective's time inside the step (HIP events around it on the launch stream: includes
    waiting for the slowest rank), the collective alone, and the same step with the two-step direct exchange
    (MDF_GRAD_EXCHANGE=direct: all-to-all of shards + all-gather of the owners' sums) as a second figure."""
    import statistics
    from mdfnet_hip import synth, ddp
    from mdfnet_hip.optim import FlatAdam
    from net import loss as loss_mod
    model = build(dev).train()
    bucket = ddp.FlatBucket(model)
    bucket.broadcast_parameters(0)
    opt = FlatAdam(bucket, lr=1e-3)
    crit = loss_mod.Loss().to(dev)
    imgs, extr, intr, dr = (t.to(dev) for t in synth.make_scene(TRAIN_W, TRAIN_H, TRAIN_V, batch=1, rot_deg=2.0, seed=3 + rank))
    g = torch.Generator(device="cpu").manual_seed(11 + rank)
    gt = {str(k): (torch.rand(1, TRAIN_H >> k, TRAIN_W >> k, generator=g) * 400 + 480).to(dev) for k in (3, 2, 1, 0)}
    on_gpu = dist.get_backend() == "nccl"

    def step(mode, ev=None):
        out = model(imgs, extr, intr, dr)
        loss = crit(out, gt, dr)
        bucket.zero_grad()
        loss.backward()
        if mode is None:
            bucket.gather()                              # no exchange: what one rank does alone
        else:
            if ev is not None:
                ev[0].record()
            bucket.allreduce_gradients(mode)
            if ev is not None:
                ev[1].record()
        opt.step()
        return loss.detach()

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev if on_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(mode):
        for _ in range(3):
            last = step(mode)
        times, coll = [], []
        for _ in range(blocks):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                last = step(mode, evs[i] if mode is not None else None)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            times.append(max_over_ranks((time.perf_counter() - t0) / steps))
            if mode is not None:
                coll.append(sum(a.elapsed_time(b) for a, b in evs) / steps)
        assert torch.isfinite(last).item()
        med = statistics.median(times)
        r = {"ms_per_step": round(1e3 * med, 3), "samples_per_s": round(world / med, 2),
             "ms_per_step_blocks": [round(1e3 * t, 3) for t in times], "loss_last": round(float(last), 3)}
        if coll:
            r["exchange_ms_in_step"] = round(max_over_ranks(statistics.median(coll)), 3)
        return r

    def collective_alone(mode, reps=50):
        for _ in range(5):
            bucket.allreduce_gradients(mode)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            bucket.allreduce_gradients(mode)
        torch.cuda.synchronize()
        return round(1e3 * max_over_ranks((time.perf_counter() - t0) / reps), 4)

    rec = {"workload": f"BlendedMVS-shaped data-parallel training step {TRAIN_W}x{TRAIN_H}, {TRAIN_V} views, batch 1 per rank = global batch "
                       f"{world} (BASELINE configs[2]): forward + masked smooth-L1 loss + backward + ONE flat-bucket gradient exchange "
                       f"({bucket.flat.numel()} fp32 = {bucket.flat.numel() * 4 / 1e6:.2f} MB) + Adam on every rank; synthetic tensors, seeded weights",
           "n_gpus": world, "backend": ("rccl" if on_gpu else dist.get_backend()), "steps": steps, "blocks": blocks, "dtype": "f32",
           "scaling": "weak", "timing": "per block: barrier + synchronize, K steps, synchronize + barrier; max over ranks; median block"}
    rec["allreduce"] = timed("allreduce")
    rec["allreduce"]["collective_alone_ms"] = collective_alone("allreduce")
    rec["samples_per_s"] = rec["allreduce"]["samples_per_s"]
    rec["ms_per_step"] = rec["allreduce"]["ms_per_step"]
    rec["no_exchange"] = timed(None)
    try:
        rec["direct"] = timed("direct")
        rec["direct"]["collective_alone_ms"] = collective_alone("direct")
    except Exception as e:      # noqa: BLE001  (a backend without all_to_all on device tensors: reported, the default figure stands)
        rec["direct"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return rec


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
    ap.add_argument("--blocks", type=int, default=0,
                    help="the K-step timed block is repeated this many times (each one barrier + synchronize bracketed); the "
                         "MEDIAN block is reported, all of them and their spread under `blocks_ms_per_step`.  0 (default) = as many "
                         "as it takes for >= 1 s of timed work in total, at least 3, at most 41 (the driver's --steps 20 is 0.08 s a block)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="self-launched multi-rank runs (plain `python bench.py --gpus N`): seconds after which all ranks are stopped")
    ap.add_argument("--no-training", action="store_true", help="skip the BASELINE configs[2] training-step block (N=1: one GPU's step; "
                                                               "N>1: the data-parallel step with the gradient exchange, all ranks)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the cfg4 (1920x1056, 7 / 11 views) and cfg5_scan (49-view scan + "
                                                                    "consistency filter) blocks (rank 0, N=1)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--train-stock-steps", type=int, default=2, help="steps of the stock PyTorch-ROCm autograd baseline (0 = skip)")
    ap.add_argument("--no-train-graph", action="store_true", help="skip the hipGraph-replayed variant of the training step (a child process)")
    ap.add_argument("--train-graph-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before this process touches the GPU (no HIP call
        # has happened yet; the children are fresh subprocesses, nothing is exec'ed over an initialised process)
        from mdfnet_hip import shard
        raise SystemExit(shard.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:], timeout=args.rank_timeout))
    if args.train_graph_child:
        if not torch.cuda.device_count():
            raise SystemExit("bench.py needs an MI355X (no GPU visible)")
        training_graph_child(args.train_steps, min(3, max(1, args.blocks if args.blocks > 0 else 3)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:      # own share of the host cores, set in-process before the first HIP call (is_available() below is one)
        from mdfnet_hip import shard
        shard.pin_rank_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
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
        block_dts = []
        n_blocks = args.blocks if args.blocks > 0 else 3
        bi = 0
        while bi < n_blocks:
            bi += 1
            # one block = EXACTLY K steps bracketed by barrier + synchronize on both sides, max over ranks
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
            bdt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([bdt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                bdt = float(t.item())
            block_dts.append(bdt)
            if args.blocks <= 0 and bi == 1:                 # auto: >= 1 s of timed work (every rank computes the same count from the max-over-ranks time)
                n_blocks = min(41, max(3, int(math.ceil(1.0 / max(bdt, 1e-4)))) | 1)
        dt = sorted(block_dts)[len(block_dts) // 2]          # the median block
        args.blocks = len(block_dts) if args.blocks <= 0 else args.blocks      # the other legs (serial view, training) take 3
        out = last["out"]
        # the same K steps strictly one at a time (latency view of the same work), rank 0 only, outside the timed region
        dt_serial = None
        if rank == 0 and args.in_flight > 1:
            ser = []
            for _ in range(min(3, max(1, args.blocks))):
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(args.steps):
                    model(inputs[0], *fresh_cameras())
                torch.cuda.synchronize()
                ser.append(time.perf_counter() - ts)
            dt_serial = sorted(ser)[len(ser) // 2]
    assert torch.isfinite(out["depth"]).all()

    kernels, cpu, prof_info, training = None, None, None, None
    if rank == 0 and not args.no_profile:
        kernels, prof_info = profile_pass(model, inputs)
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra_configs:
        torch.cuda.empty_cache()
        extra["cfg4"] = cfg4_block(dev, args.in_flight)
        torch.cuda.empty_cache()
        extra["cfg5_scan"] = cfg5_scan_block(dev, args.in_flight)
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_training:
        training = training_block(dev, args.train_steps, min(3, max(1, args.blocks)), args.train_stock_steps)
        if not args.no_train_graph:
            training["graph_replay"] = training_graph(args.train_steps, min(3, max(1, args.blocks)))
        primary_training_figure(training)
    if world > 1 and not args.no_training:
        barrier()                      # rank 0's profile pass is over: every rank enters the training leg together
        training = training_ddp_block(dev, world, rank, args.train_steps, min(3, max(1, args.blocks)))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        views_per_s = world * args.steps / dt
        rec = {"metric": "views/sec at DTU 1600x1200x5-view x4-scale", "value": round(views_per_s, 3), "unit": "views/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "items_in_flight": args.in_flight,
               "blocks_ms_per_step": {"all": [round(1e3 * b / args.steps, 3) for b in block_dts], "reported": "median",
                                      "min": round(1e3 * min(block_dts) / args.steps, 3), "max": round(1e3 * max(block_dts) / args.steps, 3),
                                      "spread_pct": round(100.0 * (max(block_dts) - min(block_dts)) / dt, 2), "timed_s": round(sum(block_dts), 3)},
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
            traffic, tsource = measured_traffic("eval")
            attach_traffic(kernels, traffic, tsource, (("conv", "mfma_conv"), ("warp", "warp_aggregate"), ("prob", "prob_head")))
            mf = [k for k in kernels if k["bound"] == "mfma" and "achieved" in k]
            if mf:
                ms = sum(k["ms_per_step"] for k in mf)
                gf = sum(k["algorithmic_gflop_per_step"] for k in mf)
                ach = gf / ms
                rec["roofline"] = {"kernel": "fp32-MFMA implicit-GEMM conv family (conv_lds_kernel + wino3d_kernel + wino2d_kernel + conv3d_kernel + convtr_all_kernel + conv_pair_valu_kernel + res_pair_kernel + conv1x1_kernel + conv1x1_heads_kernel + refine_tail_kernel + prob_fused_kernel)", "bound": "mfma",
                                   "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "traffic": (round(traffic["mfma_conv"]["hbm_bytes_per_forward"]) if "mfma_conv" in traffic else None),
                                   "traffic_unit": f"HBM bytes per step over all launches of the family (PMC, offline: {tsource})",
                                   "instrumentation": prof_info,
                                   "mode": "one view at a time on one stream (profile pass after the timed region: per-launch HIP events "
                                           "on the launch stream); compare with one_at_a_time, not with the in-flight headline",
                                   "ms_per_step": round(ms, 3),
                                   "launches_per_step": sum(k["launches_per_step"] for k in mf),
                                   "algorithmic_gflop_per_step": round(gf, 1)}
                rp = rocprof_conv_family()
                if rp:
                    rp["achieved"] = round(gf / rp["ms_per_step"], 2)
                    rp["frac"] = round(gf / rp["ms_per_step"] / PEAK_FP32_MFMA_TFLOPS, 4)
                    if rp["launches_per_step"] != rec["roofline"]["launches_per_step"]:
                        rp["warning"] = (f"launch count differs from the live pass ({rp['launches_per_step']} vs "
                                         f"{rec['roofline']['launches_per_step']}): the committed profile is of another tree or a "
                                         "kernel of the family is missing from mdfnet_hip/kernel_families.py")
                        print("bench.py: WARNING " + rp["warning"], file=sys.stderr)
                    rec["roofline"]["rocprof"] = rp
            rec["kernels"] = kernels
            rec["hip_kernels_ms_per_step"] = round(sum(k["ms_per_step"] for k in kernels), 3)
        if training:
            rec["training"] = training
        rec.update(extra)
        if cpu:
            rec["cpu_baseline"] = cpu
        print(json.dumps(rec), flush=True)
    if world > 1:
        barrier()                      # rank 0's profile pass is over: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
