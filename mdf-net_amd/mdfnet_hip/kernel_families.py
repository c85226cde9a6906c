"""Which measurement family every `__global__` kernel of csrc/*.hip belongs to -- the ONE table behind bench.py's rocprof
cross-check, scripts/summarize_traffic.py (PMC bytes per family) and scripts/summarize_rocprof.py ("hand-written" time).

A kernel is looked up by its exact function name (templates, namespaces and the parameter list of a rocprof `Name` column are
stripped first), never by substring: round 3 lost `convtr_all_kernel` from three substring filters after a rename and reported
the conv family 6.6 % too fast.  tests/test_kernel_families_cpu.py asserts that the keys of KERNEL_FAMILY are exactly the
`__global__` names in csrc/, so a new or renamed kernel fails the CPU suite until it is classified here."""
import re

# family keys (the names the profiles/*.json files carry)
MFMA_CONV = "mfma_conv"                    # everything reached through mdf_conv{2d,3d}[_train]_fwd, prob_fused, refine_tail
WARP = "warp_aggregate"                    # eval warp + aggregation
WARP_TRAIN = "aggregate_train_passes"
WARP_SCATTER = "aggregate_scatter"
WGRAD = "wgrad"
BN = "batchnorm_train"
PROB = "prob_head"
HEADS = "heads"
FILTER = "consistency_filter"
CONTROL = "control"                        # weight packing, Adam, loss, finalisers: launch-bound one-block kernels

KERNEL_FAMILY = {
    # conv_lds.hip / conv3d.hip / conv_pair.hip / conv1x1.hip / refine_tail.hip / prob_fused.hip
    "conv_lds_kernel": MFMA_CONV, "conv3d_kernel": MFMA_CONV, "conv3d_wlds_kernel": MFMA_CONV, "convtr_all_kernel": MFMA_CONV, "convtr_cls_kernel": MFMA_CONV, "conv_pair_kernel": MFMA_CONV, "conv_pair_valu_kernel": MFMA_CONV,   # (the pair on packed fp32 FMAs: same layers, same family)
    "conv1x1_kernel": MFMA_CONV, "conv1x1_heads_kernel": MFMA_CONV, "refine_tail_kernel": MFMA_CONV, "prob_fused_kernel": MFMA_CONV, "res_pair_kernel": MFMA_CONV,
    "wino3d_kernel": MFMA_CONV, "wino2d_kernel": MFMA_CONV,
    "pack_weights_kernel": CONTROL, "pack_batch_kernel": CONTROL,
    # warp_aggregate.hip
    "warp_kernel": WARP, "warp_vec8_kernel": WARP, "warp_vec_win_kernel": WARP, "corner_index_kernel": WARP,
    # warp_aggregate_train.hip
    "warp_train_kernel": WARP_TRAIN, "warp_bwd_kernel": WARP_SCATTER,
    "agg_prepare_kernel": CONTROL, "agg_finalize_kernel": CONTROL, "agg_bwd_finalize_kernel": WARP_SCATTER,
    # wgrad.hip / wgrad_lds.hip
    "wgrad_kernel": WGRAD, "wgrad_a1_kernel": WGRAD, "wgrad_a1_valu_kernel": WGRAD, "wgrad2d_kernel": WGRAD,
    "wgrad_lds_kernel": WGRAD, "wgrad_lds_batch_kernel": WGRAD, "slab_sum_kernel": WGRAD, "slab_sum_batch_kernel": WGRAD,
    # bn_train.hip
    "bn_reduce_kernel": BN, "bn_finalize_kernel": BN, "bn_relu_apply_kernel": BN, "bn_finalize_apply_kernel": BN,
    "bn_relu_bwd_kernel": BN,
    # prob_head.hip / prob_bwd.hip / upsample_bwd.hip
    "prob_head_kernel": PROB, "prob_head_tiled_kernel": PROB, "prob_from_partials_kernel": PROB,
    "softmax_regress_bwd_kernel": PROB, "prob_conv_dgrad_kernel": PROB, "upsample2_bwd_kernel": PROB,
    # regress.hip / fpn_compose.hip
    "depth_regress_kernel": HEADS, "confidence_kernel": HEADS, "confidence_up2_kernel": HEADS, "range_affine_kernel": HEADS,
    "hypos_fit_kernel": HEADS, "hypos_from_fit_kernel": HEADS, "refine_head_kernel": HEADS, "fpn_compose_fwd_kernel": HEADS, "fpn_compose_bwd_kernel": HEADS,
    # consistency.hip
    "consistency_fuse_kernel": FILTER,
    # loss.hip
    "masked_smooth_l1_reduce_kernel": CONTROL, "masked_smooth_l1_finalize_kernel": CONTROL, "masked_smooth_l1_bwd_kernel": CONTROL,
    "masked_smooth_l1_reduce_multi_kernel": CONTROL, "masked_smooth_l1_bwd_multi_kernel": CONTROL, "adam_step_kernel": CONTROL,
}

def function_name(rocprof_name):
    """`void (anonymous namespace)::conv_lds_kernel<16, 8, ...>(LdsConvParams)` -> `conv_lds_kernel`."""
    s = rocprof_name.strip()
    if s.startswith("void "):
        s = s[5:]
    s = s.replace("(anonymous namespace)::", "")
    head = re.split(r"[<(]", s, maxsplit=1)[0]
    return head.split("::")[-1].strip()


def family(rocprof_name):
    """Family of a kernel as rocprofv3 names it, or None for a kernel that is not one of csrc/'s (ATen, MIOpen, RCCL ...)."""
    return KERNEL_FAMILY.get(function_name(rocprof_name))


def is_ours(rocprof_name):
    return function_name(rocprof_name) in KERNEL_FAMILY


def globals_in_sources(csrc_dir):
    """Every `__global__` function name declared in csrc_dir/*.hip (the test's ground truth)."""
    import os
    names = {}
    for f in sorted(os.listdir(csrc_dir)):
        if not f.endswith(".hip"):
            continue
        text = open(os.path.join(csrc_dir, f)).read()
        for m in re.finditer(r"__global__\b[\s\S]*?\bvoid\s+([A-Za-z_][A-Za-z_0-9]*)\s*\(", text):
            names.setdefault(m.group(1), f)
    return names


def fetch_size_factor(rocprof_name):
    """FETCH_SIZE (KiB) -> bytes correction on gfx950: 2 for every kernel of this repository.  The counter tallies a 128-B request as
    64 B, so any access whose lanes form contiguous runs of >= 128 B reads HALF its bytes -- streaming 16 B/lane, 128-B and 256-B texel
    gathers, even a dword walk of a cache line all measure 0.500 -- while isolated 64-B requests are counted exactly (1.000:
    scripts/micro/fetch_calib.hip, profiles/r04_fetch_calibration.md).  The only candidates for the exact case are the aggregation
    kernels at C = 16 (64-B texels), but there the texels of neighbouring pixels are neighbours in memory and coalesce: the raw counter
    of `warp_kernel<16,1>` at cfg2 is 77.5 MB per launch against 166.8 MB of algorithmic reads, so it cannot be exact -- factor 2 (0.93x
    of the algorithmic bytes).  Kept as a function so that a kernel with genuinely scattered 64-B reads can be given factor 1."""
    return 2.0
