"""ctypes binding of libmdfnet_hip.so (C ABI: include/mdfnet_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is
raised.  Build it with `python -m mdfnet_hip.build` (or __graft_entry__.build()).
"""
import ctypes
import os
import threading

# torch bundles its own libamdhip64; it must be loaded BEFORE libmdfnet_hip.so so that the library's
# libamdhip64.so.7 dependency resolves to the SAME runtime instance (two HIP runtimes in one process do not
# share devices/streams: "no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDF_HIP_LIB") or os.path.join(_PKG, "libmdfnet_hip.so")     # (MDF_HIP_LIB: dev A/B of two builds on one box)
ABI_VERSION = 1

_lib = None
_lock = threading.Lock()

c_fp = ctypes.c_void_p  # device pointers are passed as integers (tensor.data_ptr())
c_int = ctypes.c_int
c_i64 = ctypes.c_int64

# name -> (restype, argtypes); mirrors include/mdfnet_hip.h one to one
SIGNATURES = {
    "mdf_abi_version": (c_int, []),
    "mdf_last_error": (ctypes.c_char_p, []),
    "mdf_last_launch": (ctypes.c_char_p, []),
    "mdf_release_stream": (c_int, [c_fp]),
    "mdf_homo_warp_fwd": (c_int, [c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_int] + [c_int] * 5 + [c_fp]),
    "mdf_warp_corner_indices": (c_int, [c_fp, c_fp, c_int, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_warp_aggregate_vec_fwd": (c_int, [c_fp, ctypes.POINTER(c_fp), c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_int]
                                   + [c_int] * 7 + [c_fp]),
    "mdf_warp_aggregate_var_fwd": (c_int, [c_fp, ctypes.POINTER(c_fp), c_int, c_fp, c_fp, c_int, c_fp, c_int]
                                   + [c_int] * 6 + [c_fp]),
    "mdf_conv3d_fwd": (c_int, [c_fp] * 6 + [c_int] * 9 + [c_fp]),
    "mdf_conv3d_packed_size": (c_i64, [c_int, c_int]),
    "mdf_conv3d_pack_weights": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "mdf_conv2d_fwd": (c_int, [c_fp] * 5 + [ctypes.c_float, c_fp, c_fp] + [c_int] * 10 + [c_fp]),
    "mdf_conv2d_pair_fwd": (c_int, [c_fp] * 8 + [c_int] * 3 + [c_fp]),
    "mdf_conv1x1_heads_fwd": (c_int, [c_fp, c_int, ctypes.POINTER(c_fp), ctypes.POINTER(c_fp), ctypes.POINTER(c_fp), ctypes.POINTER(c_fp),
                                      ctypes.POINTER(c_int)] + [c_int] * 4 + [c_fp]),
    "mdf_refine_head_fwd": (c_int, [c_fp] * 5 + [c_int] * 3 + [c_fp]),
    "mdf_conv2d_res_pair_fwd": (c_int, [c_fp] * 3 + [ctypes.c_float, c_fp] + [c_int] * 3 + [c_fp]),
    "mdf_refine_tail_fwd": (c_int, [c_fp] * 6 + [c_int] * 3 + [c_fp]),
    "mdf_conv_packed_size": (c_i64, [c_int, c_int, c_int]),
    "mdf_conv_pack_weights": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp]),
    "mdf_prob_softmax_regress_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "mdf_prob_from_partials_fwd": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_prob_fused_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "mdf_depth_regress_fwd": (c_int, [c_fp, c_fp, c_int, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_confidence_fwd": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_confidence_up2_fwd": (c_int, [c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_range_affine_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_i64, c_fp]),
    "mdf_hypos_fit_fwd": (c_int, [c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_consistency_fuse_fwd": (c_int, [c_fp, c_fp, ctypes.POINTER(c_fp), c_fp, c_int, c_int, c_int, ctypes.c_float, c_int,
                                         ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "mdf_bn_stats_fwd": (c_int, [c_fp, c_i64, c_int, c_int, c_fp, c_fp]),
    "mdf_bn_finalize_fwd": (c_int, [c_fp, c_fp, c_fp, ctypes.c_float, ctypes.c_float, c_i64, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "mdf_bn_relu_apply_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_fp]),
    "mdf_bn_finalize_apply_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_float, ctypes.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_int, c_fp]),
    "mdf_bn_relu_bwd_reduce": (c_int, [c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_fp, c_fp]),
    "mdf_bn_relu_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_fp]),
    "mdf_conv3d_wgrad_workspace": (c_i64, [c_int] * 6),
    "mdf_conv3d_wgrad": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp]),
    "mdf_conv3d_train_fwd": (c_int, [c_fp] * 4 + [c_int] * 9 + [c_fp] * 3 + [c_int, c_fp]),
    "mdf_conv2d_train_fwd": (c_int, [c_fp] * 3 + [c_int] * 9 + [c_fp] * 3 + [c_int, c_int, c_fp]),
    "mdf_conv3d_wgrad_partial": (c_int, [c_fp] * 4 + [c_int] * 7 + [ctypes.POINTER(c_int), c_fp]),
    "mdf_conv2d_wgrad_partial": (c_int, [c_fp] * 4 + [c_int] * 7 + [ctypes.POINTER(c_int), c_fp]),
    "mdf_wgrad_sum_batch": (c_int, [ctypes.POINTER(c_fp), ctypes.POINTER(c_fp), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_fp]),
    "mdf_wgrad_batch_begin": (c_int, []),
    "mdf_wgrad_batch_flush": (c_int, [c_fp]),
    "mdf_conv2d_wgrad_workspace": (c_i64, [c_int] * 6),
    "mdf_conv2d_wgrad": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp]),
    "mdf_pack_job_bytes": (c_i64, []),
    "mdf_pack_job_fill": (c_i64, [c_fp, c_int, c_fp, c_fp] + [c_int] * 9),
    "mdf_pack_batch": (c_int, [c_fp, c_fp, c_int, c_fp]),
    "mdf_upsample2_bilinear_bwd": (c_int, [c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "mdf_prob_softmax_regress_bwd": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp] + [c_int] * 4 + [c_fp]),
    "mdf_prob_conv_dgrad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "mdf_prob_conv_dgrad_stat": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 5 + [c_fp, c_fp, c_fp, c_int, c_fp]),
    "mdf_warp_aggregate_vec_train": (c_int, [c_int, c_fp, ctypes.POINTER(c_fp), c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                             c_fp, ctypes.POINTER(c_fp), c_fp, c_fp] + [c_int] * 7 + [c_fp]),
    "mdf_aggregate_train_prepare": (c_int, [c_fp] * 4 + [c_i64, c_int, c_int, c_fp, c_fp, c_int, c_fp]),
    "mdf_aggregate_train_finalize": (c_int, [c_fp] * 3 + [ctypes.c_float, ctypes.c_float, c_i64, c_int, c_int] + [c_fp] * 5),
    "mdf_aggregate_train_bwd_finalize": (c_int, [c_fp, c_fp, c_int, c_i64, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp]),
    "mdf_masked_smooth_l1_reduce": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_i64, c_fp, c_fp]),
    "mdf_masked_smooth_l1_finalize": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp]),
    "mdf_masked_smooth_l1_bwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_i64, c_fp, c_fp, c_fp, c_fp]),
    "mdf_masked_smooth_l1_reduce_multi": (c_int, [ctypes.POINTER(c_fp), ctypes.POINTER(c_fp), ctypes.POINTER(c_i64), c_int, c_fp, c_int, c_int, c_int, c_fp, c_fp]),
    "mdf_masked_smooth_l1_bwd_multi": (c_int, [ctypes.POINTER(c_fp), ctypes.POINTER(c_fp), ctypes.POINTER(c_i64), c_int, c_fp, c_int, c_int, c_int, c_fp, c_fp,
                                               ctypes.POINTER(c_fp), c_fp]),
    "mdf_fpn_compose_fwd": (c_int, [c_fp] * 6 + [c_int] * 3 + [c_fp, c_fp]),
    "mdf_fpn_compose_bwd": (c_int, [c_fp] * 14 + [c_int] * 3 + [c_fp] * 7),
    "mdf_adam_job_bytes": (c_i64, []),
    "mdf_adam_job_fill": (c_i64, [c_fp, c_int, c_fp, c_i64, c_i64, c_int]),
    "mdf_adam_step": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp] + [ctypes.c_float] * 5 + [c_i64, c_fp]),
    "mdf_adam_step_hyper": (c_int, [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp] + [ctypes.c_float] * 4 + [c_fp]),
    "mdf_hypos_from_fit_fwd": (c_int, [c_int, c_fp, c_fp, c_fp, ctypes.c_float, c_fp] + [c_int] * 5 + [c_fp]),
}


class MdfHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise MdfHipError(
                        f"{LIB_PATH} not found: the HIP extension is not built and there is no fallback path. "
                        "Run `python -m mdfnet_hip.build` (needs hipcc, --offload-arch=gfx950).")
                h = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(h, name)  # AttributeError if the .so is stale
                    fn.restype = res
                    fn.argtypes = args
                if h.mdf_abi_version() != ABI_VERSION:
                    raise MdfHipError(f"ABI version mismatch: library {h.mdf_abi_version()}, binding {ABI_VERSION}")
                _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mdf_last_error().decode("utf-8", "replace")
        raise MdfHipError(f"{what} failed with code {rc}: {msg}")
