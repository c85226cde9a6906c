"""C-ABI error conventions on the device (include/mdfnet_hip.h:12-20): bad arguments return MDF_EARG, configurations that
are not built return MDF_EUNSUPPORTED, both with a message in mdf_last_error(); nothing aborts, and the library stays
usable afterwards.  Plus degenerate-but-legal shapes (single plane, single row, tile-smaller-than-kernel)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mdfnet_hip
from mdfnet_hip import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EARG, EUNSUP = -1, -2


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _last():
    return mdfnet_hip.lib().mdf_last_error().decode()


def test_bad_arguments_return_codes_not_aborts():
    lib = mdfnet_hip.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros(1, 8, 8, 16, device=DEV)
    y = torch.zeros(1, 8, 8, 64, device=DEV)
    w = torch.zeros(1 << 16, device=DEV)
    # conv2d: configuration not built
    rc = lib.mdf_conv2d_fwd(_ptr(x), _ptr(w), None, None, None, ctypes.c_float(1.0), None, _ptr(y), 1, 8, 8, 16, 24, 3, 1, 0, 0, 0, st)
    assert rc == EUNSUP and "not built" in _last()
    # conv2d: null pointer, bad shape, pixel shuffle on a wrong Cout
    assert lib.mdf_conv2d_fwd(None, _ptr(w), None, None, None, ctypes.c_float(1.0), None, _ptr(y), 1, 8, 8, 16, 16, 3, 1, 0, 0, 0, st) == EARG
    assert lib.mdf_conv2d_fwd(_ptr(x), _ptr(w), None, None, None, ctypes.c_float(1.0), None, _ptr(y), 1, 0, 8, 16, 16, 3, 1, 0, 0, 0, st) == EARG
    assert lib.mdf_conv2d_fwd(_ptr(x), _ptr(w), None, None, None, ctypes.c_float(1.0), None, _ptr(y), 1, 8, 8, 16, 16, 3, 1, 0, 0, 1, st) == EARG
    assert "pixel_shuffle2" in _last()
    # conv3d: channel count outside the built set
    assert lib.mdf_conv3d_fwd(_ptr(x), _ptr(w), None, None, None, _ptr(y), 1, 2, 4, 4, 24, 16, 1, 0, 0, st) < 0
    assert lib.mdf_conv3d_pack_weights(_ptr(w), _ptr(w), 12, 16, 0, st) == EARG
    # warp + aggregate: C not built, too many source views, C/G != 2
    srcs = (ctypes.c_void_p * 17)(*[x.data_ptr()] * 17)
    proj = torch.zeros(17, 1, 12, device=DEV)
    hyp = torch.ones(1, 4, device=DEV)
    wp = torch.zeros(64, device=DEV)
    cost = torch.zeros(1, 4, 8, 8, 8, device=DEV)
    args = lambda C, G, n: (_ptr(x), srcs, 1, _ptr(proj), _ptr(hyp), 0, _ptr(wp), _ptr(cost), 1, 1, C, G, 4, 8, 8, n, st)
    assert lib.mdf_warp_aggregate_vec_fwd(*args(48, 24, 2)) == EUNSUP
    assert lib.mdf_warp_aggregate_vec_fwd(*args(16, 8, 17)) == EARG and "n_src" in _last()
    assert lib.mdf_warp_aggregate_vec_fwd(*args(16, 4, 2)) == EUNSUP
    # prob head in partial-sum form: D beyond the built register tiles
    assert lib.mdf_prob_from_partials_fwd(_ptr(y), None, 0, _ptr(x), None, 1, 200, 4, 4, st) == EUNSUP
    # the one-launch prob head: channel count not built, depth output without hypotheses, a depth whose logits do not fit the LDS
    assert lib.mdf_prob_fused_fwd(_ptr(x), _ptr(w), None, 0, _ptr(y), None, 1, 4, 8, 8, 32, st) == EUNSUP and "Cin" in _last()
    assert lib.mdf_prob_fused_fwd(_ptr(x), _ptr(w), None, 0, _ptr(y), _ptr(y), 1, 4, 8, 8, 8, st) == EARG
    assert lib.mdf_prob_fused_fwd(_ptr(x), _ptr(w), None, 0, _ptr(y), None, 1, 160, 8, 8, 8, st) == EUNSUP and "LDS" in _last()
    # the refine tail: lo without span, null weights
    assert lib.mdf_refine_tail_fwd(_ptr(x), _ptr(w), _ptr(w), _ptr(wp), None, _ptr(y), 1, 8, 8, st) == EARG and "lo and span" in _last()
    assert lib.mdf_refine_tail_fwd(_ptr(x), None, _ptr(w), None, None, _ptr(y), 1, 8, 8, st) == EARG
    # ... and the library still works
    out = ops.conv2d_nhwc(torch.ones(1, 8, 8, 16, device=DEV), ops.pack_conv2d_weight(torch.ones(16, 16, 3, 3, device=DEV)), 16, 16, 3, 1)
    assert float(out[0, 4, 4, 0]) == 16 * 9
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(1, 1, 5, 7), (1, 2, 1, 9), (2, 3, 4, 1), (1, 1, 1, 1)])
def test_conv3d_degenerate_volumes(shape):
    """Single plane / single row / single column / single voxel: every tap but the centre one may fall outside."""
    b, d, h, w = shape
    g = torch.Generator().manual_seed(d * 100 + h * 10 + w)
    for cin, cout, tr in ((16, 16, False), (8, 8, False), (16, 8, True), (8, 16, "s2")):
        x = torch.randn(b, cin, d, h, w, generator=g)
        if tr is True:
            wt = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.1
            ref = F.conv_transpose3d(x, wt, None, 2, 1, 1)
        else:
            wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
            ref = F.conv3d(x, wt, None, 2 if tr == "s2" else 1, 1)
        wp = ops.pack_conv3d_weight(wt.to(DEV), tr is True)
        y = ops.conv3d_ndhwc(ops.to_ndhwc(x.to(DEV)), wp, cin, cout, 2 if tr else 1, tr is True)
        np.testing.assert_allclose(ops.from_ndhwc(y).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{cin}->{cout} {tr} {shape}")


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 1, 70), (2, 5, 1), (1, 3, 2)])
def test_conv2d_degenerate_images(shape):
    b, h, w = shape
    g = torch.Generator().manual_seed(h * 10 + w)
    for cin, cout, k, s in ((8, 8, 3, 1), (3, 8, 3, 1), (8, 16, 5, 2), (64, 64, 1, 1), (8, 32, 3, 1), (8, 1, 3, 1), (16, 4, 3, 1)):
        x = torch.randn(b, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) * 0.1
        ref = F.conv2d(x, wt, None, s, (k - 1) // 2)
        y = ops.conv2d_nhwc(ops.to_nhwc(x.to(DEV)), ops.pack_conv2d_weight(wt.to(DEV)), cin, cout, k, s)
        np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5, err_msg=f"{cin}->{cout} k{k}s{s} {shape}")
