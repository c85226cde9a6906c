"""CPU ORACLE for the MDF-Net multi-stage MVS hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (own code, functional style over a plain state_dict)
of the algorithm in the reference's net/core.py + net/unit/*.py.  Nothing in the
shipped product imports it: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may (as the checker / the timed CPU baseline, never as the
product path).

Parity status: PINNED.  The reference has no tests/goldens of its own
(SURVEY.md section 4), so the oracle is pinned by outputs of the reference itself
run in the build container: oracle/gen_golden.py imports /root/reference/net,
loads the deterministic weights of mdfnet_hip.synth.seeded_state_dict and dumps
tests/golden/*.npz; tests/test_oracle_golden.py checks every function below
against those vectors.

Third-party arithmetic (L0 of the reference = PyTorch ATen ops) is used here
exactly where the reference uses it (conv3d/conv2d/batch_norm/softmax/
grid_sample/interpolate/inverse/matmul); in addition the warp is restated with
explicit index arithmetic (`warp_positions`/`warp_corners`/`homo_warping_explicit`)
whose rounding order was determined empirically to be BIT-IDENTICAL to
torch-2.10 CPU `matmul` + `grid_sample` (see the functions' docstrings); that
explicit form is what the HIP kernel is compared with for "indexing bit-exact".

All citations are reference paths relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
# Working precision of every function below: float32 = the reference's arithmetic (what the goldens pin).  `precision(torch.float64)`
# runs the SAME restatement in float64 -- the yardstick of the training-parity tests (how far is an fp32 implementation from the
# exact result of this algorithm on these inputs?), never a parity target of its own.
WORK = torch.float32


class precision:
    """with precision(torch.float64): every cast / constant of the oracle uses that dtype (inputs and weights are the caller's)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global WORK
        self.prev, WORK = WORK, self.dtype
        return self

    def __exit__(self, *exc):
        global WORK
        WORK = self.prev


# --------------------------------------------------------------------------- helpers
def _fma(a, b, c):
    """fp32 fused multiply-add emulated through fp64 (product of two f32 is exact in f64;
    the single f64 add + final f32 rounding differs from a true fma only in ~2^-29 of cases)."""
    return (a.double() * b.double() + c.double()).to(WORK)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


BN_TRACE = None      # a list while oracle/train_check.py:bn_trace() is active: (prefix, batch mean, unbiased batch variance) per training-mode call


def _bn(x, p, pre, training):
    """BatchNorm{2,3}d; eval uses running stats, training uses batch stats.  The oracle is functional: the running-stat update of
    nn.BatchNorm is not performed, but the batch statistics it would be made from are recorded in BN_TRACE when that is a list."""
    if training and BN_TRACE is not None:
        with torch.no_grad():
            dims = [0] + list(range(2, x.dim()))
            BN_TRACE.append((pre, x.mean(dims), x.var(dims, unbiased=True)))
    return F.batch_norm(x, None if training else p[pre + "running_mean"],
                        None if training else p[pre + "running_var"],
                        p[pre + "weight"], p[pre + "bias"], training, 0.1, BN_EPS)


# --------------------------------------------------------------------------- a2 scale_cam
def scale_cam(intrinsics, extrinsics, stage):
    """net/unit/scale.py:4-20.  level = 3-stage; K[:2] /= 2^level; P[:3,:4] = K @ E[:3,:4].
    Returns (ref_proj [B,4,4], tuple of V-1 src_proj [B,4,4]); inputs are not mutated."""
    div = float(2 ** (3 - stage))
    k = intrinsics.clone()
    k[:, :, 0:2, :] = k[:, :, 0:2, :] / div
    p = extrinsics.clone()
    p[:, :, :3, :4] = torch.matmul(k, extrinsics[:, :, :3, :4])
    views = p.unbind(1)
    return views[0], tuple(views[1:])


def relative_projection(src_proj, ref_proj):
    """net/unit/base.py:98  proj = src_proj @ inverse(ref_proj)  -> [B,4,4]."""
    return torch.matmul(src_proj, torch.inverse(ref_proj))


# --------------------------------------------------------------------------- a4 homo_warping
def homo_warping(src_fea, src_proj, ref_proj, depth_hypos):
    """net/unit/base.py:85-126, restated with the same ATen calls (the canonical oracle).
    src_fea [B,C,h,w]; hypos [B,D,1,1] | [B,D,h,w]  ->  [B,C,D,h,w]."""
    b, c, h, w = src_fea.shape
    d = depth_hypos.shape[1]
    with torch.no_grad():
        proj = relative_projection(src_proj, ref_proj)
        rot, trans = proj[:, :3, :3], proj[:, :3, 3:4]
        gy, gx = torch.meshgrid(torch.arange(h, dtype=WORK), torch.arange(w, dtype=WORK),
                                indexing="ij")
        pix = torch.stack((gx.reshape(-1), gy.reshape(-1), torch.ones(h * w)))  # [3,hw], integer pixel centres
        rot_xyz = torch.matmul(rot, pix.unsqueeze(0).expand(b, 3, h * w))       # base.py:110
        pts = rot_xyz.unsqueeze(2) * depth_hypos.reshape(b, 1, d, -1)            # base.py:112
        pts = pts + trans.reshape(b, 3, 1, 1)                                    # base.py:114
        xy = pts[:, :2] / pts[:, 2:3]                                            # base.py:115 (no z>0 test)
        xn = xy[:, 0] / ((w - 1) / 2) - 1                                        # base.py:117 align_corners=True style
        yn = xy[:, 1] / ((h - 1) / 2) - 1
        grid = torch.stack((xn, yn), dim=3).reshape(b, d * h, w, 2)
    out = F.grid_sample(src_fea, grid, mode="bilinear", padding_mode="zeros", align_corners=False)  # base.py:122
    return out.reshape(b, c, d, h, w)


def warp_positions(proj, depth_hypos, h, w):
    """Explicit restatement of base.py:99-118 + ATen's CPU grid_sample unnormalise.
    Returns (ix, iy) float32 [B,D,h*w]: sample position in SOURCE pixel units.

    Rounding order (verified bit-identical to torch 2.10 CPU on this image, see
    tests/test_oracle_golden.py::test_explicit_warp_bitwise):
      rot_xyz_i = fma(r_i2, 1, fma(r_i1, y, r_i0 * x))        (MKL sgemm, K=3)
      P = rot_xyz * depth ; P = P + t                         (separate mul, add)
      px = Px / Pz ; py = Py / Pz                             (IEEE divide)
      xn = px / f32((w-1)/2) - 1                              (true divide, then sub)
      ix = fma(xn + 1, w/2, -0.5)                             (ATen vectorised unnormalise, FMA-contracted)
    """
    b, d = depth_hypos.shape[:2]
    gy, gx = torch.meshgrid(torch.arange(h, dtype=WORK), torch.arange(w, dtype=WORK), indexing="ij")
    x, y = gx.reshape(1, -1), gy.reshape(1, -1)
    one = torch.ones_like(x)
    r = proj[:, :3, :3]
    t = proj[:, :3, 3]
    rows = []
    for i in range(3):
        r0, r1, r2 = (r[:, i, j].reshape(b, 1) for j in range(3))
        rows.append(_fma(r2.expand(b, h * w), one.expand(b, h * w),
                         _fma(r1.expand(b, h * w), y.expand(b, h * w), r0 * x)))
    dep = depth_hypos.reshape(b, d, -1)  # [B,D,hw] or [B,D,1]
    px3, py3, pz3 = (rows[i].unsqueeze(1) * dep + t[:, i].reshape(b, 1, 1) for i in range(3))
    px = px3 / pz3
    py = py3 / pz3
    xn = px / torch.tensor((w - 1) / 2, dtype=WORK) - 1
    yn = py / torch.tensor((h - 1) / 2, dtype=WORK) - 1
    ix = _fma(xn + 1, torch.tensor(w / 2, dtype=WORK).expand_as(xn), torch.tensor(-0.5).expand_as(xn))
    iy = _fma(yn + 1, torch.tensor(h / 2, dtype=WORK).expand_as(yn), torch.tensor(-0.5).expand_as(yn))
    return ix, iy


def warp_corners(ix, iy, h, w):
    """Integer corner indices (the 'indexing bit-exact' item) and bilinear weights.
    Returns x0,y0 int32 (floor; clamped to [-2^30, 2^30] and 0x80000000 for non-finite
    positions), weights (nw,ne,sw,se) f32 and the 4 in-bounds masks."""
    x0f, y0f = torch.floor(ix), torch.floor(iy)
    x1f, y1f = x0f + 1, y0f + 1
    wts = ((x1f - ix) * (y1f - iy), (ix - x0f) * (y1f - iy), (x1f - ix) * (iy - y0f), (ix - x0f) * (iy - y0f))
    masks = tuple(((xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1))
                  for xx, yy in ((x0f, y0f), (x1f, y0f), (x0f, y1f), (x1f, y1f)))
    lim = float(2 ** 30)
    fin = torch.isfinite(ix) & torch.isfinite(iy)
    x0 = torch.where(fin, x0f.clamp(-lim, lim), torch.zeros_like(x0f)).to(torch.int32)
    y0 = torch.where(fin, y0f.clamp(-lim, lim), torch.zeros_like(y0f)).to(torch.int32)
    sentinel = torch.tensor(-2 ** 31, dtype=torch.int32)
    x0 = torch.where(fin, x0, sentinel)
    y0 = torch.where(fin, y0, sentinel)
    return x0, y0, wts, masks


def homo_warping_explicit(src_fea, src_proj, ref_proj, depth_hypos):
    """Gather-based warp from the explicit positions; tap accumulation order as ATen:
    out = fma(v_se,w_se, fma(v_sw,w_sw, fma(v_ne,w_ne, v_nw*w_nw))); out-of-bounds taps read 0.
    Non-finite positions give NaN (H4: z == 0 planes)."""
    b, c, h, w = src_fea.shape
    d = depth_hypos.shape[1]
    proj = relative_projection(src_proj, ref_proj)
    ix, iy = warp_positions(proj, depth_hypos, h, w)
    ix = ix.expand(b, d, h * w)
    iy = iy.expand(b, d, h * w)
    x0f, y0f = torch.floor(ix), torch.floor(iy)
    _, _, wts, masks = warp_corners(ix, iy, h, w)
    flat = src_fea.reshape(b, c, 1, h * w).expand(b, c, d, h * w)

    def tap(xf, yf, m):
        xi = torch.nan_to_num(xf, nan=0.0, posinf=0.0, neginf=0.0).clamp(0, w - 1).long()
        yi = torch.nan_to_num(yf, nan=0.0, posinf=0.0, neginf=0.0).clamp(0, h - 1).long()
        idx = (yi * w + xi).unsqueeze(1).expand(b, c, d, h * w)
        return torch.gather(flat, 3, idx) * m.unsqueeze(1)

    taps = (tap(x0f, y0f, masks[0]), tap(x0f + 1, y0f, masks[1]), tap(x0f, y0f + 1, masks[2]),
            tap(x0f + 1, y0f + 1, masks[3]))
    ws = [wt.unsqueeze(1).expand(b, c, d, h * w) for wt in wts]
    acc = taps[0] * ws[0]
    for k in (1, 2, 3):
        acc = _fma(taps[k], ws[k], acc)
    return acc.reshape(b, c, d, h, w)


# --------------------------------------------------------------------------- a5 VectorAggregate
def view_weight(sim, p, training=False):
    """depth_weight head, net/unit/homoaggregate.py:16-20: Conv3d(G->1,1x1x1,no bias) -> BN3d(1)
    -> ReLU -> Conv3d(1->1,bias) -> Sigmoid.   sim [B,G,D,h,w] -> [B,1,D,h,w]."""
    z = F.conv3d(sim, p["depth_weight.0.conv.weight"])
    z = F.relu(_bn(z, p, "depth_weight.0.bn.", training))
    z = F.conv3d(z, p["depth_weight.1.weight"], p["depth_weight.1.bias"])
    return torch.sigmoid(z)


def vector_aggregate(features, ref_proj, src_projs, depth_hypos, ngroups, p, training=False, warp=homo_warping):
    """net/unit/homoaggregate.py:25-46.  features: list of V [B,C,h,w] (view 0 = reference).
    Group-wise softmax over C/G channels of ref and warped src, inner product -> similarity
    [B,G,D,h,w]; learned per-voxel view weight; returns sum(w*sim)/sum(w)."""
    ref = features[0]
    b, c, h, w = ref.shape
    d = depth_hypos.shape[1]
    g = ngroups
    ref_unit = F.softmax(ref.reshape(b, g, c // g, 1, h, w), dim=2)  # identical for every depth plane
    num, den = 0.0, 0.0
    for src, sp in zip(features[1:], src_projs):
        vol = warp(src, sp, ref_proj, depth_hypos).reshape(b, g, c // g, d, h, w)
        sim = (F.softmax(vol, dim=2) * ref_unit).sum(dim=2)
        wgt = view_weight(sim, p, training)
        den = den + wgt
        num = num + wgt * sim
    return num / den


def variance_aggregate(features, ref_proj, src_projs, depth_hypos, warp=homo_warping):
    """net/unit/homoaggregate.py:49-69 (inactive in config.py but named by north_star):
    sum / sum-of-squares over {ref, softmax_C(warped src_v)}; var = E[x^2] - E[x]^2 -> [B,C,D,h,w]."""
    ref = features[0].unsqueeze(2)
    s1, s2 = ref, ref ** 2
    for src, sp in zip(features[1:], src_projs):
        vol = F.softmax(warp(src, sp, ref_proj, depth_hypos), dim=1)
        s1 = s1 + vol
        s2 = s2 + vol ** 2
    n = len(features)
    return s2 / n - (s1 / n) ** 2


# --------------------------------------------------------------------------- a6-a8 regularisers
RELU3_HOOK = None    # test hook (oracle/train_check.py:relu_hook): called with every ReLU input of the regularisers in layer order INSTEAD of F.relu


def _relu3(x):
    return F.relu(x) if RELU3_HOOK is None else RELU3_HOOK(x)


def _cbr3(x, p, pre, stride=1, training=False):
    """ConvBNReLU3D, net/unit/base.py:50-68 (k=3, pad=1, no conv bias)."""
    x = F.conv3d(x, p[pre + "conv.weight"], None, stride, 1)
    return _relu3(_bn(x, p, pre + "bn.", training))


def _tbr3(x, p, pre_conv, pre_bn, training=False):
    """ConvTranspose3d(k3,s2,p1,op1,no bias) + BN + ReLU (regular.py:32-34,38-40,95-108)."""
    x = F.conv_transpose3d(x, p[pre_conv + "weight"], None, 2, 1, 1)
    return _relu3(_bn(x, p, pre_bn, training))


def regular_3scales_logits(x, p, training=False):
    """RegularNet_3Scales up to (excluding) the softmax, net/unit/regular.py:47-67."""
    assert x.shape[-1] % 4 == 0 and x.shape[-2] % 4 == 0
    x = _cbr3(_cbr3(x, p, "conv01.0.", 1, training), p, "conv01.1.", 1, training)
    x1 = _cbr3(x, p, "conv12.0.", 2, training)
    x1 = _cbr3(_cbr3(x1, p, "conv12.1.", 1, training), p, "conv12.2.", 1, training)
    y = _cbr3(x1, p, "conv232.0.", 2, training)
    y = _cbr3(_cbr3(y, p, "conv232.1.", 1, training), p, "conv232.2.", 1, training)
    x1 = x1 + _tbr3(y, p, "conv232.3.", "conv232.4.", training)
    x = x + _tbr3(x1, p, "conv10.0.", "conv10.1.", training)
    return F.conv3d(x, p["prob.weight"], None, 1, 1).squeeze(1)


def regular_4scales_logits(x, p, training=False):
    """RegularNet_4Scales up to (excluding) the softmax, net/unit/regular.py:114-131."""
    assert x.shape[-1] % 8 == 0 and x.shape[-2] % 8 == 0
    x1 = _cbr3(x, p, "conv01.", 1, training)
    x2 = _cbr3(_cbr3(x1, p, "conv12.0.", 2, training), p, "conv12.1.", 1, training)
    x3 = _cbr3(_cbr3(x2, p, "conv23.0.", 2, training), p, "conv23.1.", 1, training)
    y = _cbr3(_cbr3(x3, p, "conv343.0.", 2, training), p, "conv343.1.", 1, training)
    x3 = x3 + _tbr3(y, p, "conv343.2.", "conv343.3.", training)
    x2 = x2 + _tbr3(x3, p, "trconv32.0.", "trconv32.1.", training)
    x1 = x1 + _tbr3(x2, p, "trconv21.0.", "trconv21.1.", training)
    return F.conv3d(x1, p["prob.weight"], None, 1, 1).squeeze(1)


def regular(x, p, training=False):
    """Dispatch on the key set: 3-scale (stage 0, regular.py:9-69) or 4-scale (stages 1,2, :72-133);
    softmax over the depth axis (regular.py:69,133).  cost [B,G,D,h,w] -> prob [B,D,h,w]."""
    fn = regular_3scales_logits if "conv232.0.conv.weight" in p else regular_4scales_logits
    return F.softmax(fn(x, p, training), dim=1)


# --------------------------------------------------------------------------- a9/a10 regression
def depth_regression(prob, depth_hypos):
    """net/unit/regress.py:5-7  soft-argmin."""
    return torch.sum(prob * depth_hypos, 1)


def confidence_regress(prob):
    """net/unit/regress.py:9-25 (n=4, pad=(1,2) along D, last_confidence=None):
    conf = sum(prob[idx-1 .. idx+2]) with idx = trunc(sum_d prob_d * d) (int64)."""
    b, d, h, w = prob.shape
    with torch.no_grad():
        win = 4 * F.avg_pool3d(F.pad(prob.unsqueeze(1), (0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1).squeeze(1)
        ramp = torch.arange(d, dtype=WORK).reshape(1, d, 1, 1).expand(b, d, 1, 1)
        idx = depth_regression(prob, ramp).long()
        return torch.gather(win, 1, idx.unsqueeze(1)).squeeze(1)


def confidence_index(prob):
    """The int64 index used by confidence_regress (regress.py:15-17), exposed for index parity."""
    b, d = prob.shape[:2]
    ramp = torch.arange(d, dtype=WORK).reshape(1, d, 1, 1).expand(b, d, 1, 1)
    return depth_regression(prob, ramp).long()


# --------------------------------------------------------------------------- a3 HyposByFit
def uniform_hypos(depth_range, ndepths):
    """net/unit/depthhypos.py:31-38 -> [B,D,1,1]."""
    b = depth_range.shape[0]
    dmin = depth_range[:, 0].to(WORK).reshape(b, 1, 1, 1)
    dmax = depth_range[:, 1].to(WORK).reshape(b, 1, 1, 1)
    step = (dmax - dmin) / (ndepths - 1)
    hyp = dmin.unsqueeze(1) + torch.arange(0, ndepths).reshape(1, -1) * step.unsqueeze(1)
    return hyp.reshape(b, ndepths, 1, 1)


def gauss1_fit(prob, depth_hypos):
    """depthhypos.py:169-215: LS fit ln p = b0 x^2 + b1 x + b2 per pixel, s = |-1/b0| -> [B,h,w]."""
    b, d, h, w = prob.shape
    hyp = depth_hypos if depth_hypos.shape[-1] == w else depth_hypos.reshape(b, d, 1, 1).repeat(1, 1, h, w)
    z = torch.log(prob.clamp(min=1e-40)).unsqueeze(-1).permute(0, 2, 3, 1, 4)
    x = torch.stack([hyp ** 2, hyp, torch.ones_like(hyp)], dim=-1).permute(0, 2, 3, 1, 4)
    xt = x.transpose(-1, -2)
    coef = torch.matmul(torch.matmul(torch.inverse(torch.matmul(xt, x)), xt), z).squeeze(-1)
    return torch.abs(-1 / coef[..., 0])


def gauss1_row0(depth_hypos_b_d):
    """Row 0 of (X^T X)^-1 X^T for hypotheses shared by every pixel ([B,D]), bit-identical to the per-pixel
    matrices inside gauss1_fit: same op sequence on a 2x2-pixel replica (operand strides select the BLAS path,
    and with cond ~1e14 the path decides the bits)."""
    b, d = depth_hypos_b_d.shape
    hyp = depth_hypos_b_d.reshape(b, d, 1, 1).repeat(1, 1, 2, 2)
    x = torch.stack([hyp ** 2, hyp, torch.ones_like(hyp)], dim=-1).permute(0, 2, 3, 1, 4)
    xt = x.transpose(-1, -2)
    return torch.matmul(torch.inverse(torch.matmul(xt, x)), xt)[:, 0, 0, 0, :]


def gauss1_fit_explicit(prob, depth_hypos_b_d):
    """gauss1_fit with the per-pixel matmul written out: b0 = sequential fp32 sum_k row[k]*ln p[k] (ATen's naive
    bmm order).  Bit-identical to gauss1_fit (test_oracle_golden.py); this is the form the HIP kernel mirrors."""
    row = gauss1_row0(depth_hypos_b_d)
    z = torch.log(prob.clamp(min=1e-40))
    acc = torch.zeros_like(z[:, 0])
    for k in range(z.shape[1]):
        acc = acc + row[:, k].reshape(-1, 1, 1) * z[:, k]
    return torch.abs(-1 / acc)


def laplace_fit(depth, prob, depth_hypos):
    """depthhypos.py:78-125: b = 1/|sum(x*y)/sum(x*x)|, x = |hyp - depth|, y = ln max(p,1e-40)."""
    b, d, h, w = prob.shape
    hyp = depth_hypos if depth_hypos.shape[-1] == w else depth_hypos.reshape(b, d, 1, 1).repeat(1, 1, h, w)
    y = torch.log(prob.clamp(min=1e-40)).permute(0, 2, 3, 1)
    x = torch.abs(hyp - depth.unsqueeze(1)).permute(0, 2, 3, 1)
    return 1 / torch.abs(torch.sum(x * y, dim=-1) / torch.sum(x * x, dim=-1))


def hypos_by_fit(depth, depth_range, prob, prev_hypos, ndepths, curve, prob_thresh, upsample=True):
    """net/unit/depthhypos.py:27-76.  curve in {None,'gauss1','laplace'} -> [B,D,1,1] | [B,D,2h,2w]."""
    if depth is None:
        return uniform_hypos(depth_range, ndepths)
    b = depth_range.shape[0]
    dmin, dmax = depth_range[:, 0].to(WORK), depth_range[:, 1].to(WORK)
    thr = torch.tensor(prob_thresh)
    with torch.no_grad():
        s = gauss1_fit(prob, prev_hypos) if curve == "gauss1" else laplace_fit(depth, prob, prev_hypos)
        if upsample:
            s = F.interpolate(s.unsqueeze(1), scale_factor=2, mode="bilinear").squeeze(1)
            depth = F.interpolate(depth.unsqueeze(1), scale_factor=2, mode="bilinear").squeeze(1)
        if curve == "gauss1":
            res = torch.sqrt(-1 * s * torch.log(thr))
        else:
            res = torch.abs(s * torch.log(thr))
        res = res.clamp(min=1e-6, max=(dmax.max() - dmin.min()) / 2)
        res = torch.minimum(res, ((dmax - dmin) * 0.2).reshape(b, 1, 1))
        step = res / (ndepths - 1)
        base = depth - 0.5 * res
        hyp = torch.stack([base + step * k if k else base + 0.0 for k in range(ndepths)], dim=1)
        lo, hi = dmin.reshape(b, 1, 1, 1), dmax.reshape(b, 1, 1, 1)
        hyp = lo + (hyp - lo).clamp(min=0)
        hyp = hi + (hyp - hi).clamp(max=0)
        return hyp


# --------------------------------------------------------------------------- a11/a12 surface
def _cbr2(x, p, pre, k, stride, training=False):
    x = F.conv2d(x, p[pre + "conv.weight"], None, stride, (k - 1) // 2)
    return F.relu(_bn(x, p, pre + "bn.", training))


def fpn_4scales(x, p, training=False):
    """net/unit/backbone.py:50-66 -> (f8 [B,64,H/8,W/8], f4 [B,32,H/4,W/4], f2 [B,16,H/2,W/2])."""
    x = _cbr2(_cbr2(x, p, "conv01.0.", 3, 1, training), p, "conv01.1.", 3, 1, training)
    x2 = _cbr2(x, p, "conv12.0.", 5, 2, training)
    x2 = _cbr2(_cbr2(x2, p, "conv12.1.", 3, 1, training), p, "conv12.2.", 3, 1, training)
    x3 = _cbr2(x2, p, "conv23.0.", 5, 2, training)
    x3 = _cbr2(_cbr2(x3, p, "conv23.1.", 3, 1, training), p, "conv23.2.", 3, 1, training)
    x4 = _cbr2(x3, p, "conv34.0.", 5, 2, training)
    x4 = _cbr2(_cbr2(x4, p, "conv34.1.", 3, 1, training), p, "conv34.2.", 3, 1, training)
    y4 = F.conv2d(x4, p["out4.weight"])
    x3 = F.interpolate(x4, scale_factor=2.0, mode="bilinear", align_corners=False) + \
        F.conv2d(x3, p["lat3.weight"], p["lat3.bias"])
    y3 = F.conv2d(x3, p["out3.weight"])
    x2 = F.interpolate(x3, scale_factor=2.0, mode="bilinear", align_corners=False) + \
        F.conv2d(x2, p["lat2.weight"], p["lat2.bias"])
    y2 = F.conv2d(x2, p["out2.weight"])
    return y4, y3, y2


def refine_net2(depth, depth_range, p):
    """net/unit/refine.py:25-46: normalise to [0,1], conv -> 3x Res(x + 0.1*conv(relu(conv(x)))) -> conv,
    skip add, conv -> PixelShuffle(2) -> conv, de-normalise.  [B,h,w] -> [B,2h,2w]."""
    b = depth.shape[0]
    dmin = depth_range[:, 0].to(WORK).reshape(b, 1, 1, 1)
    dmax = depth_range[:, 1].to(WORK).reshape(b, 1, 1, 1)
    x = (depth.unsqueeze(1).detach() - dmin) / (dmax - dmin)
    x0 = F.conv2d(x, p["conv0.weight"], None, 1, 1)
    y = x0
    for i in range(3):
        r = F.conv2d(F.relu(F.conv2d(y, p[f"ress.{i}.conv.0.weight"], None, 1, 1)), p[f"ress.{i}.conv.2.weight"], None, 1, 1)
        y = y + r * 0.1
    y = F.conv2d(y, p["conv1.weight"], None, 1, 1)
    y = F.conv2d(x0 + y, p["conv2.0.weight"], None, 1, 1)
    y = F.conv2d(F.pixel_shuffle(y, 2), p["conv2.2.weight"], None, 1, 1)
    return (dmin + y * (dmax - dmin)).squeeze(1)


# --------------------------------------------------------------------------- a13 loss
def mvs_loss(depths, depth_gt, depth_range):
    """net/loss.py:10-27: sum over the 4 scales of smooth-L1(mean) on gt > depth_min."""
    total = 0.0
    for est, gt in zip(depths, depth_gt.values()):
        m = gt > depth_range[:, 0].reshape(-1, 1, 1)
        total = total + F.smooth_l1_loss(est[m], gt[m], reduction="mean")
    return total


# --------------------------------------------------------------------------- a1 CoreNet.forward
CURVES = (None, "gauss1", "laplace")
THRESH = (0.0, 0.95, 1e-5)


def core_forward(sd, imgs, extrinsics, intrinsics, depth_range, training=False,
                 ndepths=(48, 24, 8), ngroups=(32, 16, 8), keep=False, warp=homo_warping):
    """net/core.py:30-78 composed as config.py:186-218 wires it.
    Returns the output dict; with keep=True also every per-stage intermediate."""
    views = torch.unbind(imgs.to(WORK), 1)
    bb = _sub(sd, "Backbone.")
    feats = [fpn_4scales(v, bb, training) for v in views]
    depth = hyp = prob = None
    depths, trace = [], {}
    for st in range(3):
        fea = [f[st] for f in feats]
        ref_proj, src_projs = scale_cam(intrinsics, extrinsics, st)
        hyp = hypos_by_fit(depth, depth_range, prob, hyp, ndepths[st], CURVES[st], THRESH[st], True)
        cost = vector_aggregate(fea, ref_proj, src_projs, hyp, ngroups[st], _sub(sd, f"Homoaggre.{st}."), training, warp)
        prob = regular(cost, _sub(sd, f"Regular.{st}."), training)
        depth = depth_regression(prob, hyp)
        depths.append(depth)
        if keep:
            trace[f"fea{st}"] = fea
            trace[f"ref_proj{st}"], trace[f"src_projs{st}"] = ref_proj, src_projs
            trace[f"hypos{st}"], trace[f"cost{st}"], trace[f"prob{st}"], trace[f"depth{st}"] = hyp, cost, prob, depth
    depth = refine_net2(depth, depth_range, _sub(sd, "Refine."))
    depths.append(depth)
    if training:
        out = {"depth": depths}
    else:
        conf = confidence_regress(prob)
        conf = F.interpolate(conf.unsqueeze(1), scale_factor=2, mode="nearest").squeeze(1)
        out = {"depth": depth, "confidence": conf}
    return (out, trace) if keep else out
