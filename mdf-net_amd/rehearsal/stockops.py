"""Stock PyTorch-op restatement of the hot-path slots in TRAINING mode -- NOT the product's GPU training path.

On an MI355X `model.train()` runs the hand-written training kernels (mdfnet_hip/train_ops.py).  This module is what a
slot runs for CPU tensors in training mode, and exists for the gloo rehearsals of the data-parallel driver on machines
without a GPU (tests/test_train_cpu.py -- where it is itself pinned to the reference's training golden, loss + gradients at
rtol 1e-4 --, tests/test_train_data_cpu.py) and, with `MDF_TRAIN_STOCK=1`, as bench.py's stated PyTorch-ROCm baseline
(scripts/bench_train.py).  It is NOT a checker of the GPU path: tests/test_train_gpu.py compares the kernels with oracle/
(oracle/train_check.py, fp32 and float64).  It is an explicit mode, never a silent fallback: inference has no stock-op route
(net/core.py raises without a GPU).
Arithmetic follows the reference: net/unit/base.py:85-126, homoaggregate.py:25-69, depthhypos.py:27-215, regress.py:5-25."""
import torch
import torch.nn.functional as F


def homo_warping(src_fea, src_proj, ref_proj, depth_hypos):
    """Plane-sweep warp; gradient flows to src_fea only (the sampling grid is built under no_grad, base.py:97)."""
    b, c, h, w = src_fea.shape
    d = depth_hypos.shape[1]
    with torch.no_grad():
        m = torch.matmul(src_proj, torch.inverse(ref_proj))
        ys, xs = torch.meshgrid(torch.arange(h, dtype=src_fea.dtype, device=src_fea.device),
                                torch.arange(w, dtype=src_fea.dtype, device=src_fea.device), indexing="ij")
        pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, dtype=src_fea.dtype, device=src_fea.device)))
        ray = torch.matmul(m[:, :3, :3], pix.unsqueeze(0).expand(b, 3, h * w))
        pts = ray.unsqueeze(2) * depth_hypos.reshape(b, 1, d, -1) + m[:, :3, 3].reshape(b, 3, 1, 1)
        uv = pts[:, :2] / pts[:, 2:3]
        gx = uv[:, 0] / ((w - 1) / 2) - 1
        gy = uv[:, 1] / ((h - 1) / 2) - 1
        grid = torch.stack((gx, gy), dim=3).reshape(b, d * h, w, 2)
    out = F.grid_sample(src_fea, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return out.reshape(b, c, d, h, w)


def vector_aggregate(weight_head, ngroups, features, ref_proj, src_projs, depth_hypos):
    """VectorAggregate.forward with `weight_head` = the module's depth_weight Sequential (train-mode BN inside)."""
    ref = features[0]
    b, c, h, w = ref.shape
    d = depth_hypos.shape[1]
    ref_unit = F.softmax(ref.reshape(b, ngroups, c // ngroups, 1, h, w), dim=2)
    num = den = 0.0
    for fea, proj in zip(features[1:], src_projs):
        vol = homo_warping(fea, proj, ref_proj, depth_hypos).reshape(b, ngroups, c // ngroups, d, h, w)
        sim = (F.softmax(vol, dim=2) * ref_unit).sum(dim=2)
        wgt = weight_head(sim)
        den = den + wgt
        num = num + wgt * sim
    return num / den


def variance_aggregate(features, ref_proj, src_projs, depth_hypos):
    ref = features[0].unsqueeze(2)
    s1, s2 = ref, ref ** 2
    for fea, proj in zip(features[1:], src_projs):
        vol = F.softmax(homo_warping(fea, proj, ref_proj, depth_hypos), dim=1)
        s1 = s1 + vol
        s2 = s2 + vol ** 2
    n = len(features)
    return s2 / n - (s1 / n) ** 2


def depth_regression(prob, depth_hypos):
    return torch.sum(prob * depth_hypos, 1)


def _fit_gauss1(prob, hypos):
    b, d, h, w = prob.shape
    hyp = hypos if hypos.shape[-1] == w else hypos.reshape(b, d, 1, 1).repeat(1, 1, h, w)
    z = torch.log(prob.clamp(min=1e-40)).unsqueeze(-1).permute(0, 2, 3, 1, 4)
    x = torch.stack([hyp ** 2, hyp, torch.ones_like(hyp)], dim=-1).permute(0, 2, 3, 1, 4)
    xt = x.transpose(-1, -2)
    coef = torch.matmul(torch.matmul(torch.inverse(torch.matmul(xt, x)), xt), z).squeeze(-1)
    return torch.abs(-1 / coef[..., 0])


def _fit_laplace(depth, prob, hypos):
    b, d, h, w = prob.shape
    hyp = hypos if hypos.shape[-1] == w else hypos.reshape(b, d, 1, 1).repeat(1, 1, h, w)
    y = torch.log(prob.clamp(min=1e-40)).permute(0, 2, 3, 1)
    x = torch.abs(hyp - depth.unsqueeze(1)).permute(0, 2, 3, 1)
    return 1 / torch.abs(torch.sum(x * y, dim=-1) / torch.sum(x * x, dim=-1))


def hypos_by_fit(curve, prob_thresh, ndepths, depth, depth_range, prob, hypos, upsample):
    """HyposByFit.forward for depth is not None (always under no_grad in the reference, depthhypos.py:40)."""
    b = depth_range.shape[0]
    lo, hi = depth_range[:, 0].float(), depth_range[:, 1].float()
    with torch.no_grad():
        s = _fit_gauss1(prob, hypos) if curve == "gauss1" else _fit_laplace(depth, prob, hypos)
        if upsample:
            s = F.interpolate(s.unsqueeze(1), scale_factor=2, mode="bilinear").squeeze(1)
            depth = F.interpolate(depth.unsqueeze(1), scale_factor=2, mode="bilinear").squeeze(1)
        thr = prob_thresh.to(s.device)
        res = torch.sqrt(-1 * s * torch.log(thr)) if curve == "gauss1" else torch.abs(s * torch.log(thr))
        res = res.clamp(min=1e-6, max=(hi.max() - lo.min()) / 2)
        res = torch.minimum(res, ((hi - lo) * 0.2).reshape(b, 1, 1))
        step = res / (ndepths - 1)
        start = depth - 0.5 * res
        hyp = torch.stack([start + step * k for k in range(ndepths)], dim=1)
        lo4, hi4 = lo.reshape(b, 1, 1, 1), hi.reshape(b, 1, 1, 1)
        hyp = lo4 + (hyp - lo4).clamp(min=0)
        return hi4 + (hyp - hi4).clamp(max=0)
