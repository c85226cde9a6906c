"""CPU ORACLE for the depth-map consistency filter / fusion.  TEST INFRASTRUCTURE ONLY.

Restatement of tools/filter/dynamic_filter_gpu.py: `reproject_with_depth` (:194-238), `check_geometric_consistency`
(:166-191) and the per-reference-view fusion of `filter` (:63-103, :130-144), with `bilinear_sampler`
(tools/filter/data_io.py:117-131: pixel coordinates -> grid_sample(align_corners=True), zero padding).
Pinned by tests/golden/filter.npz, produced by the reference's own functions (oracle/gen_golden.py --only-filter)."""
import torch
import torch.nn.functional as F


def bilinear_sampler(img, coords):
    h, w = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xg / (w - 1) - 1, 2 * yg / (h - 1) - 1], dim=-1)
    return F.grid_sample(img, grid, align_corners=True)


def reproject_with_depth(depth_ref, k_ref, e_ref, depth_src, k_src, e_src):
    """dynamic_filter_gpu.py:194-238 -> (depth_reprojected, x_reprojected, y_reprojected, x_src, y_src), each [1,h,w]."""
    h, w = depth_ref.shape
    ys, xs = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    xs, ys = xs.reshape(1, -1), ys.reshape(1, -1)
    ones = torch.ones_like(xs)
    pts = torch.matmul(torch.inverse(k_ref), torch.stack((xs, ys, ones), dim=1) * depth_ref.reshape(1, 1, -1))
    in_src = torch.matmul(torch.matmul(e_src, torch.inverse(e_ref)), torch.cat((pts, ones.unsqueeze(1)), dim=1))[:, :3]
    proj = torch.matmul(k_src, in_src)
    xy_src = proj[:, :2] / proj[:, 2:3]
    x_src = xy_src[:, 0].reshape(1, h, w).float()
    y_src = xy_src[:, 1].reshape(1, h, w).float()
    sampled = bilinear_sampler(depth_src.view(1, 1, h, w), torch.stack((x_src, y_src), dim=-1).view(1, h, w, 2))
    back = torch.matmul(torch.inverse(k_src), torch.cat((xy_src, ones.unsqueeze(1)), dim=1) * sampled.reshape(1, 1, -1))
    in_ref = torch.matmul(torch.matmul(e_ref, torch.inverse(e_src)), torch.cat((back, ones.unsqueeze(1)), dim=1))[:, :3]
    depth_rep = in_ref[:, 2].reshape(1, h, w).float()
    pr = torch.matmul(k_ref, in_ref)
    xy = pr[:, :2] / pr[:, 2:3]
    return depth_rep, xy[:, 0].reshape(1, h, w).float(), xy[:, 1].reshape(1, h, w).float(), x_src, y_src


def _fma(a, b, c):
    return (a.double() * b.double() + c.double()).float()


def _mm(m, rows):
    """[r,k] x k rows of [N] with the rounding order of torch's CPU matmul on the build host (verified bit-exact on the
    goldens): acc = m0*x0, then acc = fma(m_j, x_j, acc)."""
    out = []
    for i in range(m.shape[0]):
        acc = m[i, 0] * rows[0]
        for j in range(1, m.shape[1]):
            acc = _fma(m[i, j].expand_as(rows[j]), rows[j], acc)
        out.append(acc)
    return out


def reproject_explicit(depth_ref, k_ref, e_ref, depth_src, k_src, e_src):
    """reproject_with_depth with every elementwise step written out (host-independent: the ATen form above depends on the
    host BLAS's accumulation order).  Only the 3x3/4x4 inverses and products come from torch (same calls as the product's
    host side).  Bit-identical to the reference's output on the build container (tests/test_filter_oracle_cpu.py)."""
    h, w = depth_ref.shape
    ys, xs = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    x, y = xs.reshape(-1).float(), ys.reshape(-1).float()
    d = depth_ref.reshape(-1)
    one = torch.ones_like(x)
    kr_inv, t_rs = torch.inverse(k_ref), torch.matmul(e_src, torch.inverse(e_ref))
    ks_inv, t_sr = torch.inverse(k_src), torch.matmul(e_ref, torch.inverse(e_src))
    c = _mm(kr_inv, [x * d, y * d, one * d])
    s = _mm(t_rs, c + [one])[:3]
    q = _mm(k_src, s)
    x_src, y_src = q[0] / q[2], q[1] / q[2]
    gx = 2 * x_src / (w - 1) - 1
    gy = 2 * y_src / (h - 1) - 1
    ix = (gx + 1) * torch.tensor((w - 1) / 2, dtype=torch.float32)
    iy = (gy + 1) * torch.tensor((h - 1) / 2, dtype=torch.float32)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    fw = ix - x0
    fe = 1 - fw
    fn = iy - y0
    fs = 1 - fn
    src = depth_src.reshape(-1)

    def tap(xf, yf):
        m = (xf >= 0) & (xf <= w - 1) & (yf >= 0) & (yf <= h - 1)
        xi = torch.nan_to_num(xf, nan=0.0, posinf=0.0, neginf=0.0).clamp(0, w - 1).long()
        yi = torch.nan_to_num(yf, nan=0.0, posinf=0.0, neginf=0.0).clamp(0, h - 1).long()
        return src[yi * w + xi] * m

    samp = tap(x0, y0) * (fs * fe)
    samp = _fma(tap(x0 + 1, y0), fs * fw, samp)
    samp = _fma(tap(x0, y0 + 1), fn * fe, samp)
    samp = _fma(tap(x0 + 1, y0 + 1), fn * fw, samp)
    b = _mm(ks_inv, [x_src * samp, y_src * samp, one * samp])
    r = _mm(t_sr, b + [one])[:3]
    u = _mm(k_ref, r)
    shp = (1, h, w)
    return r[2].reshape(shp), (u[0] / u[2]).reshape(shp), (u[1] / u[2]).reshape(shp), x_src.reshape(shp), y_src.reshape(shp)


def check_geometric_consistency(depth_ref, k_ref, e_ref, depth_src, k_src, e_src, thre1=4, thre2=1300.0, explicit=False):
    """dynamic_filter_gpu.py:166-191 -> (9 masks for i = 2..10, last mask, depth_reprojected zeroed outside it)."""
    h, w = depth_ref.shape
    ys, xs = torch.meshgrid(torch.arange(0, h), torch.arange(0, w), indexing="ij")
    fn_ = reproject_explicit if explicit else reproject_with_depth
    depth_rep, xr, yr, _, _ = fn_(depth_ref, k_ref, e_ref, depth_src, k_src, e_src)
    dist = torch.sqrt((xr - xs.unsqueeze(0)) ** 2 + (yr - ys.unsqueeze(0)) ** 2)
    rel = torch.abs(depth_rep - depth_ref) / depth_ref
    masks = [torch.logical_and(dist < i / thre1, rel < i / thre2) for i in range(2, 11)]
    depth_rep = depth_rep.clone()
    depth_rep[~masks[-1]] = 0
    return masks, masks[-1], depth_rep


def fuse_view(depth_ref, conf, k_ref, e_ref, src_depths, src_ks, src_es, photo_threshold=0.8, nconditions=5, thre1=4,
              thre2=1300.0, explicit=False):
    """filter():63-103 for one reference view -> dict(geo_mask, photo_mask, final_mask, depth_avg, counts[9], nvalid)."""
    counts = [torch.zeros(1, *depth_ref.shape) for _ in range(9)]
    nvalid = torch.zeros(1, *depth_ref.shape)
    acc = torch.zeros(1, *depth_ref.shape)
    for d, k, e in zip(src_depths, src_ks, src_es):
        masks, last, rep = check_geometric_consistency(depth_ref, k_ref, e_ref, d, k, e, thre1, thre2, explicit)
        for i in range(9):
            counts[i] = counts[i] + masks[i].float()
        nvalid = nvalid + last
        acc = acc + rep
    geo = sum((counts[i - 2] >= i).to(torch.int64) for i in range(2, 11))
    depth_avg = (acc + depth_ref) / (nvalid + 1)
    geo_mask = geo >= nconditions
    photo_mask = conf > photo_threshold
    return {"geo_mask": geo_mask[0], "photo_mask": photo_mask, "final_mask": torch.logical_and(photo_mask, geo_mask)[0],
            "depth_avg": depth_avg[0], "counts": torch.stack(counts, 0)[:, 0], "nvalid": nvalid[0]}


def backproject(depth_avg, mask, k_ref, e_ref):
    """filter():130-143 in numpy semantics (int64 pixel grid * float32 depth -> float64): world points of valid pixels."""
    import numpy as np
    h, w = depth_avg.shape
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    m = mask.numpy()
    x, y, d = x[m], y[m], depth_avg.numpy()[m]
    cam = np.matmul(np.linalg.inv(k_ref.numpy()), np.vstack((x, y, np.ones_like(x))) * d)
    world = np.matmul(np.linalg.inv(e_ref.numpy()), np.vstack((cam, np.ones_like(x))))[:3]
    return world.transpose((1, 0))
