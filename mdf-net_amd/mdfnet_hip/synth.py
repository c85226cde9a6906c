"""Deterministic synthetic inputs and weights (no dataset / checkpoint exists offline).

Recipe follows SURVEY.md section 8(d): DTU-like intrinsics scaled to the image size,
x-baseline extrinsics in +-40 mm steps with an optional small seeded rotation,
depth range [425, 935] mm, images ~ U[0,1).

Everything here is pure numpy (MT19937 is stable across numpy releases) so the
same tensors are produced in the build container (golden generation from the
real reference) and on the GPU box (parity tests, bench).
"""
import zlib

import numpy as np
import torch

DTU_K = np.array([[2892.33, 0.0, 823.2], [0.0, 2883.18, 619.07], [0.0, 0.0, 1.0]], dtype=np.float64)
DTU_RANGE = (425.0, 935.0)
NDEPTHS = (48, 24, 8)
NGROUPS = (32, 16, 8)


def _rot(rng, max_deg):
    ang = np.deg2rad(rng.uniform(-max_deg, max_deg, size=3))
    cx, cy, cz = np.cos(ang)
    sx, sy, sz = np.sin(ang)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def make_cameras(width, height, nviews, batch=1, rot_deg=0.0, seed=1,
                 depth_range=DTU_RANGE, base_k=DTU_K, base_size=(1600, 1200), baseline=40.0):
    """-> intrinsics [B,V,3,3] f32, extrinsics [B,V,4,4] f32, depth_range [B,2] f64."""
    rng = np.random.RandomState(seed)
    k = base_k.copy()
    k[0, :] *= width / base_size[0]
    k[1, :] *= height / base_size[1]
    intr = np.zeros((batch, nviews, 3, 3), np.float32)
    extr = np.zeros((batch, nviews, 4, 4), np.float32)
    for b in range(batch):
        for v in range(nviews):
            e = np.eye(4)
            if v > 0:
                step = (v + 1) // 2
                sign = 1.0 if v % 2 == 1 else -1.0
                e[0, 3] = sign * baseline * step + (2.0 * b)
                e[1, 3] = 3.0 * (v - 1) * (1 if b % 2 == 0 else -1)
                if rot_deg > 0:
                    e[:3, :3] = _rot(rng, rot_deg)
            intr[b, v] = k
            extr[b, v] = e
    dr = np.tile(np.array(depth_range, np.float64)[None], (batch, 1))
    return torch.from_numpy(intr), torch.from_numpy(extr), torch.from_numpy(dr)


def make_images(width, height, nviews, batch=1, seed=0, smooth=True):
    """imgs [B,V,3,H,W] f32 in [0,1). `smooth` adds low-frequency structure shared by the
    views (shifted) so the cost volume has something to match."""
    rng = np.random.RandomState(seed)
    if not smooth:
        return torch.from_numpy(rng.rand(batch, nviews, 3, height, width).astype(np.float32))
    big = rng.rand(batch, 3, height // 4 + 16, width // 4 + 64).astype(np.float32)
    big = np.repeat(np.repeat(big, 4, axis=2), 4, axis=3)
    out = np.empty((batch, nviews, 3, height, width), np.float32)
    for v in range(nviews):
        sh = 8 * (v % 32)
        out[:, v] = 0.7 * big[:, :, 16:16 + height, sh:sh + width]
    out += 0.3 * rng.rand(batch, nviews, 3, height, width).astype(np.float32)
    return torch.from_numpy(np.clip(out, 0.0, 0.999999).astype(np.float32))


def make_scene(width, height, nviews, batch=1, rot_deg=0.0, seed=0):
    imgs = make_images(width, height, nviews, batch, seed)
    intr, extr, dr = make_cameras(width, height, nviews, batch, rot_deg, seed + 1)
    return imgs, extr, intr, dr


def _key_rng(seed, key):
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)


def seeded_state_dict(reference_sd, seed=1, prob_gain=6.0):
    """Deterministic weights keyed by state_dict name/shape only (independent of module
    construction order and of torch's RNG).

    conv weights ~ N(0, 2/fan_in); BN gamma in [0.8,1.2], beta in [-0.1,0.1], running_mean
    in [-0.2,0.2], running_var in [0.5,1.5] (so BN folding is exercised); the `prob` convs
    are scaled by `prob_gain` so probability volumes are peaked.
    """
    out = {}
    for key, ref in reference_sd.items():
        shape = tuple(ref.shape)
        rng = _key_rng(seed, key)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.tensor(3, dtype=torch.int64)
            continue
        if key.endswith("running_mean"):
            a = rng.uniform(-0.2, 0.2, size=shape)
        elif key.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, size=shape)
        elif len(shape) == 1 and key.endswith(".bias"):  # BN beta, conv bias
            a = rng.uniform(-0.1, 0.1, size=shape)
        elif len(shape) == 1:  # BN gamma
            a = rng.uniform(0.8, 1.2, size=shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.normal(0.0, np.sqrt(2.0 / max(fan_in, 1)), size=shape)
            if ".prob." in key:
                a = a * prob_gain
        out[key] = torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(shape)
    return out
