"""Host-side file I/O of the eval data plane (counterpart of the reference's tools/data_io.py:6-130).
PFM: 'Pf\\n<w> <h>\\n-1.000000\\n' + bottom-up little-endian float32 rows (data_io.py:44-71)."""
import re
import sys

import numpy as np
import torch


def read_pfm(filename):
    with open(filename, "rb") as f:
        kind = f.readline().decode("utf-8").rstrip()
        if kind not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if kind == "PF" else (height, width)
    return np.flipud(data.reshape(shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    image = np.flipud(np.asarray(image))
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        header = "PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        header = "Pf\n"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    if image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(header.encode("utf-8"))
        f.write("{} {}\n".format(image.shape[1], image.shape[0]).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        image.tofile(f)


def write_depth_img(filename, depth):
    from PIL import Image
    Image.fromarray((depth - 500) / 2).convert("L").save(filename)
    return 1


def read_pairfile(pair_path):
    """-> (num_viewpoint, [[ref_view, [src views sorted by score]], ...])   (data_io.py:79-89)"""
    pairs = []
    with open(pair_path) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            srcs = [int(x) for x in f.readline().rstrip().split()[1::2]]
            pairs.append([ref, srcs])
    return n, pairs


def read_cam_file(filename, with_range=False):
    """extrinsic 4x4 on lines 1-4, intrinsic 3x3 on lines 7-9, optional depth range on line 11 (data_io.py:92-101)."""
    with open(filename) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extrinsic = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsic = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    if with_range:
        return intrinsic, extrinsic, np.array(lines[11].split(), dtype=np.float32)
    return intrinsic, extrinsic


def resize_nearest(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST) without OpenCV (absent here; used by load/dtutrain.py:55-57
    and load/blendedtrain.py:48-50 for the 1/8, 1/4, 1/2 ground-truth maps).  OpenCV's nearest neighbour is NOT
    centre-aligned: source index = min(floor(dst_index * (src/dst)), src - 1) with the ratio in double precision
    (resize.cpp, resizeNN).  For the power-of-two factors the loaders use on divisible sizes this is plain striding."""
    w, h = int(size[0]), int(size[1])
    sh, sw = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h, dtype=np.float64) * (sh / h)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=np.float64) * (sw / w)).astype(np.int64), sw - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def read_img(filename):
    from PIL import Image
    return np.array(Image.open(filename), dtype=np.float32) / 255.0


def tocuda(data_batch, device, parallel=False):
    out = {}
    for k, v in data_batch.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, dict):
            out[k] = {k2: v2.to(device) for k2, v2 in v.items()}
    return out
