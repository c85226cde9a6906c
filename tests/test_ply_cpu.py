"""CPU: the .ply writer of the filter driver (binary little-endian, float xyz + uchar rgb) round-trips."""
import numpy as np


def test_ply_round_trip(tmp_path):
    from tools.filter import dynamic_filter_gpu as filt
    rng = np.random.RandomState(0)
    xyz = rng.randn(1000, 3).astype(np.float32) * 100
    rgb = rng.randint(0, 256, (1000, 3)).astype(np.uint8)
    p = str(tmp_path / "a.ply")
    filt.write_ply(p, xyz, rgb)
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 1000\nproperty float x\n")
    a, b = filt.read_ply(p)
    assert np.array_equal(a, xyz) and np.array_equal(b, rgb)
    assert os_size(p) == len(head.split(b"end_header\n")[0]) + len(b"end_header\n") + 1000 * 15


def os_size(p):
    import os
    return os.path.getsize(p)
