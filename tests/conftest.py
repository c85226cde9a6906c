import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mdf-net_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def rehearsal_backend():
    """CPU tests of the training drivers select the stock-op rehearsal backend explicitly (mdf-net_amd/rehearsal); the product's own
    dispatch refuses CPU tensors."""
    import rehearsal
    with rehearsal.mode():
        yield


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def seeded_sd():
    """Deterministic weights (same recipe the goldens were generated with)."""
    import numpy as np
    import torch
    from mdfnet_hip import synth

    meta = np.load(os.path.join(GOLDEN, "state_dict_meta.npz"))
    shapes = {}
    for k, s, dt in zip(meta["keys"], meta["shapes"], meta["dtypes"]):
        shape = tuple(int(x) for x in s.strip("[]").split(",") if x.strip())
        shapes[str(k)] = torch.empty(shape, dtype=torch.int64 if "int64" in str(dt) else torch.float32)
    return synth.seeded_state_dict(shapes, seed=1)
