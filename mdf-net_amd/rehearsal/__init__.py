"""Rehearsal backend: the slots' TRAINING mode on stock PyTorch ops -- not part of the product's dispatch.

The product runs every slot on hand-written MI355X kernels and refuses CPU tensors in eval AND in training mode
(mdfnet_hip/layers.py: use_hip / hip_train raise).  What lives here is selected explicitly, by the callers that need it:
  * the gloo rehearsals of the data-parallel driver on machines without a GPU (tests/test_train_cpu.py, test_train_data_cpu.py,
    test_shard_gloo_cpu.py, `train.py` on a CPU device),
  * bench.py's stated PyTorch-ROCm autograd baseline on the GPU (`enable(on_gpu=True)`, the former MDF_TRAIN_STOCK switch).
Usage:  `import rehearsal; rehearsal.enable()`  (or `with rehearsal.mode(): ...`).  It is never a checker of the GPU path: the GPU
tests compare the kernels with oracle/ and the reference's goldens."""
import contextlib

from mdfnet_hip import layers

from . import stockops  # noqa: F401


def enable(on_gpu=False):
    """Training-mode slots take the stock-op route for CPU tensors (and, with on_gpu, for GPU tensors too: the autograd baseline)."""
    layers._REHEARSAL = stockops
    layers._TRAIN_STOCK = bool(on_gpu)


def disable():
    layers._REHEARSAL = None
    layers._TRAIN_STOCK = False


def enabled():
    return layers._REHEARSAL is not None


@contextlib.contextmanager
def mode(on_gpu=False):
    prev = (layers._REHEARSAL, layers._TRAIN_STOCK)
    enable(on_gpu)
    try:
        yield
    finally:
        layers._REHEARSAL, layers._TRAIN_STOCK = prev
