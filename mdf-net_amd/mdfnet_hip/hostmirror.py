"""Host mirrors of tiny control-plane tensors (camera matrices, depth range, shared hypotheses).

The reference's arithmetic on these (4x4 / 3x3 `torch.inverse`, small matmuls) is only reproducible bit
for bit on the CPU (LAPACK), so the product keeps a CPU copy next to the device tensor the slot API
passes around.  A mirror is registered when the host value is known (no device->host sync needed later);
`get` falls back to one explicit `.cpu()` when a caller hands in a tensor we have not seen.
"""
import threading
import weakref

import torch

_tls = threading.local()


def _table():
    t = getattr(_tls, "t", None)
    if t is None:
        t = _tls.t = {}
    return t


def put(dev_tensor, host_tensor):
    tab = _table()
    key = id(dev_tensor)
    tab[key] = (weakref.ref(dev_tensor, lambda _r, k=key, tb=tab: tb.pop(k, None)), host_tensor, dev_tensor._version)
    return dev_tensor


def fetch(tensors):
    """Device -> host for a few tiny tensors with ONE stream synchronise: pinned staging buffers + non-blocking
    copies.  (A plain `.cpu()` queued behind pending kernels stalls ~10 ms on this ROCm stack, measured with
    scripts/diag_copy_latency.py; a stream synchronise does not.)"""
    outs, dev = [], None
    for t in tensors:
        if t.is_cuda:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            buf.copy_(t.detach(), non_blocking=True)
            outs.append(buf)
            dev = t.device
        else:
            outs.append(t.detach())
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()
    return outs


def _lookup(t):
    ent = _table().get(id(t))
    if ent is not None and ent[0]() is t and ent[2] == t._version:
        return ent[1]
    return None


def ensure(tensors):
    """Register host mirrors for device tensors that do not have a current one (one sync for all of them)."""
    missing = [t for t in tensors if t.is_cuda and _lookup(t) is None]
    if missing:
        for t, h in zip(missing, fetch(missing)):
            put(t, h)


def get(t):
    """CPU copy of `t` (cached if registered and unmodified since)."""
    if not t.is_cuda:
        return t.detach()
    h = _lookup(t)
    return h if h is not None else fetch([t])[0]
