"""Host mirrors of tiny control-plane tensors (camera matrices, depth range, shared hypotheses).

The reference's arithmetic on these (4x4 / 3x3 `torch.inverse`, small matmuls) is only reproducible bit
for bit on the CPU (LAPACK), so the product keeps a CPU copy next to the device tensor the slot API
passes around.  A mirror is registered when the host value is known (no device->host sync needed later);
`get` falls back to one explicit `.cpu()` when a caller hands in a tensor we have not seen.
"""
import threading
import weakref

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


def get(t):
    """CPU copy of `t` (cached if registered and unmodified since)."""
    if not t.is_cuda:
        return t.detach()
    ent = _table().get(id(t))
    if ent is not None and ent[0]() is t and ent[2] == t._version:
        return ent[1]
    return t.detach().cpu()
