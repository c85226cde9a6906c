"""One packed host->device transfer per forward for the control plane of the builtin slots.

The reference's slots each build their few dozen floats where they need them -- scaled cameras (scale.py:4-20), the
relative projections (base.py:98), the uniform hypotheses (depthhypos.py:31-38), the gauss-fit row (depthhypos.py:191-208),
the float depth range, the refinement net's (lo, span) (refine.py:26-27) -- which here meant ten separate host->device copies
and a handful of tiny ATen launches per forward.  `prepare()` computes all of them on the host at the top of CoreNet.forward
(same torch ops on the same values, so the bits are the reference's), packs them into ONE pinned buffer, uploads it with one
non-blocking copy and leaves device views under the identities of the tensors the slots will be called with; the slots look
themselves up (`find`) and fall back to their own computation when called outside a prepared forward or with other tensors.
"""
import threading

import torch

from . import hostmirror, ops

_tls = threading.local()


class Plan:
    def __init__(self):
        self.cams = {}        # stage -> (ref_proj dev view, (src_proj dev views))
        self.projs = {}       # id(ref_proj dev view) -> [n_src,B,12] dev
        self.keep = []        # objects whose id() is a key: kept alive for the duration of the forward
        self.hyp0 = None      # (dev, host) uniform hypotheses [B,D0,1,1]
        self.fit_row = None   # [B,D0] dev
        self.rng = None       # [B,2] float32 dev
        self.lo = self.span = None   # [B] float32 dev (refine)
        self.key = None


def current():
    return getattr(_tls, "plan", None)


class active:
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        self.prev = getattr(_tls, "plan", None)
        _tls.plan = self.plan
        return self.plan

    def __exit__(self, *exc):
        _tls.plan = self.prev


class Staging:
    """A FIXED device buffer (and its layout) for the packed control plane, for a training step that is replayed as a hipGraph
    (mdfnet_hip/graphstep.py): the graph's kernels read the same addresses every replay, the step driver computes the values
    on the host and uploads them with one stream-ordered copy BEFORE each replay (`upload`), and a forward that runs under
    `staged(st)` -- the capture -- takes its device views from this buffer instead of allocating and copying its own."""

    def __init__(self, device):
        self.device, self.dev, self.where = device, None, None

    def upload(self, pieces):
        """pieces: [(name, host tensor)] as `host_pieces` returns them -> one pinned buffer, one non-blocking copy on the current
        stream (torch's pinned allocator recycles the host buffer once the copy has run)."""
        n = sum(t.numel() for _, t in pieces)
        buf = torch.empty(n, dtype=torch.float32, pin_memory=True)
        off, where = 0, {}
        for name, t in pieces:
            k = t.numel()
            buf[off:off + k].copy_(t.reshape(-1))
            where[name] = (off, tuple(t.shape))
            off += k
        if self.dev is None:
            self.dev, self.where = torch.empty(n, dtype=torch.float32, device=self.device), where
        elif where != self.where:
            raise RuntimeError("controlplane.Staging: the control plane's layout changed (other batch size, view count or slot set); "
                               "a captured step is bound to the layout it was captured with")
        self.dev.copy_(buf, non_blocking=True)


class staged:
    def __init__(self, staging):
        self.staging = staging

    def __enter__(self):
        self.prev = getattr(_tls, "staging", None)
        _tls.staging = self.staging
        return self.staging

    def __exit__(self, *exc):
        _tls.staging = self.prev


def builtin_slots(model):
    from net.unit.scale import scale_cam
    from net.unit.depthhypos import HyposByFit
    hyps = list(model.Depth_hypos)
    return model.scale is scale_cam and len(hyps) == 3 and all(isinstance(h, HyposByFit) for h in hyps)


def host_pieces(model, intrinsics, extrinsics, depth_range):
    """The host half of `prepare`: -> ([(name, host tensor)], per-stage host cameras, host hyp0).  Pure host arithmetic on the
    host mirrors of the three tensors (or on the tensors themselves when they are CPU tensors)."""
    from net.unit.scale import host_cameras
    hyps = list(model.Depth_hypos)
    pieces = []
    nstage = len(hyps)
    hosts = [host_cameras(intrinsics, extrinsics, st) for st in range(nstage)]          # [V,B,4,4] each
    rel = [ops.relative_projections(h[0], [h[v] for v in range(1, h.shape[0])]) for h in hosts]        # [n_src,B,12]
    hyp0 = hyps[0].uniform_host(depth_range)                                                     # [B,D0,1,1]
    dr = hostmirror.get(depth_range)
    rng = dr.float().contiguous()
    lo = dr[:, 0].float()
    span = dr[:, 1].float() - lo
    row = ops.gauss1_fit_row(hyp0) if hyps[1].curve_calss == "gauss1" else None
    for st in range(nstage):
        pieces += [(f"cam{st}", hosts[st]), (f"rel{st}", rel[st])]
    pieces += [("hyp0", hyp0), ("rng", rng), ("lo", lo.contiguous()), ("span", span.contiguous())]
    if row is not None:
        pieces.append(("row", row))
    return pieces, hosts, hyp0


def prepare(model, intrinsics, extrinsics, depth_range):
    """-> Plan or None (None: not the builtin slot set, CPU tensors, ...; the slots then work on their own)."""
    if not (intrinsics.is_cuda and extrinsics.is_cuda and depth_range.is_cuda) or not builtin_slots(model):
        return None
    dev = intrinsics.device
    nstage = len(model.Depth_hypos)
    pieces, hosts, hyp0 = host_pieces(model, intrinsics, extrinsics, depth_range)
    row = any(name == "row" for name, _ in pieces)
    staging = getattr(_tls, "staging", None)
    if staging is not None:
        # a step being captured (or rehearsed) for replay: the driver has uploaded this step's values into the fixed buffer
        if staging.dev is None:
            staging.upload(pieces)
        on_dev, where = staging.dev, staging.where
        if {k: v[1] for k, v in where.items()} != {name: tuple(t.shape) for name, t in pieces}:
            raise RuntimeError("controlplane.prepare: the staged control plane has another layout than this forward's")
    else:
        n = sum(t.numel() for _, t in pieces)
        buf = torch.empty(n, dtype=torch.float32, pin_memory=True)      # (torch's pinned allocator recycles it once the copy has run)
        off, where = 0, {}
        for name, t in pieces:
            k = t.numel()
            buf[off:off + k].copy_(t.reshape(-1))
            where[name] = (off, tuple(t.shape))
            off += k
        on_dev = buf.to(dev, non_blocking=True)

    def view(name):
        o, shape = where[name]
        return on_dev[o:o + (torch.Size(shape).numel())].view(shape)
    plan = Plan()
    plan.key = (id(intrinsics), id(extrinsics), id(depth_range))
    plan.keep += [intrinsics, extrinsics, depth_range, on_dev]
    for st in range(nstage):
        cam = view(f"cam{st}")
        outs = [hostmirror.put(cam[v], hosts[st][v]) for v in range(cam.shape[0])]
        plan.cams[st] = (outs[0], tuple(outs[1:]))
        plan.projs[id(outs[0])] = view(f"rel{st}")
        plan.keep.append(outs)
    plan.hyp0 = hostmirror.put(view("hyp0"), hyp0)
    plan.rng, plan.lo, plan.span = view("rng"), view("lo"), view("span")
    plan.fit_row = view("row") if row else None
    return plan


def cams(intrinsics, extrinsics, stage):
    p = current()
    if p is not None and p.key[:2] == (id(intrinsics), id(extrinsics)):
        return p.cams.get(stage)
    return None


def projections(ref_proj):
    p = current()
    return None if p is None else p.projs.get(id(ref_proj))


def for_range(depth_range):
    p = current()
    return p if (p is not None and p.key[2] == id(depth_range)) else None
