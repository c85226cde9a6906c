"""torch.optim.Adam (train.py:14) for the parameters of a ddp.FlatBucket as ONE kernel launch per step.

torch's foreach Adam walks the 158 parameter tensors in ~20 multi-tensor launches (0.28 ms of GPU time and ~1 ms of issue time
per cfg3 step).  The bucket already holds the gradients as one flat buffer in parameter order; the two moments live in flat
buffers of the same layout, the parameters stay the module's own tensors (state_dict / checkpoint surface untouched) and are
reached through a per-tensor pointer table built once.  Same update rule as torch (adam.py:_multi_tensor_adam, no amsgrad),
operation by operation; the parameters' version counters are bumped so every derived cache (packed weights) sees the update."""
import ctypes

import torch
from torch.autograd import graph as _graph

from . import lib, MdfHipError
from .ops import _abi, _stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__([{"params": bucket.params, "initial_lr": lr}], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.bucket = bucket
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        self.steps = 0
        self._table = None

    # The moments live in two flat buffers, not in torch's per-parameter `state`: carry them (and the step count that drives
    # the bias correction) through state_dict()/load_state_dict() so optimizer checkpointing cannot silently reset them.
    def state_dict(self):
        sd = super().state_dict()
        sd["flat_adam"] = {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(), "steps": int(self.steps)}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("flat_adam", None)
        if flat is None:
            raise MdfHipError("FlatAdam.load_state_dict: no 'flat_adam' entry (moments + step count); a torch.optim.Adam state cannot "
                              "be loaded into the flat buffers")
        if flat["exp_avg"].numel() != self.exp_avg.numel():
            raise MdfHipError(f"FlatAdam.load_state_dict: {flat['exp_avg'].numel()} moment elements for a bucket of {self.exp_avg.numel()}")
        super().load_state_dict(state_dict)
        with torch.no_grad():
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
        self.steps = int(flat["steps"])

    def _build_table(self):
        params = self.bucket.params
        L = lib()
        nb = int(L.mdf_adam_job_bytes())
        host = (ctypes.c_char * (nb * len(params)))()
        block_job, first, off = [], 0, 0
        for i, p in enumerate(params):
            if not (p.is_contiguous() and p.dtype == torch.float32):
                raise MdfHipError("FlatAdam needs contiguous float32 parameters")
            nblk = int(L.mdf_adam_job_fill(ctypes.addressof(host), i, p.data_ptr(), off, p.numel(), first))
            if nblk < 0:
                raise MdfHipError(f"mdf_adam_job_fill: {L.mdf_last_error().decode()}")
            block_job += [i] * nblk
            first += nblk
            off += p.numel()
        dev = self.bucket.flat.device
        self._table = (torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev), torch.tensor(block_job, dtype=torch.int32).to(dev), first,
                       tuple(p.data_ptr() for p in params))

    def hyper_values(self, step):
        """(lr, 1 - beta1^step, sqrt(1 - beta2^step)) as mdf_adam_step forms them from its arguments (float betas widened to
        double, the results narrowed to float): what a replayed step (graphstep.py) uploads for mdf_adam_step_hyper."""
        import math
        import numpy as np
        grp = self.param_groups[0]
        b1, b2 = (float(np.float32(b)) for b in grp["betas"])
        return [float(np.float32(grp["lr"])), float(np.float32(1.0 - math.pow(b1, float(step)))),
                float(np.float32(math.sqrt(1.0 - math.pow(b2, float(step)))))]

    @torch.no_grad()
    def step(self, closure=None, hyper=None):
        """hyper: optional DEVICE tensor [3] = hyper_values(step) -- the launch then takes its per-step scalars from memory and
        the caller owns the step count (`self.steps` is advanced by graphstep.GraphedTrainStep per replay, not here)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        b = self.bucket
        b.gather()                                   # (a no-op when allreduce_gradients() has just run)
        grp = self.param_groups[0]
        beta1, beta2 = grp["betas"]
        if hyper is None:
            self.steps += 1
        if not b.flat.is_cuda:
            # CPU rehearsal (gloo tests): the same update with torch ops on the flat buffers
            g = b.flat
            if grp["weight_decay"] != 0:
                g = g + grp["weight_decay"] * torch.cat([p.reshape(-1) for p in b.params])
            self.exp_avg.lerp_(g, 1 - beta1)
            self.exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            bc1, bc2 = 1 - beta1 ** self.steps, 1 - beta2 ** self.steps
            upd = (self.exp_avg / (self.exp_avg_sq.sqrt() / bc2 ** 0.5 + grp["eps"])) * (grp["lr"] / bc1)
            off = 0
            for p in b.params:
                p.sub_(upd[off:off + p.numel()].view_as(p))
                off += p.numel()
            return loss
        if self._table is None or self._table[3] != tuple(p.data_ptr() for p in b.params):
            self._build_table()
        jobs, block_job, nblocks, _ = self._table
        if hyper is not None:
            _abi("mdf_adam_step_hyper", (jobs.data_ptr(), block_job.data_ptr(), nblocks, b.flat.data_ptr(), self.exp_avg.data_ptr(),
                                         self.exp_avg_sq.data_ptr(), hyper.data_ptr(), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                         ctypes.c_float(grp["eps"]), ctypes.c_float(grp["weight_decay"]), _stream(b.flat)),
                 tag=f"{len(b.params)} tensors")
            _graph.increment_version(b.params)
            return loss
        _abi("mdf_adam_step", (jobs.data_ptr(), block_job.data_ptr(), nblocks, b.flat.data_ptr(), self.exp_avg.data_ptr(),
                               self.exp_avg_sq.data_ptr(), ctypes.c_float(grp["lr"]), ctypes.c_float(beta1), ctypes.c_float(beta2),
                               ctypes.c_float(grp["eps"]), ctypes.c_float(grp["weight_decay"]), self.steps, _stream(b.flat)),
             tag=f"{len(b.params)} tensors")
        _graph.increment_version(b.params)           # the kernel wrote through raw pointers: tell autograd and the weight caches
        return loss
