"""Data parallelism the MI355X way: one process per GPU, ONE flat fp32 gradient bucket, ONE all-reduce per step.

The reference trains with torch.nn.DataParallel (train.py:24-26): single process, one thread per GPU, parameters
re-broadcast every forward, gradients reduce-added to GPU 0, BatchNorm statistics per replica and only replica 0's
running stats surviving.  Here each rank owns a full replica; the 1,206,380 gradients (158 tensors, 4.83 MB) are
views into one contiguous buffer so a step costs exactly one RCCL all-reduce over xGMI (backend "nccl" on ROCm; "gloo"
on CPU for the tests), and rank 0's BatchNorm buffers are broadcast to mirror DataParallel's semantics."""
import os

import torch
import torch.distributed as dist


class FlatBucket:
    def __init__(self, module):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:                      # every .grad becomes a view into the bucket
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._pad = self._recv = None
        self._view_list = list(self._views())

    def zero_grad(self):
        """Gradients are GATHERED into the bucket after the backward pass (one batched copy) instead of being accumulated into
        pre-assigned views: with .grad = None autograd simply hands over each gradient tensor -- no zero fill and no
        158 tiny `grad += new` launches per step (0.7 ms of GPU time and ~1.5 ms of issue time at cfg3)."""
        for p in self.params:
            p.grad = None

    def gather(self):
        """Copy the parameters' gradients into the flat buffer (parameters without a gradient contribute zeros) and make
        every .grad a view into it, which is what the optimizer then consumes."""
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads) or any(g.data_ptr() != v.data_ptr() for g, v in zip(grads, self._view_list)):
            pieces = [(torch.zeros_like(p).reshape(-1) if g is None else g.detach().reshape(-1).to(torch.float32)) for p, g in zip(self.params, grads)]
            torch.cat(pieces, out=self.flat)
            for p, v in zip(self.params, self._view_list):
                p.grad = v

    def _views(self):
        off = 0
        for p in self.params:
            yield self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def broadcast_parameters(self, src=0):
        """All replicas start from rank 0's weights (DataParallel replicates from device 0)."""
        if self.world > 1:
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1) for p in self.module.parameters()])
                dist.broadcast(flat, src)
                off = 0
                for p in self.module.parameters():
                    p.copy_(flat[off:off + p.numel()].view_as(p))
                    off += p.numel()
            self.broadcast_buffers(src)

    def broadcast_buffers(self, src=0):
        """BatchNorm running stats: per-replica in training, rank 0's persist (as with DataParallel)."""
        if self.world > 1:
            for b in self.module.buffers():
                dist.broadcast(b, src)

    def allreduce_gradients(self, mode=None):
        """Mean over ranks of the flat gradient buffer.
        mode "allreduce" (default): ONE collective per step (RCCL picks ring/tree; gloo in the CPU tests).
        mode "direct" (MDF_GRAD_EXCHANGE=direct): the two-step exchange SURVEY section 5 proposes for a fully connected xGMI
        node -- every rank sends shard j of its 4.83-MB buffer straight to rank j (all-to-all: all 7 links busy at once, 0.6 MB
        per link), sums the shards it owns, and the owners' sums are all-gathered: 2 latency steps instead of the 14 of a ring.
        Same result up to summation order (asserted in tests/test_train_cpu.py); unmeasured on hardware, hence not the default."""
        self.gather()
        if self.world <= 1:
            return
        mode = mode or os.environ.get("MDF_GRAD_EXCHANGE", "allreduce")
        if mode == "direct":
            n, w = self.flat.numel(), self.world
            shard = (n + w - 1) // w
            if self._pad is None or self._pad.numel() != shard * w:
                self._pad = torch.zeros(shard * w, dtype=torch.float32, device=self.flat.device)
                self._recv = torch.empty_like(self._pad)
            self._pad[:n].copy_(self.flat)
            dist.all_to_all_single(self._recv, self._pad)                 # row r of _recv: rank r's copy of MY shard
            mine = self._recv.view(w, shard).sum(0).div_(w)
            dist.all_gather_into_tensor(self._pad, mine)
            self.flat.copy_(self._pad[:n])
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(self.world)
