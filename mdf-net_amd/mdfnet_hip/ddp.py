"""Data parallelism the MI355X way: one process per GPU, ONE flat fp32 gradient bucket, ONE all-reduce per step.

The reference trains with torch.nn.DataParallel (train.py:24-26): single process, one thread per GPU, parameters
re-broadcast every forward, gradients reduce-added to GPU 0, BatchNorm statistics per replica and only replica 0's
running stats surviving.  Here each rank owns a full replica; the 1,206,380 gradients (158 tensors, 4.83 MB) are
views into one contiguous buffer so a step costs exactly one RCCL all-reduce over xGMI (backend "nccl" on ROCm; "gloo"
on CPU for the tests), and rank 0's BatchNorm buffers are broadcast to mirror DataParallel's semantics."""
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

    def zero_grad(self):
        self.flat.zero_()
        for p, g in zip(self.params, self._views()):   # optimizers may have replaced .grad (set_to_none)
            p.grad = g

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

    def allreduce_gradients(self):
        """Mean over ranks of the flat gradient buffer: one collective per step."""
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world)
