"""One process per GPU: shard independent eval items over ranks (no data-path collective) and reduce timings."""
import os

import torch
import torch.distributed as dist


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed when launched under torchrun (nccl == RCCL on ROCm, gloo on CPU)."""
    rank, world, local = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_items(n_items, rank, world):
    """Items are independent (one per (scan, reference view)): rank r takes r, r+world, ... ."""
    return list(range(rank, n_items, world))


def max_over_ranks(value, device="cpu"):
    """Wall-clock of the job = slowest rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
