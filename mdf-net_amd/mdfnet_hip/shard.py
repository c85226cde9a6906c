"""One process per GPU: shard independent eval items over ranks (no data-path collective) and reduce timings."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed when launched under torchrun (nccl == RCCL on ROCm, gloo on CPU)."""
    rank, world, local = world_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (MDF_DIST_BACKEND / MDF_SHARE_GPU: rehearsal knobs for a 1-GPU box -- gloo instead of RCCL, every rank on card 0)
        backend = backend or os.environ.get("MDF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("MDF_SHARE_GPU"):
            local = 0
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, port=None, base=None):
    """The environment of each of the n rank processes of one node (what torch.distributed.run would set)."""
    port = port or _free_port()
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        envs.append(e)
    return envs


def launch_ranks(n, argv, port=None, timeout=None):
    """Self-launch: start n FRESH child processes (one per GPU) running `argv`, relay rank 0's stdout, return the worst
    exit status.  Must be called before the calling process has touched the GPU: the parent never initialises HIP and
    never execs -- the children are ordinary subprocesses (a process that has initialised the GPU must not be replaced)."""
    procs = []
    for r, env in enumerate(rank_environments(n, port)):
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, text=True,
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL)))
    status = 0
    try:
        for line in procs[0].stdout:          # rank 0's record goes to our stdout; library chatter (e.g. gloo's) to stderr
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        for p in procs:
            rc = p.wait(timeout=timeout)
            status = status or rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return status
