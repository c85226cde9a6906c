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


def pin_rank_affinity(local, local_world):
    """Give this rank process its own contiguous share of the host cores (the ones this process may use), in-process and
    BEFORE its first HIP call: N ranks that all roam over the same cores fight for the launch thread's core, and a wrapper
    (taskset / numactl) would mean exec'ing over a process.  No-op where the platform has no affinity API or the share
    would be empty.  -> the cores kept (sorted list) or None."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    per = len(cores) // max(1, local_world)
    if per < 1 or local_world <= 1:
        return None
    mine = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def launch_ranks(n, argv, port=None, timeout=None, log_dir=None, poll_s=0.2):
    """Self-launch: start n FRESH child processes (one per GPU) running `argv`, relay rank 0's stdout, return the worst
    exit status.  Must be called before the calling process has touched the GPU: the parent never initialises HIP and
    never execs -- the children are ordinary subprocesses (a process that has initialised the GPU must not be replaced),
    and a child is never restarted.

    Fail-fast: all children are polled; as soon as ANY rank exits non-zero (a HIP error or OOM during init, say) the others
    are killed and that status is returned within a second or two -- the survivors would otherwise sit in the rendezvous or
    in a collective until its timeout.  `timeout` (seconds, whole job) does the same with status 124.  Ranks above 0 write
    stdout + stderr to <log_dir>/rank<r>.log (default: a fresh temporary directory, named on stderr when a rank fails)."""
    import tempfile
    import threading
    import time
    log_dir = log_dir or os.environ.get("MDF_RANK_LOG_DIR") or tempfile.mkdtemp(prefix="mdf_ranks_")
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r, env in enumerate(rank_environments(n, port)):
        if r == 0:
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, text=True, stdout=subprocess.PIPE))
        else:
            f = open(os.path.join(log_dir, f"rank{r}.log"), "w")
            logs.append(f)
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=f, stderr=subprocess.STDOUT))

    def relay():      # rank 0's record goes to our stdout; library chatter (e.g. gloo's) to stderr
        for line in procs[0].stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    t0 = time.monotonic()
    status, failed = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed, status = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                failed, status = -1, 124
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        th.join(timeout=5)
        for f in logs:
            f.close()
    if failed is not None:
        what = f"timed out after {timeout} s" if failed < 0 else f"rank {failed} exited with status {status}"
        sys.stderr.write(f"launch_ranks: {what}; the other ranks were stopped (logs of ranks > 0: {log_dir})\n")
        if failed > 0:
            try:
                with open(os.path.join(log_dir, f"rank{failed}.log")) as f:
                    sys.stderr.write("".join(f.readlines()[-20:]))
            except OSError:
                pass
    return status
