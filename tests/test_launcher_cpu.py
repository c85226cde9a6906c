"""`bench.py --gpus N` without a torchrun environment starts the N ranks itself (mdfnet_hip/shard.py:launch_ranks):
fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout relayed, the parent never touching
the GPU.  Rehearsed here on the CPU with a gloo worker standing in for bench.py's ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mdf-net_amd")

WORKER = r'''
import os, sys, json
sys.path[:0] = [%(pkg)r]
import torch, torch.distributed as dist
from mdfnet_hip import shard
rank, world, local = shard.init("gloo")
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t)
shard.barrier()
print(json.dumps({"rank": rank, "n_gpus": world, "local": local, "sum": float(t), "addr": os.environ["MASTER_ADDR"]}), flush=True)
dist.destroy_process_group()
sys.exit(3 if (rank == 1 and "--fail" in sys.argv) else 0)
'''


def test_rank_environments():
    sys.path[:0] = [PKG]
    from mdfnet_hip import shard
    envs = shard.rank_environments(4, port=29777, base={"X": "1"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29777" for e in envs)
    assert all(e["X"] == "1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_launch_ranks_relays_rank0_and_status(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"pkg": PKG})
    drv = ("import sys; sys.path[:0]=[%r]; from mdfnet_hip import shard; "
           "raise SystemExit(shard.launch_ranks(3, [%r] + sys.argv[1:]))" % (PKG, str(script)))
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0] == {"rank": 0, "n_gpus": 3, "local": 0, "sum": 6.0, "addr": "127.0.0.1"}   # only rank 0 is relayed
    r = subprocess.run([sys.executable, "-c", drv, "--fail"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3                                                                 # a failing rank fails the job


DEAD_RANK_WORKER = r'''
import os, sys, time
sys.path[:0] = [%(pkg)r]
rank = int(os.environ["RANK"])
if rank == 1:
    sys.stderr.write("rank 1: simulated HIP error during init\\n")
    sys.exit(3)                       # dies before the rendezvous, like a HIP error / OOM during init
import torch, torch.distributed as dist
from mdfnet_hip import shard
shard.init("gloo")                    # ranks 0 and 2 wait here (rendezvous) for the rank that never comes
shard.barrier()
print("{}", flush=True)
'''


def test_launch_ranks_fails_fast_when_a_rank_dies(tmp_path):
    """rank 1 exits with status 3 during init while rank 0 is blocked in the rendezvous/barrier -- the
    parent must return that status within seconds (it used to read rank 0's stdout to EOF, i.e. until the collective timeout),
    stop the surviving ranks, and keep the dead rank's output."""
    import time
    script = tmp_path / "worker.py"
    script.write_text(DEAD_RANK_WORKER % {"pkg": PKG})
    logs = tmp_path / "logs"
    drv = ("import sys; sys.path[:0]=[%r]; from mdfnet_hip import shard; "
           "raise SystemExit(shard.launch_ranks(3, [%r], log_dir=%r))" % (PKG, str(script), str(logs)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=120)
    took = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert took < 10.0, took
    assert "rank 1 exited with status 3" in r.stderr and "simulated HIP error" in r.stderr
    assert "simulated HIP error" in (logs / "rank1.log").read_text()
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no record from a failed job


def test_launch_ranks_timeout(tmp_path):
    script = tmp_path / "sleeper.py"
    script.write_text("import time; time.sleep(600)\n")
    sys.path[:0] = [PKG]
    from mdfnet_hip import shard
    import time
    t0 = time.time()
    rc = shard.launch_ranks(2, [str(script)], timeout=1.0, log_dir=str(tmp_path / "logs"))
    assert rc == 124 and time.time() - t0 < 15.0


def test_pin_rank_affinity_partitions_the_cores():
    sys.path[:0] = [PKG]
    from mdfnet_hip import shard
    if not hasattr(os, "sched_getaffinity"):
        return
    before = sorted(os.sched_getaffinity(0))
    try:
        if len(before) >= 2:
            mine = shard.pin_rank_affinity(1, 2)
            assert mine == before[len(before) // 2:2 * (len(before) // 2)] and sorted(os.sched_getaffinity(0)) == mine
        assert shard.pin_rank_affinity(0, 1) is None                 # a single rank keeps everything
    finally:
        os.sched_setaffinity(0, before)


def test_bench_parent_launches_before_touching_the_gpu():
    """bench.py's self-launch branch sits before the first CUDA call of main()."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("shard.launch_ranks") < main.index("torch.cuda.")


def test_eval_and_bench_share_the_in_flight_default():
    sys.path[:0] = [PKG]
    import inspect
    import importlib.util
    from mdfnet_hip import pipeline
    spec = importlib.util.spec_from_file_location("mdf_eval_mod", os.path.join(PKG, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    assert inspect.signature(ev.run_eval).parameters["in_flight"].default is None      # -> pipeline.DEFAULT_IN_FLIGHT
    assert "DEFAULT_IN_FLIGHT" in open(os.path.join(ROOT, "bench.py")).read() and pipeline.DEFAULT_IN_FLIGHT == 3


def test_feature_cache_is_lru_and_keeps_the_current_items_views():
    sys.path[:0] = [PKG]
    import importlib.util
    spec = importlib.util.spec_from_file_location("mdf_eval_mod2", os.path.join(PKG, "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    c = ev.FeatureCache(max_items=3)
    for k in "abc":
        c[k] = k.upper()
    assert c["a"] == "A"                        # hit: 'a' becomes the youngest
    c["d"] = "D"                                # evicts the least recently used = 'b'
    assert set(c) == {"a", "c", "d"}
    c.pin(["v0", "v1", "v2", "v3", "v4"])       # an item with more views than the cache holds
    for k in ["v0", "v1", "v2", "v3", "v4"]:
        c[k] = k
    assert all(k in c for k in ["v0", "v1", "v2", "v3", "v4"])      # none of the item's own views was evicted
    c.pin(["w0"])
    c["w0"] = 1
    assert "w0" in c and len(c) <= 5
