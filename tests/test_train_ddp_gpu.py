"""GPU: the data-parallel TRAINING driver end to end on the hand-written kernels -- `torchrun --nproc-per-node 2 train.py -d blendedmvs`
with both ranks rehearsing on the one card of this box (MDF_SHARE_GPU=1) and gloo standing in for RCCL (two ranks cannot open one
device with RCCL): DistributedSampler split, HIP forward + backward per rank, flat-bucket gradient all-reduce, one-launch Adam,
BatchNorm buffer broadcast, rank-0 checkpoint.  The 8-GPU RCCL run itself is the driver's (train.py:24-26 is nn.DataParallel in the
reference)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from load import synthetic
from tools import data_io

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hipgraph", ["0", "1"])
def test_two_rank_training_epoch_on_one_card(tmp_path, hipgraph):
    """hipgraph = 1: the same epoch with every step after the first replayed from two recordings per rank (forward .. bucket gather, and
    Adam) around the eager all-reduce (mdfnet_hip/graphstep.py, MDF_TRAIN_HIPGRAPH)."""
    root = tmp_path / "data"
    synthetic.write_blendedmvs_set(str(root / "blendedmvs768x576"), scans=("scanA", "scanB"), nviews_total=7, width=128, height=96, short_pairs=False)
    pkg = os.path.dirname(os.path.dirname(data_io.__file__))
    env = dict(os.environ, MDF_DATA_ROOT=str(root), MDF_PTH_PATH=str(tmp_path / "pth"), MDF_MAX_EPOCH="1", OMP_NUM_THREADS="2", PYTHONPATH=pkg,
               MDF_SHARE_GPU="1", MDF_DIST_BACKEND="gloo", MDF_DUMP_RANK_STATE=str(tmp_path), MDF_TRAIN_HIPGRAPH=hipgraph)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29653 + int(hipgraph)), os.path.join(pkg, "train.py"), "-d", "blendedmvs"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ck = torch.load(str(tmp_path / "pth" / "blendedmvs_1.pth"), map_location="cpu")
    assert ck["epoch"] == 1 and len(ck["model"]) == 290
    loss = float(open(str(tmp_path / "pth" / "epoch_loss.txt")).read().split()[0])
    assert np.isfinite(loss) and loss > 0
    # both replicas hold the same parameters after the epoch (same averaged gradients, same optimizer arithmetic), they differ
    # from the initial ones, and the HIP training kernels were the ones that ran
    s0 = torch.load(str(tmp_path / "rank0_state.pt"), map_location="cpu")
    s1 = torch.load(str(tmp_path / "rank1_state.pt"), map_location="cpu")
    assert s0["device"].startswith("cuda") and s0["hip_training_calls"] > 0 and s1["hip_training_calls"] > 0
    moved = 0
    for k, v in s0["params"].items():
        assert torch.equal(v, s1["params"][k]), k
        moved += int(not torch.equal(v, s0["initial"][k]))
    assert moved > 150
