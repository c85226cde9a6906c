"""Training driver (counterpart of the reference's train.py:11-110): same CLI (-p/-d/-l), Adam lr 1e-3 with the poly
decay lr*(1-(e-1)/E)^0.9, per-epoch checkpoint {'epoch','model'} without a `module.` prefix, epoch_loss.txt.
Data parallel = one process per GPU (`torchrun --nproc-per-node N train.py ...`): flat-bucket gradient all-reduce
over RCCL (mdfnet_hip/ddp.py) instead of nn.DataParallel.  On a GPU the model's training mode runs the hand-written
training kernels (mdfnet_hip/train_ops.py: forward + backward of every slot, the loss, one-launch Adam).  On a CPU device the
driver is a REHEARSAL (gloo runs of the data-parallel wiring on machines without a GPU): it selects the stock-op backend of
mdf-net_amd/rehearsal explicitly and says so; the product's slots themselves refuse CPU tensors."""
import argparse
import logging
import os
import time

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from mdfnet_hip import ddp, shard
from net import loss as loss_mod
from tools.data_io import tocuda


HIP_GRAPH = bool(int(os.environ.get("MDF_TRAIN_HIPGRAPH", "0")))   # opt-in: the step recorded once, replayed per batch (mdfnet_hip/graphstep.py)


def train_one_epoch(model, bucket, optimizer, loss_criterion, batches, device, log=print, epoch=0):
    model.train()
    total = 0.0
    for it, data in enumerate(batches):
        host = data
        data = tocuda(data, device)
        t0 = time.time()
        if HIP_GRAPH and torch.device(device).type == "cuda":
            # the first batch is stepped eagerly (it is the recording's one warm-up step), every later one is a replay; cameras and
            # range are handed over as the loader's HOST tensors -- their arithmetic is the host's (scale.py, base.py)
            args = (data["imgs"], host["extrinsics"], host["intrinsics"], host["depth_range"], data["ref_depths"])
            step = getattr(model, "_mdf_graph_step", None)
            if step is None:
                from mdfnet_hip.graphstep import GraphedTrainStep
                example = (data["imgs"], data["extrinsics"], data["intrinsics"], data["depth_range"], data["ref_depths"])
                step = model._mdf_graph_step = GraphedTrainStep(model, loss_criterion, bucket, optimizer, example, warmup=1)
                loss = step.warmup_loss
            else:
                loss = step(*args)
        else:
            out = model(data["imgs"], data["extrinsics"], data["intrinsics"], data["depth_range"])
            loss = loss_criterion(out, data["ref_depths"], data["depth_range"])
            bucket.zero_grad()
            loss.backward()
            bucket.allreduce_gradients()
            optimizer.step()
        cur = loss.detach().item()
        total += cur
        log("\r" + "epoch: " + str(epoch) + " batch: " + str(it + 1) + "/" + str(len(batches))
            + " time:{: .3f}".format(time.time() - t0) + " loss:{: .5f}\t".format(cur), end="", flush=True)
    return total / max(len(batches), 1)


def split_global_batch(batch_size, world):
    """The reference's batch_size is the GLOBAL batch (nn.DataParallel scatters it over the GPUs, train.py:24-26, config.py:54,74);
    here every rank takes batch_size / world samples.  A batch the ranks cannot share equally would silently change the
    optimisation (6 on 4 ranks -> 4, 6 on 8 ranks -> 8): refuse it, naming the choices."""
    if batch_size % world != 0:
        ok = [d for d in range(1, batch_size + 1) if batch_size % d == 0]
        raise SystemExit(f"train.py: global batch_size {batch_size} does not divide over {world} ranks (each rank must take the same "
                         f"share); use a world size in {ok} or change batch_size in config.py")
    return batch_size // world


def main():
    import config
    parser = argparse.ArgumentParser(description="train parameter setting")
    parser.add_argument("-p", "--pre_model", default=None, type=str)
    parser.add_argument("-d", "--dataset", default="dtu", type=str, choices=["dtu", "blendedmvs"])
    parser.add_argument("-l", "--cmd_label", default="", type=str)
    args = parser.parse_args()
    rank, world, local = shard.init()
    if args.dataset == "dtu":
        load_args, train_args = config.LoadDTU(), config.TrainArgs()
        from load.dtutrain import LoadDataset
        dataset = LoadDataset(datasetpath=load_args.train_root, pairpath=load_args.train_pair, scencelist=load_args.train_label,
                              lighting_label=load_args.train_lighting_label, nviews=train_args.nviews,
                              robust_train=train_args.robust)
    else:
        load_args, train_args = config.LoadBlendedMVS(), config.BlendedMVSArgs()
        from load.blendedtrain import LoadDataset
        dataset = LoadDataset(datasetpath=load_args.train_root, nviews=train_args.nviews, robust_train=train_args.robust)
    device = train_args.DEVICE
    model = config.model
    start_epoch = train_args.start_epoch
    if args.pre_model is not None:
        ckpt = torch.load(args.pre_model, map_location="cpu")
        start_epoch = ckpt["epoch"] + 1
        model.load_state_dict(ckpt["model"])
    model.to(device)
    if torch.device(device).type != "cuda":
        import rehearsal
        rehearsal.enable()
        logging.getLogger(__name__).warning("train.py on %s: REHEARSAL run on the stock-op backend (mdf-net_amd/rehearsal), not the product's kernels", device)
    dump_dir = os.environ.get("MDF_DUMP_RANK_STATE")       # test hook (tests/test_train_ddp_gpu.py): every rank's final parameters
    if dump_dir:
        from mdfnet_hip import ops as _ops
        _ops.count_begin()
    bucket = ddp.FlatBucket(model)
    bucket.broadcast_parameters(0)
    # train.py:14 -- Adam(lr); the same update as ONE launch over the flat bucket (mdfnet_hip/optim.py)
    from mdfnet_hip.optim import FlatAdam
    optimizer = FlatAdam(bucket, lr=train_args.lr)
    initial = {k: v.detach().cpu().clone() for k, v in model.named_parameters()} if dump_dir else None
    criterion = loss_mod.Loss().to(device)
    per_rank = split_global_batch(train_args.batch_size, world)
    sampler = DistributedSampler(dataset, world, rank, shuffle=True, drop_last=True) if world > 1 else None
    batches = DataLoader(dataset, batch_size=per_rank, shuffle=(sampler is None), sampler=sampler,
                         num_workers=train_args.nworks, drop_last=True, pin_memory=True)
    for epoch in range(start_epoch, train_args.max_epoch + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        optimizer.param_groups[0]["lr"] = train_args.lr * ((1 - (epoch - 1) / train_args.max_epoch) ** train_args.factor)
        mean_loss = train_one_epoch(model, bucket, optimizer, criterion, batches, device, epoch=epoch)
        mean_loss = shard.sum_over_ranks(mean_loss, device if device.type == "cuda" else "cpu") / world
        bucket.broadcast_buffers(0)
        if rank == 0:
            logging.info("epoch: " + str(epoch) + " loss:" + str(mean_loss))
            with open(os.path.join(train_args.pth_path, "epoch_loss.txt"), "a") as f:
                f.write(str(mean_loss) + "\n")
            torch.save({"epoch": epoch, "model": model.state_dict()},
                       os.path.join(train_args.pth_path, args.dataset + "_" + str(epoch) + ".pth"))
    if dump_dir:
        calls = _ops.count_end()
        trained = sum(n for k, n in calls.items() if k.endswith("_bwd") or "wgrad" in k or "train" in k)
        torch.save({"device": str(device), "params": {k: v.detach().cpu() for k, v in model.named_parameters()}, "initial": initial,
                    "hip_training_calls": trained}, os.path.join(dump_dir, f"rank{rank}_state.pt"))


if __name__ == "__main__":
    main()
