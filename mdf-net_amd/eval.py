"""Evaluation driver (counterpart of the reference's eval.py:10-89): same CLI (-p/-d/-s), same outputs
(<out>/<scan>/depth_est/%08d.pfm|.png, <out>/<scan>/confidence/%08d.pfm), same per-item print line — plus:
one process per GPU with the item list sharded over ranks (`torchrun --nproc-per-node N eval.py ...`), three items in flight
on three HIP streams (pipeline.DEFAULT_IN_FLIGHT, the same constant bench.py uses) (the printed per-item time is the interval between completions), the cross-item feature cache, and no
collective on the data path."""
import argparse
import logging
import os
import time

import torch
from torch.utils.data import DataLoader, Subset

from mdfnet_hip import hostmirror, shard
from tools.data_io import save_pfm, write_depth_img


class FeatureCache(dict):
    """Feature pyramids per (scan, view id), bounded and least-recently-used (a 1600x1184 image's pyramid is 53 MB; 64
    entries = 3.4 GB of 288).  `pin(keys)` names the views of the item being assembled: they are never evicted, so an item
    with more views than `max_items` (Tanks&Temples, nviews=11, small caches) still finds all its pyramids."""

    def __init__(self, max_items=64):
        super().__init__()
        self.max_items = max_items
        self.pinned = ()

    def pin(self, keys):
        self.pinned = tuple(keys)

    def __getitem__(self, k):
        v = super().pop(k)                      # move to the young end on a hit
        super().__setitem__(k, v)
        return v

    def __setitem__(self, k, v):
        if k in self:
            super().pop(k)
        while len(self) >= self.max_items:
            victim = next((old for old in self if old not in self.pinned), None)
            if victim is None:
                break                           # everything left belongs to the current item: grow rather than fail
            del self[victim]
        super().__setitem__(k, v)


def run_eval(model, dataset, device, output_path, rank=0, world=1, nworks=1, log=print, cache_features=True, in_flight=None):
    """Shard `dataset` over ranks, run `model` item by item, write PFM/PNG.  Returns (n_items_this_rank, seconds).
    With cache_features (and a model that accepts it) every image goes through the feature pyramid once per scan instead
    of once per item it appears in.  in_flight > 1 issues items round-robin on that many HIP streams
    (mdfnet_hip/pipeline.py): the next item fills the idle tails of the current one and the PFM writes overlap with GPU
    work.  Outputs are identical either way."""
    from mdfnet_hip.pipeline import DEFAULT_IN_FLIGHT, InFlight
    in_flight = DEFAULT_IN_FLIGHT if in_flight is None else in_flight
    idx = shard.shard_items(len(dataset), rank, world)
    loader = DataLoader(Subset(dataset, idx), batch_size=1, num_workers=nworks, shuffle=False,
                        pin_memory=(device.type == "cuda"), drop_last=False)
    model.eval()
    import inspect
    cache = FeatureCache() if (cache_features and "feature_cache" in inspect.signature(model.forward).parameters) else None
    state = {"last": time.time(), "n": 0}

    def finished(tag, out):
        it, data = tag
        now = time.time()
        dt, state["last"] = now - state["last"], now
        state["n"] += 1
        mem = torch.cuda.max_memory_allocated(device) / (1024 ** 2) if device.type == "cuda" else 0.0
        log("batch: " + str(it + 1) + "/" + str(len(loader)) + " time: {:.3f}".format(dt) + " memory: " + str(mem) + "MB")
        for name, depth, conf in zip(data["filename"], out["depth"], out["confidence"]):
            dfile = os.path.join(output_path, name.format("depth_est", ".pfm"))
            cfile = os.path.join(output_path, name.format("confidence", ".pfm"))
            os.makedirs(os.path.dirname(dfile), exist_ok=True)
            os.makedirs(os.path.dirname(cfile), exist_ok=True)
            save_pfm(dfile, depth.cpu().numpy())
            write_depth_img(os.path.join(output_path, name.format("depth_est", ".png")), depth.cpu().numpy())
            save_pfm(cfile, conf.cpu().numpy())
            logging.info("save depth file in: " + dfile)

    pipe = InFlight(device, in_flight if device.type == "cuda" else 1, done=finished)
    t_begin = time.time()
    with torch.no_grad():
        for it, data in enumerate(loader):
            batch = {k: v.to(device, non_blocking=True) for k, v in data.items() if isinstance(v, torch.Tensor)}
            if device.type == "cuda":   # the loader's CPU tensors ARE the host mirrors: no device->host hop later
                for k in ("extrinsics", "intrinsics", "depth_range"):
                    hostmirror.put(batch[k], data[k])
            if cache is not None and "view_ids" in data and batch["imgs"].shape[0] == 1:
                keys = [(data["scan"][0], int(v)) for v in data["view_ids"][0]]
                fn = lambda b=batch, k=keys: model(b["imgs"], b["extrinsics"], b["intrinsics"], b["depth_range"],
                                                   feature_cache=cache, view_keys=k)
            else:
                fn = lambda b=batch: model(b["imgs"], b["extrinsics"], b["intrinsics"], b["depth_range"])
            pipe.submit(fn, tag=(it, data), keep=batch)
        pipe.drain()
    return len(idx), time.time() - t_begin


def main():
    import config
    parser = argparse.ArgumentParser(description="eval parameter setting")
    parser.add_argument("-p", "--pre_model", default=None, type=str, help="Pre training model")
    parser.add_argument("-d", "--dataset", default="dtu", type=str, choices=["dtu", "tanks"], help="Set dataset")
    parser.add_argument("-s", "--set", default="intermediate", type=str, choices=["intermediate", "advanced"])
    args = parser.parse_args()
    logging.info(args)
    rank, world, local = shard.init()
    if args.dataset == "dtu":
        load_args, eval_args = config.LoadDTU(), config.EvalDTU()
        from load.dtueval import LoadDataset
        dataset = LoadDataset(datasetpath=load_args.eval_root, pairpath=load_args.eval_pair,
                              scencelist=load_args.eval_label, nviews=eval_args.nviews)
    else:
        load_args, eval_args = config.LoadTanks(tanks_set=args.set), config.EvalTanks()
        from load.tankseval import LoadDataset
        dataset = LoadDataset(datasetpath=load_args.eval_root, scenelist=load_args.scenelist, nviews=eval_args.nviews)
    model = config.model
    if args.pre_model is not None:
        model.load_state_dict(torch.load(args.pre_model, map_location="cpu")["model"])
    model.to(eval_args.DEVICE)
    n, busy = run_eval(model, dataset, eval_args.DEVICE, eval_args.output_path, rank, world, eval_args.nworks)
    dev = eval_args.DEVICE if eval_args.DEVICE.type == "cuda" else "cpu"
    total, slowest = shard.sum_over_ranks(n, dev), shard.max_over_ranks(busy, dev)
    if rank == 0:
        logging.info("items: %d on %d rank(s); model time of the slowest rank %.3f s -> %.2f views/s",
                     int(total), world, slowest, total / max(slowest, 1e-9))


if __name__ == "__main__":
    main()
