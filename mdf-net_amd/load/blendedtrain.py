"""BlendedMVS training set (counterpart of the reference's load/blendedtrain.py:9-106): scans from training_list.txt, per
scan cams/pair.txt (references with fewer than nviews sources are padded with their best source), depth range from
line 11 of the reference view's camera file (min ... max)."""
import os
import random

import numpy as np
import torch

from load.dtutrain import ground_truth_pyramid
from load.getpath import get_cam_path, get_depth_path, get_img_path
from tools import data_io


class LoadDataset(torch.utils.data.Dataset):
    def __init__(self, datasetpath, nviews=5, robust_train=False):
        super().__init__()
        self.datasetpath, self.nviews, self.robust_train = datasetpath, nviews, robust_train
        self.listfile = os.path.join(datasetpath, "training_list.txt")
        self.all_compose = self._compose()

    def __len__(self):
        return len(self.all_compose)

    def _compose(self):                                                                # blendedtrain.py:70-91
        out = []
        with open(self.listfile) as f:
            scans = [line.rstrip() for line in f.readlines()]
        for scan in scans:
            with open(os.path.join(self.datasetpath, "{}/cams/pair.txt".format(scan))) as f:
                for _ in range(int(f.readline())):
                    ref_view = int(f.readline().rstrip())
                    src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
                    if len(src_views) > 0:
                        if len(src_views) < self.nviews:
                            src_views += [src_views[0]] * (self.nviews - len(src_views))
                        out.append((scan, ref_view, src_views))
        return out

    @staticmethod
    def read_cam_file(filename):                                                       # blendedtrain.py:93-106
        with open(filename) as f:
            lines = [line.rstrip() for line in f.readlines()]
        extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
        intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
        rng = lines[11].split()
        return intrinsics, extrinsics, float(rng[0]), float(rng[3])

    def __getitem__(self, item):
        scan, ref_view, src_views = self.all_compose[item]
        rs_views = [ref_view] + src_views[:self.nviews - 1]
        if self.robust_train:                                                          # blendedtrain.py:27-30
            src_views = src_views[:7]
            index = random.sample(range(1, len(src_views), 1), self.nviews - 1)
            rs_views = [ref_view] + [src_views[i] for i in index]
        imgs, extrinsics, intrinsics, ref_depths, depth_range = [], [], [], {}, None
        for i, vid in enumerate(rs_views):
            imgs.append(data_io.read_img(get_img_path(self.datasetpath, scan, vid, mode="blendedmvs")))
            intrinsic, extrinsic, depth_min, depth_max = self.read_cam_file(get_cam_path(self.datasetpath, scan, vid, mode="blendedmvs"))
            extrinsics.append(extrinsic)
            intrinsics.append(intrinsic)
            if i == 0:
                depth = np.array(data_io.read_pfm(get_depth_path(self.datasetpath, scan, vid, mode="blendedmvs"))[0], dtype=np.float32)
                ref_depths = ground_truth_pyramid(depth)
                depth_range = np.array([depth_min, depth_max])
        return {"imgs": np.stack(imgs).transpose([0, 3, 1, 2]), "intrinsics": np.stack(intrinsics), "extrinsics": np.stack(extrinsics),
                "ref_depths": ref_depths, "depth_range": depth_range}
