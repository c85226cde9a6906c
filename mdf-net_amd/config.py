"""Configuration + model composition (counterpart of the reference's config.py, which cannot travel).

Same public names (Args classes, LoadDTU/LoadBlendedMVS/LoadTanks, `model`) and the same composition as
config.py:186-218, so a reference-style `train.py` / `eval.py` keeps working.  Differences, all deliberate:
  * dataset / output roots come from the environment (MDF_DATA_ROOT, MDF_OUTPUT_ROOT) instead of /hy-tmp;
  * the visible-device list is left to the launcher (one process per GPU) instead of being forced to 0..7;
  * `build_model()` is exposed so callers can build further instances; the module-level singleton `model`
    is kept (train.py:13, eval.py:12 use it).
"""
import logging
import os
import random
import warnings

import numpy
import torch
import torch.nn as nn

warnings.filterwarnings("ignore")
logging.basicConfig(level=logging.INFO, format="%(asctime)s-%(levelname)s: %(message)s")

seed_id = 1
random.seed(seed_id)
numpy.random.seed(seed_id)
torch.manual_seed(seed_id)

DATA_ROOT = os.environ.get("MDF_DATA_ROOT", "/hy-tmp")
OUTPUT_ROOT = os.environ.get("MDF_OUTPUT_ROOT", os.path.join(DATA_ROOT, "outputs"))


class Args:
    def show_args(self):
        print(self.__class__.__name__ + ":")
        for k, v in self.__dict__.items():
            print("\t" + k, ":", v)

    def get_device(self, parallel):
        """One process per GPU: the device is this rank's (LOCAL_RANK), never a device list."""
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed_id)
            # (MDF_SHARE_GPU: rehearsal knob for a 1-GPU box -- every rank on card 0; never set in production)
            return torch.device("cuda", 0 if os.environ.get("MDF_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0")))
        return torch.device("cpu")


class _TrainBase(Args):
    def __init__(self, batch_size, nworks):
        self.nviews, self.robust = 5, True
        self.start_epoch, self.max_epoch = 1, int(os.environ.get("MDF_MAX_EPOCH", "30"))
        self.batch_size, self.nworks = batch_size, nworks
        self.lr, self.factor = 1e-3, 0.9
        self.pth_path = os.environ.get("MDF_PTH_PATH", "pth")
        os.makedirs(self.pth_path, exist_ok=True)
        self.parallel = True   # data parallel = one process per GPU + RCCL all-reduce (not nn.DataParallel)
        self.DEVICE = self.get_device(self.parallel)
        self.show_args()


class TrainArgs(_TrainBase):          # config.py:47-66
    def __init__(self):
        super().__init__(batch_size=4, nworks=2)


class BlendedMVSArgs(_TrainBase):     # config.py:72-89
    def __init__(self):
        super().__init__(batch_size=6, nworks=3)


class EvalArgs(Args):                 # config.py:95-101
    def __init__(self):
        self.output_path = OUTPUT_ROOT
        os.makedirs(self.output_path, exist_ok=True)
        self.parallel = False
        self.DEVICE = self.get_device(self.parallel)


class EvalDTU(EvalArgs):              # config.py:104-111
    def __init__(self):
        super().__init__()
        self.batch_size, self.nworks, self.nviews = 1, 1, 5
        self.show_args()


class EvalTanks(EvalArgs):            # config.py:114-121
    def __init__(self):
        super().__init__()
        self.batch_size, self.nworks, self.nviews = 1, 1, 11
        self.show_args()


def _scan_ids(spec):
    """ "2 6-8 14" -> [2, 6, 7, 8, 14] """
    out = []
    for tok in spec.split():
        lo, _, hi = tok.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


class DatasetsArgs(Args):
    def __init__(self):
        self.root_dir = DATA_ROOT


class LoadDTU(DatasetsArgs):          # config.py:131-152
    def __init__(self):
        super().__init__()
        self.train_root = os.path.join(self.root_dir, "dtu640x512")
        self.train_pair = os.path.join(self.train_root, "Cameras", "pair.txt")
        self.train_label = _scan_ids("2 6-8 14 16 18-20 22 30-31 36 39 41-42 44-47 50-53 55 57-58 60-61 63-65 68-72 74 76 83-85 87-105 107-109 111-113 115-116 119-128")   # the 79 DTU training scans
        self.train_lighting_label = list(range(7))
        self.train_robust = True
        self.eval_root = os.path.join(self.root_dir, "dtu1600x1200")
        self.eval_pair = os.path.join(self.eval_root, "pair.txt")
        self.eval_label = [int(s) for s in os.environ.get("MDF_DTU_SCANS", "11").split(",")]
        self.show_args()


class LoadBlendedMVS(DatasetsArgs):   # config.py:158-163
    def __init__(self):
        super().__init__()
        self.train_root = os.path.join(self.root_dir, "blendedmvs768x576")
        self.show_args()


class LoadTanks(DatasetsArgs):        # config.py:169-180
    def __init__(self, tanks_set="intermediate"):
        super().__init__()
        self.eval_root = os.path.join(self.root_dir, "TankandTemples", tanks_set)
        self.scenelist = {"intermediate": ["Family", "Francis", "Horse", "Lighthouse", "M60", "Panther", "Playground",
                                           "Train"],
                          "advanced": ["Auditorium", "Ballroom", "Courtroom", "Museum", "Temple", "Palace"]}[tanks_set]
        self.show_args()


# ----------------------------------------------------------------------------- net args (config.py:186-218)
from net import core  # noqa: E402
from net.unit import scale as _scale, backbone, regress, refine  # noqa: E402
from net.unit.depthhypos import HyposByFit  # noqa: E402
from net.unit.homoaggregate import VectorAggregate  # noqa: E402
from net.unit.regular import RegularNet_4Scales, RegularNet_3Scales  # noqa: E402

stages = 4
scale = _scale.scale_cam
chs = (8, 16, 32, 64)
ndepths = (48, 24, 8)
curve_calss = [None, "gauss1", "laplace"]
prob_thresh = (0.0, 0.95, 1e-5)
ngroups = (32, 16, 8)


def build_model():
    """A fresh CoreNet composed exactly like the reference's singleton."""
    hypos = nn.ModuleList([HyposByFit(ndepths[i], curve_calss[i], prob_thresh[i]) for i in range(stages - 1)])
    aggre = nn.ModuleList([VectorAggregate(ngroups[i]) for i in range(stages - 1)])
    regular = nn.ModuleList([RegularNet_3Scales(ngroups[0])] + [RegularNet_4Scales(c) for c in ngroups[1:]])
    return core.CoreNet(backbone.FPN_4Scales(chs), hypos, scale, aggre, regular,
                        [regress.depth_regression, regress.confidence_regress], refine.RefineNet2())


model = build_model()
