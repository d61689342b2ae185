"""Import harness for the upstream reference (THIS CONTAINER ONLY).

Used solely by tests/golden/make_golden.py to produce the committed golden
vectors.  /root/reference does not exist on the GPU box; nothing imported at
test/bench time depends on this module.

Recipe: stub the import-time-only third-party modules,
provide a 3-line stand-in for torchvision's tensor resize, and neutralise the
reference's hard-coded `.cuda()` / device='cuda' so the path runs on CPU.
"""
import sys
import types
from unittest import mock

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"
_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    for name in ("cv2", "imgaug", "imgaug.augmenters", "imgaug.augmentables",
                 "imgaug.augmentables.kps", "streamlit", "yacs", "yacs.config",
                 "tensorboard", "torch.utils.tensorboard", "ruamel",
                 "ruamel.yaml", "seaborn", "inquirer", "joblib_stub"):
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def _resize(img, size):
        return F.interpolate(img, size=list(size), mode="bilinear",
                             align_corners=False)
    tvf.resize = _resize
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.IntTensor = torch.IntTensor

    def _wrap(fn):
        def inner(*a, **k):
            dev = k.get("device", None)
            if dev is not None and "cuda" in str(dev):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    for fname in ("ones", "zeros", "tensor", "arange", "empty", "full"):
        setattr(torch, fname, _wrap(getattr(torch, fname)))
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def ns(**kw):
    return types.SimpleNamespace(**kw)


def make_cfg(num_cameras=12, num_joints=23, roi=128, spacing=2, bbox=256,
             center_size=256, center_model="small", kp_model="small"):
    return ns(
        PARENT_DIR="/root/reference", PROJECT_NAME="golden",
        DATASET=ns(DATASET_ROOT_DIR="datasets", DATASET_2D="x", DATASET_3D="x",
                   MEAN=[0.485, 0.456, 0.406], STD=[0.229, 0.224, 0.225]),
        CENTERDETECT=ns(MODEL_SIZE=center_model, NUM_JOINTS=1,
                        IMAGE_SIZE=center_size),
        KEYPOINTDETECT=ns(MODEL_SIZE=kp_model, NUM_JOINTS=num_joints,
                          BOUNDING_BOX_SIZE=bbox),
        HYBRIDNET=ns(NUM_CAMERAS=num_cameras, ROI_CUBE_SIZE=roi,
                     GRID_SPACING=spacing),
    )
