"""HybridNet convenience wrapper, inference mode only (mirrors
jarvis/hybridnet/hybridnet.py:31-131; the training loop is out of scope)."""
import os

import torch

from .model import HybridNetBackbone


class HybridNet:
    def __init__(self, mode, cfg, weights=None, efficienttrack_weights=None, run_name=None):
        if mode != "inference":
            raise NotImplementedError("only mode='inference' is implemented (no backward pass)")
        self.mode = mode
        self.cfg = cfg
        self.model = HybridNetBackbone(cfg, efficienttrack_weights)
        self.load_weights(weights)
        self.model.requires_grad_(False)
        self.model.eval()
        self.model = self.model.cuda()

    def load_weights(self, weights_path=None):
        """`strict=True`, silent False when the file is missing (hybridnet.py:84-97).
        A state dict may be passed instead of a path."""
        if isinstance(weights_path, dict):
            self.model.load_state_dict(weights_path, strict=True)
            return True
        if weights_path == "latest":
            weights_path = self.get_latest_weights()
        if weights_path is None:
            return True
        if not os.path.isfile(weights_path):
            return False
        self.model.load_state_dict(torch.load(weights_path, map_location="cpu"), strict=True)
        return True

    def get_latest_weights(self):
        """newest run directory holding a *_final.pth (hybridnet.py:118-131)"""
        search = os.path.join(self.cfg.PARENT_DIR, "projects", self.cfg.PROJECT_NAME, "models",
                              "HybridNet")
        if not os.path.isdir(search):
            return None
        runs = sorted((os.path.join(search, d) for d in os.listdir(search)),
                      key=os.path.getmtime, reverse=True)
        name = "HybridNet-%s_final.pth" % self.cfg.KEYPOINTDETECT.MODEL_SIZE
        for run in runs:
            if os.path.isfile(os.path.join(run, name)):
                return os.path.join(run, name)
        return None
