"""EfficientTrack convenience wrapper, inference modes only (mirrors
jarvis/efficienttrack/efficienttrack.py:33-183; training is out of scope)."""
import os

import torch

from .model import EfficientTrackBackbone


class EfficientTrack:
    def __init__(self, mode, cfg, weights=None, run_name=None):
        if mode not in ("CenterDetectInference", "KeypointDetectInference"):
            raise NotImplementedError("only the *Inference modes are implemented")
        self.mode = mode
        self.main_cfg = cfg
        self.cfg = cfg.CENTERDETECT if mode == "CenterDetectInference" else cfg.KEYPOINTDETECT
        self.model = EfficientTrackBackbone(self.cfg, model_size=self.cfg.MODEL_SIZE,
                                            output_channels=self.cfg.NUM_JOINTS)
        self.load_weights(weights)
        self.model = self.model.cuda()
        self.model.requires_grad_(False)
        self.model.eval()

    def load_weights(self, weights_path=None):
        """`strict=False`, silent False when the file is missing
        (efficienttrack.py:90-113).  A state dict may be passed instead of a path."""
        if isinstance(weights_path, dict):
            self.model.load_state_dict(weights_path, strict=False)
            return True
        if weights_path == "latest":
            weights_path = self.get_latest_weights()
        if weights_path is None:
            return True
        if not os.path.isfile(weights_path):
            return False
        self.model.load_state_dict(torch.load(weights_path, map_location="cpu"), strict=False)
        return True

    def get_latest_weights(self):
        kind = "CenterDetect" if self.mode.startswith("CenterDetect") else "KeypointDetect"
        search = os.path.join(self.main_cfg.PARENT_DIR, "projects", self.main_cfg.PROJECT_NAME,
                              "models", kind)
        if not os.path.isdir(search):
            return None
        runs = sorted((os.path.join(search, d) for d in os.listdir(search)),
                      key=os.path.getmtime, reverse=True)
        name = "EfficientTrack-%s_final.pth" % self.cfg.MODEL_SIZE
        for run in runs:
            if os.path.isfile(os.path.join(run, name)):
                return os.path.join(run, name)
        return None
