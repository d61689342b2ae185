"""JarvisPredictor2D on MI355X (mirrors jarvis/prediction/jarvis2D.py:19-155).

Same constructor / forward signature and attributes (`centerDetect`,
`keypointDetect`), same `(None, None)` convention (centre heatmap maximum
<= 40).  `forward` is one native call (jh_predictor2d_* in include/jarvis_hip.h).
"""
import ctypes

import torch
import torch.nn as nn

from .. import _native as N
from .. import arch
from .._params import flat_state
from ..efficienttrack.efficienttrack import EfficientTrack


class _Native2D:
    def __init__(self, center_state, kp_state, cfg, img_h, img_w, batch, precision=None):
        c = N.PredictorConfig(
            1, cfg.KEYPOINTDETECT.NUM_JOINTS, int(cfg.CENTERDETECT.IMAGE_SIZE),
            cfg.KEYPOINTDETECT.BOUNDING_BOX_SIZE, 0.0, 0.0,
            arch.SIZE_IDS[cfg.CENTERDETECT.MODEL_SIZE], arch.SIZE_IDS[cfg.KEYPOINTDETECT.MODEL_SIZE],
            img_h, img_w, batch, 0, 0, 1, (ctypes.c_float * 3)(*cfg.DATASET.MEAN),
            (ctypes.c_float * 3)(*cfg.DATASET.STD), N.precision_id(precision))
        self.T, self.J = batch, cfg.KEYPOINTDETECT.NUM_JOINTS
        self.handle = ctypes.c_void_p()
        pc, pk = N.Params(center_state), N.Params(kp_state)
        N.check(N.lib().jh_predictor2d_create(pc.handle, pk.handle, ctypes.byref(c),
                                              ctypes.byref(self.handle)))

    def forward(self, frames):
        dev = frames.device
        pts = torch.empty((self.T, self.J, 2), device=dev, dtype=torch.int32)
        conf = torch.empty((self.T, self.J), device=dev)
        valid = torch.empty((self.T,), device=dev, dtype=torch.int32)
        fn = N.lib().jh_predictor2d_forward_u8 if frames.dtype == torch.uint8 else \
            N.lib().jh_predictor2d_forward
        N.check(fn(self.handle, N.ptr(frames), N.ptr(pts), N.ptr(conf), N.ptr(valid), N.stream()))
        return pts, conf, valid

    def close(self):
        if getattr(self, "handle", None) and N is not None and N._lib is not None:
            N.lib().jh_predictor2d_destroy(self.handle)
            self.handle = None

    __del__ = close


class JarvisPredictor2D(nn.Module):
    def __init__(self, cfg, weights_center_detect="latest", weights_keypoint_detect="latest",
                 trt_mode="off", precision=None):
        super().__init__()
        # trt_mode 'new' / 'previous' (jarvis2D.py:39-43) select the reduced-precision mode bf16x3, see
        # jarvis3D.precision_for_trt_mode
        from .jarvis3D import precision_for_trt_mode
        self.precision = precision_for_trt_mode(trt_mode, precision)
        self.trt_mode = trt_mode
        self.cfg = cfg
        self.centerDetect = EfficientTrack("CenterDetectInference", cfg, weights_center_detect).model
        self.keypointDetect = EfficientTrack("KeypointDetectInference", cfg,
                                             weights_keypoint_detect).model
        self.bbox_hw = int(cfg.KEYPOINTDETECT.BOUNDING_BOX_SIZE / 2)
        self.bounding_box_size = cfg.KEYPOINTDETECT.BOUNDING_BOX_SIZE
        self.center_detect_img_size = int(cfg.CENTERDETECT.IMAGE_SIZE)
        self._native = {}

    def native(self, img_h, img_w, batch=1):
        key = (img_h, img_w, batch)
        if key not in self._native:
            self._native[key] = _Native2D(flat_state(self.centerDetect),
                                          flat_state(self.keypointDetect), self.cfg, img_h, img_w,
                                          batch, self.precision)
        return self._native[key]

    def forward(self, img):
        """img (1,3,H,W) RGB in [0,1] -> (points2D (J,2) int64 pixels, confidences (J,))
        or (None, None)."""
        x = N.dev(img)
        pts, conf, valid = self.native(x.shape[2], x.shape[3], x.shape[0]).forward(x)
        if int(valid[0].item()) == 0:               # jarvis2D.py:121,150-153
            return None, None
        return pts[0].long(), conf[0]

    def forward_batch(self, imgs):
        """imgs (T,3,H,W) fp32 RGB or (T,H,W,3) uint8 BGR, independent images ->
        points2D (T,J,2) int32, confidences (T,J), valid (T) int32; no host sync."""
        if imgs.dtype == torch.uint8:
            x = N.dev(imgs, torch.uint8)
            return self.native(x.shape[1], x.shape[2], x.shape[0]).forward(x)
        x = N.dev(imgs)
        return self.native(x.shape[2], x.shape[3], x.shape[0]).forward(x)
