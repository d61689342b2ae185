"""HybridNetBackbone on MI355X (mirrors jarvis/hybridnet/model.py:20-90).

Same constructor, same state dict (`effTrack.*` + `v2vNet.*`, 188 tensors for
small / 23 joints) and same forward signature / return tuple.  The forward pass
is one native call: KeypointDetect on all camera crops, reprojection gather,
V2V, soft-argmax -- no intermediate ever leaves the GPU.
"""
import torch

from .. import _native as N
from .._params import NativeModule, flat_state, weights_fingerprint
from .._predictor import NativePredictor
from ..efficienttrack.model import EfficientTrackBackbone
from .repro_layer import ReprojectionLayer
from .v2vnet import V2VNet


class HybridNetBackbone(NativeModule):
    def __init__(self, cfg, efficienttrack_weights=None):
        super().__init__()
        self.cfg = cfg
        self.root_dir = cfg.DATASET.DATASET_ROOT_DIR
        self.grid_spacing = cfg.HYBRIDNET.GRID_SPACING
        self.grid_size = cfg.HYBRIDNET.ROI_CUBE_SIZE
        self.effTrack = EfficientTrackBackbone(cfg.KEYPOINTDETECT,
                                               model_size=cfg.KEYPOINTDETECT.MODEL_SIZE,
                                               output_channels=cfg.KEYPOINTDETECT.NUM_JOINTS)
        if efficienttrack_weights is not None:
            self.effTrack.load_state_dict(torch.load(efficienttrack_weights, map_location="cpu"),
                                          strict=True)
        self.reproLayer = ReprojectionLayer(cfg)
        self.v2vNet = V2VNet(cfg.KEYPOINTDETECT.NUM_JOINTS, cfg.KEYPOINTDETECT.NUM_JOINTS)

    def _predictor(self, batch, bbox):
        fp = weights_fingerprint(self)                      # reloads of effTrack / v2vNet count
        key = (batch, bbox, fp)
        pr = self._plans.get(key)
        if pr is None:
            # drop the plans of OLDER weights only; plans of other (batch, bbox) shapes built from
            # the current weights stay (alternating batch sizes must not rebuild every call)
            for k in [k for k in self._plans if k[2] != fp]:
                self._plans.pop(k).close()
            c = self.cfg
            pr = NativePredictor(
                None, flat_state(self), num_cameras=c.HYBRIDNET.NUM_CAMERAS,
                num_joints=c.KEYPOINTDETECT.NUM_JOINTS, center_size=bbox, bbox=bbox,
                roi_cube_size=c.HYBRIDNET.ROI_CUBE_SIZE, grid_spacing=c.HYBRIDNET.GRID_SPACING,
                img_h=bbox, img_w=bbox, mean=[0, 0, 0], std=[1, 1, 1],
                kp_model=c.KEYPOINTDETECT.MODEL_SIZE, time_batch=batch)
            self._plans[key] = pr
        return pr

    def forward(self, imgs, img_size, centerHM, center3D, cameraMatrices, intrinsicMatrices,
                distortionCoefficients):
        """imgs (b,C,3,B,B) normalised crops; centerHM (b,C,2) int; center3D (b,3) int;
        calibration (b,C,...) (entry 0 is used, as in the reference, repro_layer.py:113-117)
        -> (heatmap_final (b,J,Gh,Gh,Gh), heatmaps_padded (b,C,J,hs,hs),
            points3D (b,J,3), confidences (b,J))."""
        for name, cls in (("effTrack", EfficientTrackBackbone), ("v2vNet", V2VNet)):
            if not isinstance(getattr(self, name, None), cls):
                # the reference's trt_mode seam assigns compiled callables here (jarvis3D.py:65-69): this forward is one
                # native launch plan built from the modules' weights, a foreign callable would be silently ignored
                raise RuntimeError("HybridNetBackbone.%s has been replaced by a %s; the native forward only runs its own "
                                   "%s (load weights with load_state_dict())"
                                   % (name, type(getattr(self, name, None)).__name__, cls.__name__))
        x = N.dev(imgs)
        pr = self._predictor(x.shape[0], x.shape[3])
        pr.set_calibration(cameraMatrices[0], intrinsicMatrices[0], distortionCoefficients[0])
        return pr.hybridnet_forward(x, N.dev(centerHM, torch.int32), N.dev(center3D, torch.int32))
