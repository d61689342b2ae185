"""ReprojectionLayer on MI355X (mirrors jarvis/hybridnet/repro_layer.py:12-119)."""
import torch
import torch.nn as nn

from .. import _native as N


class ReprojectionLayer(nn.Module):
    def __init__(self, cfg, num_cameras=None):
        super().__init__()
        self.cfg = cfg
        self.grid_spacing = cfg.HYBRIDNET.GRID_SPACING
        self.boxsize = cfg.HYBRIDNET.ROI_CUBE_SIZE
        self.grid_size = int(cfg.HYBRIDNET.ROI_CUBE_SIZE / cfg.HYBRIDNET.GRID_SPACING)
        self.num_cameras = num_cameras if num_cameras else cfg.HYBRIDNET.NUM_CAMERAS
        self.heatmap_size = int(cfg.KEYPOINTDETECT.BOUNDING_BOX_SIZE / 2 + 2)

    def _run(self, heatmaps, center, centerHM, cameraMatrices, intrinsicMatrices,
             distortionCoefficients, want_idx):
        hm = N.dev(heatmaps[0])                       # (C,J,hs,hs); batch index 0 only (:113-117)
        C, J, hs = hm.shape[0], hm.shape[1], hm.shape[2]
        G = self.grid_size
        vol = torch.empty((1, J, G, G, G), device=hm.device, dtype=torch.float32)
        idx = torch.empty((C, G, G, G), device=hm.device, dtype=torch.int32) if want_idx else None
        ws = N.workspace(N.lib().jh_reproject_workspace_bytes(C, J, hs, G), hm.device)
        N.check(N.lib().jh_reproject_forward(
            N.ptr(hm), C, J, hs, N.ptr(N.dev(center[0], torch.int32)),
            N.ptr(N.dev(centerHM[0], torch.int32)), N.ptr(N.dev(cameraMatrices[0])),
            N.ptr(N.dev(intrinsicMatrices[0])), N.ptr(N.dev(distortionCoefficients[0])), G,
            float(self.grid_spacing), N.ptr(vol), N.ptr(idx), ws.data_ptr(), ws.numel(),
            N.stream()))
        return vol, idx

    def forward(self, heatmaps, center, centerHM, cameraMatrices, intrinsicMatrices,
                distortionCoefficients):
        """heatmaps (1,C,J,hs,hs) padded; center (1,3) int; centerHM (1,C,2) int;
        calibration (1,C,...) -> heatmaps3D (1,J,G,G,G)."""
        return self._run(heatmaps, center, centerHM, cameraMatrices, intrinsicMatrices,
                         distortionCoefficients, False)[0]

    def gather_indices(self, heatmaps, center, centerHM, cameraMatrices, intrinsicMatrices,
                       distortionCoefficients):
        """The integer gather index of reprojectPoints (:40-85), (C,G,G,G) int64."""
        return self._run(heatmaps, center, centerHM, cameraMatrices, intrinsicMatrices,
                         distortionCoefficients, True)[1].long()
