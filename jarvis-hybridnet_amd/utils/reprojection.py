"""ReprojectionTool on MI355X (mirrors jarvis/utils/reprojection.py:16-90).

Calibration tensors use the reference's storage: cameraMatrices (C,4,3),
intrinsicMatrices (C,3,3) with the principal point in row 2,
distortionCoefficients (C,1,5).  Loading calibration files is outside the hot
path.
"""
import torch
import torch.nn as nn

from .. import _native as N


class ReprojectionTool(nn.Module):
    def __init__(self, root_dir=None, calib_paths=None, device="cuda"):
        super().__init__()
        if calib_paths is not None:
            raise NotImplementedError("calibration file loading is not part of the hot path; "
                                      "assign cameraMatrices / intrinsicMatrices / "
                                      "distortionCoefficients directly")
        self.device = device
        self.cameraMatrices = torch.tensor(0)
        self.intrinsicMatrices = torch.tensor(0)
        self.distortionCoefficients = torch.tensor(0)

    def reprojectPoint(self, point3D):
        """point3D (P,3) -> (C,2) for P == 1 (the reference squeezes), else (C,2,P)."""
        pts = N.dev(point3D)
        C = self.cameraMatrices.shape[0]
        uv = torch.empty((C, pts.shape[0], 2), device=pts.device, dtype=torch.float32)
        N.check(N.lib().jh_reproject_point(
            N.ptr(pts), pts.shape[0], C, N.ptr(N.dev(self.cameraMatrices)),
            N.ptr(N.dev(self.intrinsicMatrices)), N.ptr(N.dev(self.distortionCoefficients)),
            N.ptr(uv), N.stream()))
        return uv.permute(0, 2, 1).squeeze()

    def reconstructPoint(self, points, maxvals):
        """points (2,C) pixels, maxvals (C,1,1) weights -> (3,) mm."""
        pts = N.dev(points)
        C = pts.shape[1]
        out = torch.empty(3, device=pts.device, dtype=torch.float32)
        N.check(N.lib().jh_reconstruct_point(
            N.ptr(pts), N.ptr(N.dev(maxvals.reshape(-1))), C, N.ptr(N.dev(self.cameraMatrices)),
            N.ptr(N.dev(self.intrinsicMatrices)), N.ptr(N.dev(self.distortionCoefficients)),
            N.ptr(out), N.stream()))
        return out
