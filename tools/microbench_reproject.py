"""Micro-benchmark: HybridNet 3D stage pieces (reprojection gather) at cfg3 via the
predictor's hybridnet_forward, with per-launch HIP-event timing."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from jarvis_hybridnet_amd import _native as N, synthetic as S
from jarvis_hybridnet_amd._predictor import NativePredictor
T = int(os.environ.get("T", "8")); C, J = 12, 23
calib = S.ring_calibration(C, 1280, 1024, 1800.0)
pr = NativePredictor(None, S.hybridnet_weights("small", J, 51), num_cameras=C, num_joints=J,
                     center_size=256, bbox=256, roi_cube_size=128, grid_spacing=2, img_h=256,
                     img_w=256, mean=[0, 0, 0], std=[1, 1, 1], time_batch=T)
pr.set_calibration(*[t.cuda() for t in calib])
crops = torch.randn(T, C, 3, 256, 256, device="cuda")
chm = torch.tensor(S.project(torch.zeros(1, 3).double().numpy(), *calib)[:, 0]).int().clamp(128, 896)
chm = chm[None].repeat(T, 1, 1).cuda().contiguous()
c3 = torch.zeros(T, 3, dtype=torch.int32, device="cuda")
for _ in range(2):
    pr.hybridnet_forward(crops, chm, c3, False, False)
torch.cuda.synchronize()
recs = []
for _ in range(3):
    recs += N.profile(lambda: pr.hybridnet_forward(crops, chm, c3, False, False))
agg = {}
for name, ms, fl, by in recs:
    a = agg.setdefault(name, [0.0, 0, by]); a[0] += ms; a[1] += 1
for k in ("reproject_gather", "softargmax"):
    ms, n, by = agg[k]
    print("%-18s avg %.3f ms per %d frames  %.0f GB/s algorithmic" % (k, ms / n, T, by / (ms / n) / 1e6))
