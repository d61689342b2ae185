"""Robustness check at the edges of the configuration space: frames much larger than the BASELINE configs (4K cameras,
time batch 8: > 2^31 bytes per time batch), many cameras, many joints, large grids -- through the HIP path against the
CPU oracle.  python tools/big_frame_check.py [W H C T [J=.. bbox=.. center=.. spacing=.. G=.. focal=.. size=..]]"""
import os
import sys
from types import SimpleNamespace as NS

sys.path.insert(0, os.getcwd())
import torch

from jarvis_hybridnet_amd import synthetic as S
from jarvis_hybridnet_amd.hybridnet.repro_layer import ReprojectionLayer
from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
from oracle import hybridnet_oracle as O

W, H, C, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (3840, 2160, 4, 8)))
J, bbox, center, spacing, G, focal, size = 4, 256, 256, 2, 32, 3200.0, "small"
for kv in sys.argv[5:]:                      # further key=value overrides: J= bbox= center= spacing= G= focal= size=
    k, val = kv.split("=")
    assert k in ("J", "bbox", "center", "spacing", "G", "focal", "size"), k
    globals()[k] = val if k == "size" else (float(val) if k == "focal" else int(val))
roi = G * spacing
torch.set_num_threads(16)
calib = S.ring_calibration(C, W, H, focal)
sd_c = S.efficienttrack_weights(size, 1, 5)
sd_h = S.hybridnet_weights(size, J, 6)
cfg = NS(PARENT_DIR="/nonexistent", PROJECT_NAME="big", DATASET=NS(DATASET_ROOT_DIR="x", MEAN=S.MEAN, STD=S.STD),
         CENTERDETECT=NS(MODEL_SIZE=size, NUM_JOINTS=1, IMAGE_SIZE=center),
         KEYPOINTDETECT=NS(MODEL_SIZE=size, NUM_JOINTS=J, BOUNDING_BOX_SIZE=bbox),
         HYBRIDNET=NS(NUM_CAMERAS=C, ROI_CUBE_SIZE=roi, GRID_SPACING=spacing))
pred = JarvisPredictor3D(cfg, sd_c, sd_h)
dev = [t.cuda() for t in calib]
frames = [S.blob_frames(calib, W, H, J, 40 + t)[0] for t in range(T)]
x = torch.stack(frames).cuda()
print("time batch of %d frame sets: %.2f GB fp32" % (T, x.numel() * 4 / 1e9), flush=True)
p, c, v = pred.forward_batch(x, *dev)
torch.cuda.synchronize()
dbg = {k: q.cpu() for k, q in pred.native(H, W, time_batch=T).debug("cuda").items()}
u8 = (torch.stack(frames).permute(0, 1, 3, 4, 2)[..., [2, 1, 0]] * 255).round().to(torch.uint8).cuda()
worst = 0.0
for t in range(T):
    inter = {}
    with torch.no_grad():
        rp, rc = O.predictor3d_forward(sd_c, sd_h, frames[t], *calib, center_size=center, bbox=bbox, roi_cube_size=roi,
                                       grid_spacing=spacing, mean=S.MEAN, std=S.STD, chunk=5, intermediates=inter)
    assert (rp is None) == (int(v[t]) == 0), "validity, frame %d" % t
    if rp is not None:
        # gather indices of the library against this host's oracle on the oracle's integer centres (torch's CPU
        # kernels flip a few indices against any other implementation, DESIGN.md section 1: such frames are counted)
        c3i, chm = inter["center3d"].int()[None], inter["center_hm"][None]
        idx = ReprojectionLayer(cfg).gather_indices(inter["heatmaps_padded"].cuda(), c3i.cuda(), chm.cuda(), dev[0][None],
                                                    dev[1][None], dev[2][None]).cpu()
        ridx = O.reprojection_indices(O.reprojection_grid(roi, spacing) + c3i[0], *calib, chm[0], bbox // 2 + 2, G)[0]
        flips = int((idx != ridx).sum())
        same_int = torch.equal(dbg["center3d_int"][t], c3i[0]) and torch.equal(dbg["center_hm"][t], chm[0])
        if flips or not same_int:
            print("frame %d: %d host index flips, integer centres equal: %s -- counted only" % (t, flips, same_int))
            continue
        # per joint, in units of the bar (1e-3 mm, scaled by 0.02 / confidence for nearly empty joints as in
        # tools/config_sweep.py)
        ej = (p[t].cpu() - rp[0]).abs().max(dim=-1)[0]
        bars = 1e-3 * torch.clamp(0.02 / rc[0].clamp_min(1e-6), min=1.0)
        e = float((ej / bars).max()) * 1e-3
        worst = max(worst, e)
        print("frame %d: per-joint error (mm) %s, confidences %s" % (
            t, ["%.1e" % float(q) for q in ej], [round(float(q), 3) for q in rc[0]]), flush=True)
# the single-frame call and the uint8 path on the last frame set
p1, c1 = pred(frames[-1].cuda(), *dev)
p8, c8 = pred.forward_uint8(u8[-1], *dev)
torch.cuda.synchronize()
print("single-frame call vs batch: %.3g mm; uint8 path vs fp32: %.3g mm" % (
    float((p1[0] - p[T - 1]).abs().max()), float((p8[0] - p[T - 1]).abs().max())))
print("big frame check: worst %.3g of the bar" % (worst / 1e-3))
sys.exit(0 if worst < 1e-3 else 1)
