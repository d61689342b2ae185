"""Scout for PREDICTOR_CASES that hit the reference's detection / clamp edge cases.

Runs only the centre stage of the CPU oracle (resize -> CenterDetect -> argmax -> reconstructPoint ->
reprojectPoint; prediction/jarvis3D.py:129-166 of the reference) over candidate parameters and prints, per
candidate, the per-camera heat-map maxima (in the units the `> 50` test sees), n_detect and which cameras
clamp.  Test infrastructure: used to choose the seeds in tests/cases.py, never by the product.

    python tools/edge_case_scout.py cfg2 --std 1.2 --fseeds 53,60,61 --focal 900,2400
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from jarvis_hybridnet_amd import synthetic as S  # noqa: E402
from oracle import hybridnet_oracle as O  # noqa: E402
from tests import cases  # noqa: E402


def centre_stage(c, std=None, fseed=None, focal=None, cseed=None):
    c = dict(c)
    if fseed is not None:
        c["fseed"] = fseed
    if focal is not None:
        c["focal"] = focal
    if cseed is not None:
        c["cseed"] = cseed
    size = c.get("size", "small")
    calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
    sd_c = S.efficienttrack_weights(size, 1, c["cseed"], deconv_std=std if std is not None else c.get("deconv_std", 1.2))
    imgs, _, _ = S.blob_frames(calib, c["W"], c["H"], c["J"], c["fseed"])
    cs, hw = c["center_size"], c["bbox"] // 2
    mean_t = torch.tensor(S.MEAN).view(3, 1, 1)
    std_t = torch.tensor(S.STD).view(3, 1, 1)
    small = (F.interpolate(imgs, size=[cs, cs], mode="bilinear", align_corners=False) - mean_t) / std_t
    with torch.no_grad():
        hm = O.efficienttrack_forward(sd_c, small, size, want_res1=False)[1]
    flat = hm.view(hm.shape[0], hm.shape[1], -1)
    m = flat.argmax(2).view(flat.shape[0], flat.shape[1], 1)
    preds = torch.cat((m % hm.shape[2], m // hm.shape[3]), dim=2)
    maxvals = flat.gather(2, m)
    n = torch.numel(maxvals[maxvals > 50])
    res = dict(maxvals=maxvals.flatten().tolist(), n_detect=n)
    if n >= 2:
        scale = torch.tensor([c["W"] / float(cs), c["H"] / float(cs)]).float()
        c3 = O.reconstruct_point((preds.reshape(c["C"], 2) * (scale * 2)).transpose(0, 1), maxvals / 255.,
                                 *calib)
        raw = O.reproject_point(c3.unsqueeze(0), *calib).int()
        res.update(center3d=c3.tolist(), raw=raw.tolist(),
                   x_lo=int((raw[:, 0] < hw).sum()), x_hi=int((raw[:, 0] > c["W"] - hw).sum()),
                   y_lo=int((raw[:, 1] < hw).sum()), y_hi=int((raw[:, 1] > c["H"] - hw).sum()))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--std", default="")
    ap.add_argument("--fseeds", default="")
    ap.add_argument("--cseeds", default="")
    ap.add_argument("--focal", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    c = cases.PREDICTOR_CASES[a.tag]
    for std in [float(v) for v in a.std.split(",") if v] or [None]:
        for fs in [int(v) for v in a.fseeds.split(",") if v] or [None]:
            for cs in [int(v) for v in a.cseeds.split(",") if v] or [None]:
                for fo in [float(v) for v in a.focal.split(",") if v] or [None]:
                    r = centre_stage(c, std, fs, fo, cs)
                    mv = " ".join("%.1f" % v for v in r.pop("maxvals"))
                    print("std", std, "fseed", fs, "cseed", cs, "focal", fo, "| maxvals", mv, "|", r, flush=True)
