"""Every launch of one time batch, in launch order, with its median HIP-event duration (one stream, kernels timed
alone) -- the per-launch view behind bench.py's per-kernel-name table.

    python tools/launch_table.py [--config cfg3] [--model-size small] [-T 32] [--out file.tsv]
"""
import argparse
import os
import sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
from jarvis_hybridnet_amd import _native as N, synthetic as S  # noqa: E402
from jarvis_hybridnet_amd._predictor import NativePredictor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg3")
ap.add_argument("--model-size", default=None)
ap.add_argument("-T", type=int, default=None)
ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--out", default=None)
ap.add_argument("--precision", default=None)
a = ap.parse_args()
c = bench.CONFIGS[a.config]
a.model_size = a.model_size or c.get("size", "small")
T = a.T or c["time_batch"]
calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
sd_c = S.efficienttrack_weights(a.model_size, 1, c["seeds"][0])
sd_h = S.hybridnet_weights(a.model_size, c["J"], c["seeds"][1])
pr = NativePredictor(sd_c, sd_h, num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                     roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"], mean=S.MEAN,
                     std=S.STD, time_batch=T, center_model=a.model_size, kp_model=a.model_size, precision=a.precision)
pr.set_calibration(*[t.cuda() for t in calib])
nd = min(T, 8)
base = torch.stack([S.blob_frames(calib, c["W"], c["H"], c["J"], c["seeds"][2] + i)[0] for i in range(nd)]).cuda()
fr = base[torch.arange(T, device="cuda") % nd].contiguous()
for _ in range(2):
    pr.forward(fr)
torch.cuda.synchronize()
recs = [N.profile(lambda: pr.forward(fr)) for _ in range(a.passes)]
n = len(recs[0])
rows = []
for i in range(n):
    name, _, fl, by = recs[0][i]
    ms = sorted(recs[p][i][1] for p in range(a.passes))[a.passes // 2]
    pr = bench.price(name, fl, by, ms * 1e-3)        # the binding roofline of the launch (bench.py)
    rows.append((i, name, ms, pr["bound"], pr["frac"], fl, by))
total = sum(r[2] for r in rows)
lines = ["# %s %s T=%d: %d launches, %.3f ms of kernel time" % (a.config, a.model_size, T, n, total),
         "idx\tkernel\tms\tbound\tfrac\tcum_ms\talg_gflop\talg_mb"]
cum = 0.0
for i, name, ms, fam, frac, fl, by in rows:
    cum += ms
    lines.append("%d\t%s\t%.4f\t%s\t%.3f\t%.3f\t%.2f\t%.1f" % (i, name, ms, fam, frac, cum, fl / 1e9, by / 1e6))
text = "\n".join(lines)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write(text + "\n")
print(text)
