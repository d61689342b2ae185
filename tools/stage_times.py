"""Per-stage kernel time of one time batch on one GPU (HIP events around every launch, one stream): the numbers
the multi-GPU scaling prediction of DESIGN.md section 5 is built from.   python tools/stage_times.py [cfg3|cfg5] [T]"""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from jarvis_hybridnet_amd import _native as N, synthetic as S
from jarvis_hybridnet_amd._predictor import NativePredictor
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
c = bench.CONFIGS[cfg]
T = int(sys.argv[2]) if len(sys.argv) > 2 else c["time_batch"]
calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
sd_c = S.efficienttrack_weights("small", 1, c["seeds"][0])
sd_h = S.hybridnet_weights("small", c["J"], c["seeds"][1])
pr = NativePredictor(sd_c, sd_h, num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                     roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"], mean=S.MEAN,
                     std=S.STD, time_batch=T)
pr.set_calibration(*[t.cuda() for t in calib])
base = torch.stack([S.blob_frames(calib, c["W"], c["H"], c["J"], c["seeds"][2] + i)[0] for i in range(4)]).cuda()
fr = base[torch.arange(T, device="cuda") % 4].contiguous()
for _ in range(2):
    pr.forward(fr)
torch.cuda.synchronize()
passes = 5
recs = []
for _ in range(passes):
    recs.append(N.profile(lambda: pr.forward(fr)))
n = len(recs[0])
names = [r[0] for r in recs[0]]
ms = [sorted(recs[p][i][1] for p in range(passes))[passes // 2] for i in range(n)]
# stage boundaries: center net ends at center_argmax; 2D keypoint stage ends before reproject_gather
i_arg = names.index("center_argmax")
i_rep = names.index("reproject_gather")
st = {"stage1_center (resize stem .. argmax)": sum(ms[:i_arg + 1]),
      "stage2_keypoints (triangulate, crop stem .. heatmaps)": sum(ms[i_arg + 1:i_rep]),
      "stage3_3d (reprojection, V2V, soft-argmax)": sum(ms[i_rep:])}
print("config %s, time batch %d, %d launches, %.2f ms of kernel time" % (cfg, T, n, sum(ms)))
for k, v in st.items():
    print("  %-58s %7.3f ms" % (k, v))
