"""Single-frame-set latency probe (the reference driver's call pattern, T = 1) at BASELINE
configs[2]: wall time per forward (graph replay and plain launches) and, from one profiled
pass, the per-launch HIP-event durations in launch order -- what the 150 launches of one
forward cost on the GPU when each covers a single frame set.

    python3 tools/latency_probe.py [T [model size]]        (also the command for rocprofv3 --kernel-trace)
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from jarvis_hybridnet_amd import _native as N  # noqa: E402
from jarvis_hybridnet_amd import synthetic as S  # noqa: E402
from jarvis_hybridnet_amd._predictor import NativePredictor  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1
SIZE = sys.argv[2] if len(sys.argv) > 2 else "small"
C, W, H, J = 12, 1280, 1024, 23
calib = S.ring_calibration(C, W, H, 1800.0)
sd_c, sd_h = S.efficienttrack_weights(SIZE, 1, 50), S.hybridnet_weights(SIZE, J, 51)
fr = torch.stack([S.blob_frames(calib, W, H, J, 52 + i)[0] for i in range(T)]).cuda()
p = NativePredictor(sd_c, sd_h, num_cameras=C, num_joints=J, center_size=256, bbox=256, roi_cube_size=128,
                    grid_spacing=2, img_h=H, img_w=W, mean=S.MEAN, std=S.STD, time_batch=T, center_model=SIZE, kp_model=SIZE)
p.set_calibration(*[t.cuda() for t in calib])
out = None
for mode in ((True, False) if p.Cloc == p.C else (False,)):
    p.graph_replay = mode
    for _ in range(5):
        out = p.forward(fr, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        p.forward(fr, out)
    torch.cuda.synchronize()
    print("T=%d %s: %.3f ms per forward" % (T, "graph replay" if mode else "plain launches",
                                            1e3 * (time.perf_counter() - t0) / 50))
N.profile(lambda: p.forward(fr, out))
recs = N.profile(lambda: p.forward(fr, out))
tot = sum(r[1] for r in recs)
print("T=%d sum of %d per-launch HIP-event durations: %.3f ms" % (T, len(recs), tot))
agg = {}
for name, ms, fl, by in recs:
    a = agg.setdefault(name, [0.0, 0])
    a[0] += ms
    a[1] += 1
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %-34s n=%2d total %.3f ms  avg %.1f us" % (k, n, ms, 1e3 * ms / n))
