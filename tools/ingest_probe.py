"""Where the time of the predict3D_frames ingest pipeline goes on this host: memcpy into pinned memory by N
threads, pinned host -> HBM bandwidth, and the driver with (a) numpy frame sets, (b) in-place fill callables
that write nothing (the pipeline without the host copy).  Run on the GPU box."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jarvis_hybridnet_amd import synthetic as S  # noqa: E402
from jarvis_hybridnet_amd.prediction._ingest import usable_cores  # noqa: E402

C, H, W, J, T, K = 12, 1024, 1280, 23, 32, 3
print("usable cores", usable_cores(), "affinity", len(os.sched_getaffinity(0)))
src = np.random.randint(0, 255, (8, C, H, W, 3), dtype=np.uint8)
for pinned in (False, True):
    dst = torch.empty((T, C, H, W, 3), dtype=torch.uint8, pin_memory=pinned).numpy()
    dst[:] = 1
    for nt in (1, 2, 4, 8, 12, 16):
        pool = ThreadPoolExecutor(nt)
        chunk = 4 << 20
        t0 = time.perf_counter()
        jobs = []
        for t in range(T):
            d, s = dst[t].reshape(-1), src[t % 8].reshape(-1)
            for o in range(0, d.size, chunk):
                jobs.append(pool.submit(np.copyto, d[o:o + chunk], s[o:o + chunk]))
        for j in jobs:
            j.result()
        dt = time.perf_counter() - t0
        print("memcpy pinned=%d threads=%2d: %.1f GB/s (%.1f ms per 32 frame sets)" %
              (pinned, nt, dst.nbytes / dt / 1e9, dt * 1e3))
        pool.shutdown()
host = torch.empty((T, C, H, W, 3), dtype=torch.uint8, pin_memory=True)
dev = torch.empty_like(host, device="cuda")
for _ in range(2):
    dev.copy_(host, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(host, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("H2D pinned: %.1f GB/s (%.1f ms per 32 frame sets)" % (host.numel() / dt / 1e9, dt * 1e3))

from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D  # noqa: E402
from jarvis_hybridnet_amd.prediction.predict3D import predict3D_frames  # noqa: E402
calib = S.ring_calibration(C, W, H, 1800.0)
cfg = NS(PARENT_DIR="/nonexistent", PROJECT_NAME="probe", DATASET=NS(DATASET_ROOT_DIR="x", MEAN=S.MEAN, STD=S.STD),
         CENTERDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=1, IMAGE_SIZE=256),
         KEYPOINTDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=J, BOUNDING_BOX_SIZE=256),
         HYBRIDNET=NS(NUM_CAMERAS=C, ROI_CUBE_SIZE=128, GRID_SPACING=2))
jp = JarvisPredictor3D(cfg, S.efficienttrack_weights("small", 1, 50), S.hybridnet_weights("small", J, 51))
cal = [t.cuda() for t in calib]
sets = [(S.blob_frames(calib, W, H, J, 52 + i)[0].permute(0, 2, 3, 1)[..., [2, 1, 0]] * 255).round().to(torch.uint8).numpy()
        for i in range(8)]
out = "/dev/shm/jh_probe"
spec = ((C, H, W, 3), torch.uint8)
for name, gen in (("numpy frame sets", lambda n: (sets[i % 8] for i in range(n))),
                  ("fill callables (no host copy)", lambda n: ((lambda dst: None) for i in range(n)))):
    predict3D_frames(jp, gen(T * K), *cal, cfg, out, time_batch=T, streams=K, frame_spec=spec)
    torch.cuda.synchronize()
    n = T * K * 8
    t0 = time.perf_counter()
    predict3D_frames(jp, gen(n), *cal, cfg, out, time_batch=T, streams=K, frame_spec=spec)
    dt = time.perf_counter() - t0
    st = next(iter(jp._ingest_cache.values())).stats
    print("predict3D_frames, %s: %.0f frames/s; per batch ms: %s" % (
        name, n / dt, {k: round(1e3 * v / st["batches"], 2) for k, v in st.items() if k != "batches"}))
