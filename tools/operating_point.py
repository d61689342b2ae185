#!/usr/bin/env python3
"""Operating-point sweep of the single-GPU hot path: frames/s for time batch T x HIP streams K.

    python tools/operating_point.py [--config cfg3] [--model-size small] [--T 4,8,12,16,32] [--K 1,2,3,4,6,8]
                                    [--out profiles/r05_operating_point.json]

Same workload, weights and frame sets as bench.py's headline (BASELINE configs[2], frames resident in HBM as
fp32); one cell = K MultiStreamPredictor streams x T frame sets each, timed like bench.py's step loop (wall clock
between two device synchronisations over enough steps to cover >= `--seconds`).  Writes one JSON table.  (HBM
traffic is collected separately under rocprofv3 --pmc by tools/pmc_traffic.sh, at the benched time batch on one stream.)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--model-size", default="small")
    ap.add_argument("--T", default="4,8,12,16,32")
    ap.add_argument("--K", default="1,2,3,4,6,8")
    ap.add_argument("--seconds", type=float, default=0.6)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import MultiStreamPredictor, NativePredictor
    c = bench.CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
    calib_dev = [t.to(dev) for t in calib]
    sd_c = S.efficienttrack_weights(a.model_size, 1, c["seeds"][0])
    sd_h = S.hybridnet_weights(a.model_size, c["J"], c["seeds"][1])
    Ts = [int(v) for v in a.T.split(",")]
    Ks = [int(v) for v in a.K.split(",")]
    nd = max(Ts)
    distinct = torch.stack([S.blob_frames(calib, c["W"], c["H"], c["J"], c["seeds"][2] + i)[0]
                            for i in range(min(nd, 32))]).to(dev)
    cells = []
    for T in Ts:
        fr = distinct[torch.arange(T, device=dev) % distinct.shape[0]].contiguous()
        kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                  roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"], mean=S.MEAN,
                  std=S.STD, time_batch=T, center_model=a.model_size, kp_model=a.model_size)
        for K in Ks:
            try:
                msp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, **kw), streams=K)
                msp.set_calibration(*calib_dev)
                outs = [(torch.empty((T, c["J"], 3), device=dev), torch.empty((T, c["J"]), device=dev),
                         torch.empty((T,), device=dev, dtype=torch.int32)) for _ in range(K)]
                for _ in range(2):
                    for i in range(K):
                        msp.forward(fr, outs[i])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    msp.forward(fr, outs[i])
                torch.cuda.synchronize()
                one = time.perf_counter() - t0
                steps = max(3, int(a.seconds / max(one, 1e-4)))
                t0 = time.perf_counter()
                for _ in range(steps):
                    for i in range(K):
                        msp.forward(fr, outs[i])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                cell = dict(time_batch=T, streams=K, frames_per_s=T * K * steps / dt, ms_per_step=1e3 * dt / steps,
                            steps=steps, launches_per_time_batch=int(msp.preds[0].launches),
                            device_mb=round(sum(p.device_bytes for p in msp.preds) / 2 ** 20))
                del msp, outs
            except Exception as e:          # noqa: BLE001 -- a cell that does not fit says so
                cell = dict(time_batch=T, streams=K, error=repr(e)[:200])
            torch.cuda.empty_cache()
            cells.append(cell)
            print(json.dumps(cell), file=sys.stderr, flush=True)
    best = max((x for x in cells if "frames_per_s" in x), key=lambda x: x["frames_per_s"])
    table = dict(workload=c["workload"].replace("small/small", "%s/%s" % (a.model_size, a.model_size)),
                 dtype="f32", method="wall clock between device synchronisations, frames resident in HBM, "
                 ">= %.1f s per cell after 2 warm-up steps" % a.seconds, best=best, cells=cells)
    text = json.dumps(table, indent=1)
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w") as f:
            f.write(text + "\n")
    print(json.dumps(dict(best=best)))


if __name__ == "__main__":
    main()
