#!/usr/bin/env python3
"""Benchmark of the multi-view inference hot path on MI355X.

Metric (BASELINE.json): multi-view frames/s.  One "step" = one pass of the hot
path (JarvisPredictor3D.forward: resize -> CenterDetect -> argmax ->
triangulation -> crops -> KeypointDetect -> reprojection -> V2V -> soft-argmax)
over one time batch of T independent synthetic multi-view frames that are
already resident in HBM as fp32 (C,3,H,W) tensors (the API's input type).
Workload = BASELINE.json configs[2]: 12 cameras 1280x1024, 23 keypoints, 64^3
voxel grid, small/small models, fp32 (the reference's precision).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL); cameras are
sharded over the ranks (jarvis_hybridnet_amd/distributed.py).  Rank 0 prints ONE
JSON line; `roofline` is measured live with HIP events around every launch of a
profiled pass, `cpu_baseline` times the CPU oracle on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    "cfg3": dict(C=12, W=1280, H=1024, J=23, roi=128, spacing=2, bbox=256, center=256, focal=1800.0),
    "cfg2": dict(C=4, W=640, H=512, J=23, roi=96, spacing=2, bbox=256, center=256, focal=900.0),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup
    CPU quota (the GPU boxes expose 256 logical CPUs but a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--time-batch", type=int, default=None,
                    help="multi-view frames per time batch (default: 32 per stream on one GPU, 64 per "
                         "GPU when camera-sharded)")
    ap.add_argument("--streams", type=int, default=3,
                    help="single GPU: independent time batches in flight on that many HIP streams; one "
                         "step = one time batch per stream")
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-uint8", action="store_true", help="skip the uint8-ingest side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph")
    ap.add_argument("--sharded-streams", type=int, default=2,
                    help="multi-GPU: camera-sharded pipelines in flight per rank (own plans and HIP stream each, "
                         "one RCCL communicator: the collectives are issued in the same order on every rank)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="multi-GPU: do not overlap the heatmap exchange with the next step's 2D stage")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU (camera-sharded, RCCL) code path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" %
                         (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ["NCCL_DEBUG"] = "WARN"          # keep RCCL's banner and warnings off stdout:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/jh_rccl_%h_%p.log")   # the JSON line is last
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))

    from jarvis_hybridnet_amd import _native as N
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor

    c = CONFIGS[args.config]
    if args.time_batch is None:
        args.time_batch = 32
    KS = max(1, args.sharded_streams) if (sharded and not args.no_pipeline) else 1
    K = KS if sharded else (1 if args.graph else max(1, args.streams))
    # Weak scaling: every rank always does the work of `--time-batch` whole frames
    # (T*C images of 2D work, T frames of 3D work).  Ranks form groups of `gs` GPUs
    # that shard the cameras of T*gs frames; world/gs groups run side by side.
    from jarvis_hybridnet_amd.distributed import plan_groups
    gs, n_groups = plan_groups(world, c["C"])
    grank, gidx = rank % gs, rank // gs
    T = args.time_batch * gs                       # frames per group and step
    calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
    sd_c = S.efficienttrack_weights("small", 1, 50)
    sd_h = S.hybridnet_weights("small", c["J"], 51)
    distinct = [S.blob_frames(calib, c["W"], c["H"], c["J"], 52 + i)[0] for i in range(min(T, 2))]

    def device_frames(cam_lo, cam_n):
        """(T, cam_n, 3, H, W) on the GPU, assembled there from the distinct frames so
        the host never holds T copies (T = 64 frames per group at 8 GPUs)."""
        base = torch.stack([d[cam_lo:cam_lo + cam_n] for d in distinct]).to(dev)
        idx = torch.arange(T, device=dev) % len(distinct)
        return base[idx].contiguous()

    common = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                  roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
                  mean=S.MEAN, std=S.STD, time_batch=T)
    if not sharded:
        from jarvis_hybridnet_amd._predictor import MultiStreamPredictor
        msp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, **common), streams=K)
        msp.set_calibration(*[t.to(dev) for t in calib])
        pred = msp.preds[0]                        # the per-kernel profile and the side legs use one
        fr = device_frames(0, c["C"])
        outs = [(torch.empty((T, c["J"], 3), device=dev), torch.empty((T, c["J"]), device=dev),
                 torch.empty((T,), device=dev, dtype=torch.int32)) for _ in range(K)]
        out = outs[0]
        torch.cuda.synchronize()

        def step():
            # one step = one time batch per stream (K * T frame sets)
            res0 = None
            for i in range(K):
                r = msp.forward(fr, outs[i]) if K > 1 else pred.forward(fr, outs[i])
                res0 = r if i == 0 else res0
            return res0
    else:
        from jarvis_hybridnet_amd.distributed import ShardedPredictor, camera_range
        groups = [dist.new_group(list(range(g * gs, (g + 1) * gs))) for g in range(n_groups)]
        cam_lo, cam_n = camera_range(c["C"], grank, gs)
        fr = device_frames(cam_lo, cam_n)
        preds, shs = [], []
        for _ in range(KS):
            p_ = NativePredictor(sd_c, sd_h, time_batch_3d=T // gs, cam_lo=cam_lo, cam_n=cam_n, **common)
            p_.set_calibration(*[t.to(dev) for t in calib])
            preds.append(p_)
            shs.append(ShardedPredictor(p_, num_cameras=c["C"], num_joints=c["J"], time_batch=T,
                                        heat_shape=(p_.Hh, p_.Hh, p_.Jp), rank=grank, world=gs,
                                        device=dev, exchange=args.exchange, group=groups[gidx]))
        pred, sh = preds[0], shs[0]
        sh_streams = [torch.cuda.Stream() for _ in range(KS)]
        torch.cuda.synchronize()

        def step():
            return sh.step(fr)

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        res = step()
    run = step
    if args.graph and not sharded:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        run = g.replay
        run()
    if sharded and not args.no_pipeline:
        # consecutive time batches pipelined: the heatmap exchange of step i runs under the
        # CenterDetect stage of step i+1; K submits + the final flush = exactly K whole steps.
        # KS such pipelines run on KS HIP streams (every rank issues their collectives in the
        # same order on the one communicator of its group).
        for k in range(1, KS):                     # warm the other pipelines too
            with torch.cuda.stream(sh_streams[k]):
                shs[k].step(fr)

        def run():
            for k in range(KS):
                with torch.cuda.stream(sh_streams[k]):
                    shs[k].submit(fr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    if sharded and not args.no_pipeline:
        for k in range(KS):
            with torch.cuda.stream(sh_streams[k]):
                shs[k].flush()
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    res = step()
    torch.cuda.synchronize()
    res = [r.clone() if torch.is_tensor(r) else r for r in res]      # `out` is reused below
    valid = int(res[2].sum().item())
    fps = T * K * n_groups * args.steps / dt

    line = {
        "metric": "multi-view frames/s (12cam 1280x1024, 23kpt, 64^3 grid)",
        "value": fps, "unit": "multi-view frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded blob frames, ring calibration, random-init weights)",
        "config": {"workload": "BASELINE configs[2]: HybridNet 12-camera 1280x1024, 23 kpts, "
                               "64^3 grid, small/small" if args.config == "cfg3" else args.config,
                   "cameras": c["C"], "frame": [c["H"], c["W"]], "joints": c["J"],
                   "grid": int(c["roi"] / c["spacing"]), "time_batch": T, "streams": K,
                   "frames_per_step": T * K * n_groups, "valid_frames_last_step": valid,
                   "parallelism": ("single GPU, %d time batches in flight on %d HIP streams" % (K, K))
                   if not sharded else
                   "%d group(s) x %d GPUs: camera-sharded 2D (%d cams/GPU) + RCCL %s of "
                   "heatmaps%s + frame-sharded 3D; %d such pipeline(s) per rank" % (
                       n_groups, gs, c["C"] // gs, args.exchange,
                       "" if args.no_pipeline else " (overlapped with the next step's CenterDetect)", KS),
                   "launches_per_step": int(pred.launches) * K, "launches_per_time_batch": int(pred.launches),
                   "hipgraph": bool(args.graph)},
    }

    if sharded and rank != 0:
        for _ in range(3):                         # rank 0 profiles 3 steps: keep the collectives matched
            step()
    if rank == 0:
        # ---- roofline of the dominant kernel, HIP events around every launch (at N > 1: rank 0's
        # share of the camera-sharded step)
        recs = []
        for _ in range(3):
            recs += N.profile(step if sharded else (lambda: pred.forward(fr, out)))   # one stream: kernels timed alone
        agg = {}
        for name, ms, fl, by in recs:
            a = agg.setdefault(name, [0.0, 0, fl, by])
            a[0] += ms
            a[1] += 1
        total_ms = sum(a[0] for a in agg.values())
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])
        name, (ms, cnt, fl, by) = top[0]
        avg_s = ms / cnt * 1e-3
        if fl > 0 and name.startswith("conv"):
            line["roofline"] = {"kernel": name, "bound": "mfma", "achieved": fl / avg_s / 1e12,
                                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": fl / avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                                "avg_launch_ms": ms / cnt, "launches_per_step": cnt // 3,
                                "share_of_step": ms / total_ms}
        else:
            line["roofline"] = {"kernel": name, "bound": "hbm", "achieved": by / avg_s / 1e9,
                                "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": by / avg_s / 1e9 / PEAK_HBM_GBS, "traffic": None,
                                "avg_launch_ms": ms / cnt, "launches_per_step": cnt // 3,
                                "share_of_step": ms / total_ms}
        if "wino" in name:
            # Winograd F(2x2,3x3) over (y,x) x direct z executes 12 instead of 27 multiplies per
            # output and channel pair; `achieved` above is the ALGORITHMIC (direct-convolution)
            # rate, so it may exceed the MFMA peak.  The matrix cores' own utilisation:
            cin, cout = [int(v) for v in name.split("_")[2].split("@")[0].split("x")]
            pad = ((cin + 7) // 8 * 8) * ((cout + 15) // 16 * 16) / float(cin * cout)
            ex = line["roofline"]["achieved"] * 12.0 / 27.0 * pad
            line["roofline"]["algorithm"] = ("Winograd F(2x2,3x3) x 3 z taps: 2.25x fewer multiplies than "
                                             "the direct algorithm `achieved` is counted in")
            line["roofline"]["mfma_executed"] = {"achieved": ex, "unit": "TFLOP/s",
                                                 "frac": ex / PEAK_F32_MFMA_TFLOPS}
        # HBM bytes of that kernel from the committed PMC passes (rocprofv3 cannot run
        # inside this process), scaled to this run's time batch
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if name in pmc:
                line["roofline"]["traffic"] = pmc[name]["hbm_bytes_per_launch"] * (T // gs if sharded else T) / pmc[name]["time_batch"]
                line["roofline"]["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
                line["roofline"]["algorithmic_bytes"] = by
        except (OSError, ValueError):
            pass
        line["kernel_breakdown_ms_per_step"] = {k: round(v[0] / 3, 4) for k, v in top[:int(os.environ.get("JH_BENCH_TOP", "12"))]}
        line["kernel_time_ms_per_step"] = total_ms / 3

    if rank == 0 and not sharded and not args.no_uint8:
        # ---- the same step fed uint8 BGR frames as the decoder delivers
        # them, (a) resident in HBM, (b) copied from pinned host memory inside the timed
        # region (PCIe-inclusive rate; never the headline `value`)
        u8 = torch.stack([(d.permute(0, 2, 3, 1)[..., [2, 1, 0]] * 255).round().to(torch.uint8)
                          for d in distinct])
        host = u8[torch.arange(T) % len(distinct)].contiguous().pin_memory()
        dev_u8 = host.to(dev)
        for _ in range(2):
            pred.forward(dev_u8, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pred.forward(dev_u8, out)
        torch.cuda.synchronize()
        fps_res = T * 10 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(10):
            dev_u8.copy_(host, non_blocking=True)
            pred.forward(dev_u8, out)
        torch.cuda.synchronize()
        fps_pcie = T * 10 / (time.perf_counter() - t0)
        # double-buffered: the copy of batch i+1 runs on its own stream under batch i
        bufs = [dev_u8, torch.empty_like(dev_u8)]
        copy_s, comp_s = torch.cuda.Stream(), torch.cuda.current_stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]
        for e in freed:
            e.record(comp_s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 12
        for i in range(nb + 1):
            if i < nb:
                with torch.cuda.stream(copy_s):
                    copy_s.wait_event(freed[i % 2])
                    bufs[i % 2].copy_(host, non_blocking=True)
                    ready[i % 2].record(copy_s)
            if i > 0:
                j = (i - 1) % 2
                comp_s.wait_event(ready[j])
                pred.forward(bufs[j], out)
                freed[j].record(comp_s)
        torch.cuda.synchronize()
        fps_overlap = T * nb / (time.perf_counter() - t0)
        line["uint8_ingest"] = {"frames_per_s_resident": fps_res,
                                "frames_per_s_incl_pcie_h2d_serial": fps_pcie,
                                "frames_per_s_incl_pcie_h2d_overlapped": fps_overlap,
                                "bytes_per_frame": int(host[0].numel()),
                                "note": "pinned host uint8 BGR -> HBM inside the timed region; "
                                        "overlapped = copy of batch i+1 on a second HIP stream"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (port of the reference) on the host cores,
        # same workload, bounded sample of whole frames
        from oracle import hybridnet_oracle as O
        cores = usable_cores()
        torch.set_num_threads(cores)
        kw = dict(center_size=c["center"], bbox=c["bbox"], roi_cube_size=c["roi"],
                  grid_spacing=c["spacing"], mean=S.MEAN, std=S.STD)
        with torch.no_grad():
            ref = O.predictor3d_forward(sd_c, sd_h, distinct[0], *calib, **kw)     # warm-up
            n, t0 = 0, time.perf_counter()
            while n < 1 or (time.perf_counter() - t0 < args.cpu_seconds and n < 50):
                O.predictor3d_forward(sd_c, sd_h, distinct[n % len(distinct)], *calib, **kw)
                n += 1
            cpu_dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": n / cpu_dt, "unit": "multi-view frames/s", "cores": cores,
                                "kind": "port",
                                "sample": "%d whole frames of the same workload through the CPU "
                                          "oracle (torch %s, %d threads)" % (n, torch.__version__, cores)}
        if ref[0] is not None:
            # NOTE: torch's CPU kernels differ in the last bit between CPU models, which
            # flips a few reprojection gather indices of the reference itself (DESIGN.md,
            # "reproducibility of the reference"); the pinned comparison is the next one.
            line["parity_max_abs_mm_vs_host_oracle"] = (res[0][0].cpu() - ref[0][0]).abs().max().item()
    if rank == 0 and world == 1 and args.config == "cfg3":
        # frame 0 of this workload is fixture case `cfg3` of tests/golden/predictor.npz,
        # i.e. the output of the imported upstream reference on the same input
        import numpy as np
        gpath = os.path.join(ROOT, "tests", "golden", "predictor.npz")
        if os.path.isfile(gpath):
            gold = np.load(gpath)["cfg3.points3D"]
            line["parity_max_abs_mm_vs_reference_fixture"] = float(
                np.abs(res[0][0].cpu().numpy() - gold[0]).max())

    if sharded and (world > 1 or os.environ.get("JH_BENCH_REPLICAS")):
        # next to the camera-sharded number, the frame-parallel upper bound --
        # every rank runs the whole path on its own `--time-batch` frames, no data-path
        # collective (what a throughput-only deployment would do)
        try:
            del sh, pred, fr, shs, preds
            torch.cuda.empty_cache()
            from jarvis_hybridnet_amd._predictor import MultiStreamPredictor
            Tb, Kr = 32, max(1, args.streams)
            rp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, **dict(common, time_batch=Tb)),
                                      streams=Kr)
            rp.set_calibration(*[t.to(dev) for t in calib])
            T_keep, T = T, Tb
            rfr = device_frames(0, c["C"])
            T = T_keep
            routs = [(torch.empty((Tb, c["J"], 3), device=dev), torch.empty((Tb, c["J"]), device=dev),
                      torch.empty((Tb,), device=dev, dtype=torch.int32)) for _ in range(Kr)]
            torch.cuda.synchronize()
            for _ in range(max(1, args.warmup)):
                for i in range(Kr):
                    rp.forward(rfr, routs[i])
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for i in range(Kr):
                    rp.forward(rfr, routs[i])
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            line["replicas_only"] = {"value": Tb * Kr * world * args.steps / tt.item(),
                                     "unit": "multi-view frames/s",
                                     "note": "frame-parallel upper bound: each rank runs all %d cameras "
                                             "of its own frames (%d streams x %d frames per step), no "
                                             "collective" % (c["C"], Kr, Tb)}
        except Exception as e:                      # never lose the headline number to the extra
            line["replicas_only"] = {"error": repr(e)[:200]}
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)       # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
