#!/usr/bin/env python3
"""Benchmark of the multi-view inference hot path on MI355X.

Metric (BASELINE.json): multi-view frames/s.  One "step" = one pass of the hot
path (JarvisPredictor3D.forward: resize -> CenterDetect -> argmax ->
triangulation -> crops -> KeypointDetect -> reprojection -> V2V -> soft-argmax)
over one time batch of T independent synthetic multi-view frames per HIP stream,
frames already resident in HBM as fp32 (C,3,H,W) tensors (the API's input type).
Default workload = BASELINE.json configs[2]: 12 cameras 1280x1024, 23 keypoints,
64^3 voxel grid, small/small models, fp32 (the reference's precision).
`--config cfg5` = configs[4] (16 cameras, 30 keypoints, 96^3), `--config cfg2` =
configs[1]'s geometry in fp32.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL, cameras sharded over the ranks
(jarvis_hybridnet_amd/distributed.py): either under torch.distributed.run (the driver's
launch line; RANK / WORLD_SIZE come from the environment) or, when WORLD_SIZE is unset, this
script starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself as a
CHILD process -- before anything here touches the GPU -- and relays rank 0's line and the
children's exit code.  Rank 0 prints ONE JSON line.  `roofline` and `kernels` are measured live with HIP events around every
launch of profiled passes on one stream; `cpu_baseline` times the CPU oracle on the
host cores (N = 1 only).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONFIGS = {
    # seeds = (CenterDetect weights, HybridNet weights, first frame set): those of the fixture case of
    # the same name in tests/cases.py, so frame 0 of every workload has the reference's own output on file
    "cfg3": dict(C=12, W=1280, H=1024, J=23, roi=128, spacing=2, bbox=256, center=256, focal=1800.0,
                 time_batch=32, seeds=(50, 51, 52),
                 metric="multi-view frames/s (12cam 1280x1024, 23kpt, 64^3 grid)",
                 workload="BASELINE configs[2]: HybridNet 12-camera 1280x1024, 23 kpts, 64^3 grid, "
                          "small/small"),
    "cfg2": dict(C=4, W=640, H=512, J=23, roi=96, spacing=2, bbox=256, center=256, focal=900.0,
                 time_batch=32, seeds=(50, 51, 53),
                 metric="multi-view frames/s (4cam 640x512, 23kpt, 48^3 grid)",
                 workload="BASELINE configs[1] geometry in fp32: HybridNet 4-camera 640x512, 23 kpts, "
                          "48^3 grid, small/small"),
    "cfg5": dict(C=16, W=1280, H=1024, J=30, roi=192, spacing=2, bbox=256, center=256, focal=1800.0,
                 time_batch=8, seeds=(54, 55, 56),
                 metric="multi-view frames/s (16cam 1280x1024, 30kpt, 96^3 grid)",
                 workload="BASELINE configs[4]: HybridNet 16-camera 1280x1024, 30 kpts, 96^3 grid, "
                          "small/small, batched multi-subject stream"),
    # the one configuration the reference ships (projects/Example_Project/config.yaml:36-37: ROI_CUBE_SIZE 144,
    # GRID_SPACING 2 => a 72^3 grid); not a BASELINE config, reported under profiles/ as a secondary workload
    "ex72": dict(C=12, W=1280, H=1024, J=23, roi=144, spacing=2, bbox=256, center=256, focal=1800.0,
                 time_batch=24, seeds=(50, 51, 64),
                 metric="multi-view frames/s (12cam 1280x1024, 23kpt, 72^3 grid)",
                 workload="reference Example_Project geometry: HybridNet 12-camera 1280x1024, 23 kpts, "
                          "72^3 grid (ROI 144 / spacing 2), small/small"),
    # the reference's DEFAULT configuration (jarvis/config/config.py:36-37,49-51: medium / medium models,
    # CENTERDETECT.IMAGE_SIZE 320, KEYPOINTDETECT.BOUNDING_BOX_SIZE 320) on the shipped rig and 72^3 grid; frame 0 =
    # fixture case `default_medium_320` of tests/cases.py
    "def320": dict(C=12, W=1280, H=1024, J=23, roi=144, spacing=2, bbox=320, center=320, focal=1800.0,
                   time_batch=16, seeds=(66, 63, 52), size="medium", fixture="default_medium_320",
                   metric="multi-view frames/s (12cam 1280x1024, 23kpt, 72^3 grid, 320 px crops)",
                   workload="reference default configuration: HybridNet 12-camera 1280x1024, 23 kpts, 72^3 grid, "
                            "CenterDetect 320x320, crops 320x320, medium/medium"),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_bf16 / 32x32x16, dense
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC = os.path.join("profiles", "r06_pmc_traffic.json")


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


def kernel_family(name):
    """Which rooflines can bound a profiled launch, by kernel FAMILY (not by name prefix): the MFMA convolutions
    (`conv_mfma_kernel`, the Winograd / bf16x3 / fused ConvTranspose kernels -- all named conv{2,3}d_*) are priced
    against the matrix-core peak of their arithmetic type AND against HBM, whichever floor is higher (kernel_table);
    everything else -- the vector-ALU stem (`stem_conv_*`:
    HBM-bound on the frame rows it fetches), the one-channel CenterDetect head (`deconv_*_c1`), fused BiFPN nodes,
    InstanceNorm passes, depthwise convolutions, the reprojection gather, argmax / soft-argmax -- against HBM."""
    return "mfma" if name.startswith(("conv2d_", "conv3d_")) else "hbm"


def executed_flops(name, flops):
    """FLOPs the matrix cores really execute for a conv launch whose ALGORITHMIC
    (direct-convolution, unpadded) count is `flops`: channel padding (input channels to 8,
    output channels to 16) and, for the Winograd kernel, 12 instead of 27 multiplies per
    output and channel pair (F(2x2,3x3) over (y,x) x 3 z taps)."""
    try:
        cin, cout = [int(v) for v in name.split("_")[-1].split("@")[0].split("x")]
    except ValueError:
        return flops
    pad = (-(-cin // 8) * 8) * (-(-cout // 16) * 16) / float(cin * cout)
    if "bf16x3" in name:            # three bf16 MFMAs per product (hi hi + lo hi + hi lo), channels padded to 16
        return flops * 3.0 * (-(-cin // 16) * 16) * (-(-cout // 16) * 16) / float(cin * cout)
    return flops * pad * (12.0 / 27.0 if "wino" in name else 1.0)


def price(name, flops, nbytes, s):
    """Roofline row of one launch (or of all launches of one kernel name): a convolution is bound by whichever floor is
    HIGHER -- the matrix cores (executed FLOPs / peak) or HBM (algorithmic bytes: one read of the input, one write of the
    output, the weights / 8 TB/s); the few-channel 1x1 and stem-side layers are HBM-bound, pricing them against the matrix
    peak hid their head-room.  Both fractions are kept; `frac` is the binding one.  Everything else: HBM."""
    if flops > 0 and kernel_family(name) == "mfma":
        ex = executed_flops(name, flops)
        peak = PEAK_BF16_MFMA_TFLOPS if "bf16x3" in name else PEAK_F32_MFMA_TFLOPS
        t_mfma, t_hbm = ex / (peak * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
        fm, fh = t_mfma / s, t_hbm / s
        if t_hbm > t_mfma:
            return dict(kernel=name, bound="hbm", achieved=nbytes / s / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=fh, frac_mfma=fm, frac_hbm=fh)
        return dict(kernel=name, bound="mfma", achieved=ex / s / 1e12, peak=peak, unit="TFLOP/s", frac=fm,
                    frac_mfma=fm, frac_hbm=fh, algorithmic_equiv=flops / s / 1e12)
    return dict(kernel=name, bound="hbm", achieved=nbytes / s / 1e9, peak=PEAK_HBM_GBS, unit="GB/s",
                frac=nbytes / s / 1e9 / PEAK_HBM_GBS)


def kernel_table(recs, passes, top=10):
    """Per-kernel roofline rows from HIP-event records of `passes` profiled passes:
    median duration per launch position, summed per kernel name."""
    per_pass = len(recs) // passes
    rows = {}
    for i in range(per_pass):
        name, _, fl, by = recs[i]
        ms = statistics.median(recs[p * per_pass + i][1] for p in range(passes))
        r = rows.setdefault(name, dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
        r["ms"] += ms
        r["n"] += 1
        r["flops"] += fl
        r["bytes"] += by
    total = sum(r["ms"] for r in rows.values())
    out = []
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        s = r["ms"] * 1e-3
        row = price(name, r["flops"], r["bytes"], s)
        if name.startswith("stem_conv"):
            # which floor `algorithmic_bytes` is for the fused resize / crop + stem: the bytes of the taps themselves.  The
            # no-antialias 1280 -> 256 resize touches 1 pixel in 5 on 2 rows in 4, and HBM delivers 64-byte sectors: the
            # sector-granular floor of fp32 frames is ~1.9x the tap bytes (PMC: profiles/*_pmc_traffic.json), so against
            # THAT floor the resize stem sits at ~2x the fraction printed here; uint8 frames are 4x lighter
            row["bytes_floor"] = "tap bytes (not sector-granular)"
        row.update(ms_per_step=r["ms"], launches_per_step=r["n"], avg_launch_ms=r["ms"] / r["n"],
                   share_of_step=r["ms"] / total, algorithmic_flops_per_launch=r["flops"] / r["n"],
                   algorithmic_bytes_per_launch=r["bytes"] / r["n"])
        out.append(row)
    # executed matrix-core FLOPs of the whole pass (all MFMA launches, not only the top rows)
    ex_total = sum(executed_flops(n, r["flops"]) for n, r in rows.items()
                   if r["flops"] > 0 and kernel_family(n) == "mfma")
    return out[:top], total, ex_total


def percentiles(ms):
    ms = sorted(ms)

    def q(p):
        return ms[min(len(ms) - 1, max(0, int(round(p * (len(ms) - 1)))))]
    return dict(median=statistics.median(ms), p10=q(0.10), p90=q(0.90), n=len(ms))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (never an
    exec: this parent stays a plain relay that has not touched the GPU), pass the children's
    stderr through, print rank 0's JSON line last and return the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    json_line = None
    for ln in reversed(lines):
        if ln.lstrip().startswith("{"):
            json_line = ln
            break
    for ln in lines:
        if ln is not json_line:
            print(ln, file=sys.stderr)
    if json_line is not None and out.returncode == 0:
        print(json_line, flush=True)
    elif out.returncode == 0:
        print("bench.py: the ranks exited 0 without printing a JSON line", file=sys.stderr)
        return 1
    return out.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--time-batch", type=int, default=None,
                    help="multi-view frames per time batch and HIP stream (default: 32 for cfg3 / cfg2, "
                         "8 for cfg5); at N > 1 every rank carries this many whole frames of work")
    ap.add_argument("--streams", type=int, default=3,
                    help="single GPU: independent time batches in flight on that many HIP streams; one "
                         "step = one time batch per stream")
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--model-size", default=None, choices=["small", "medium", "large"],
                    help="EfficientTrack size of both 2D networks (default: the config's -- small, the only size the "
                         "reference ships weights for; medium for def320, the reference's config default)")
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    ap.add_argument("--three-d", default="sharded", choices=["sharded", "rank0"],
                    help="multi-GPU placement of the 3D stage: frame-sharded over the ranks of a group "
                         "(default) or all of it on rank 0 (the literal BASELINE configs[3] placement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-uint8", action="store_true", help="skip the uint8-ingest side measurement")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary line (medium/medium models, SURVEY 8d)")
    ap.add_argument("--no-side-legs", action="store_true", help="multi-GPU: skip rank0-3D and replicas legs")
    ap.add_argument("--no-reduced-precision", action="store_true",
                    help="skip the separately labelled bf16x3 line (the headline is always fp32)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph")
    ap.add_argument("--sharded-streams", type=int, default=2,
                    help="multi-GPU: camera-sharded pipelines in flight per rank (own plans and HIP stream each, "
                         "one RCCL communicator: the collectives are issued in the same order on every rank)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="multi-GPU: do not overlap the heatmap exchange with the next step's 2D stage")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU (camera-sharded, RCCL) code path even with one rank")
    ap.add_argument("--profile-passes", type=int, default=5)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # nothing above has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: run `python bench.py --gpus N` without a "
                         "launcher, or torch.distributed.run with --nproc-per-node equal to --gpus" %
                         (args.gpus, world))
    # the bound is this NODE's share of the job (a multi-node launch has WORLD_SIZE > the node's GPUs; a rank
    # restricted by *_VISIBLE_DEVICES sees fewer): every local rank needs its own visible device
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    have = torch.cuda.device_count()
    if local_rank >= have:
        raise SystemExit("bench.py rank %d: --gpus %d needs %d GPUs on this node, %d visible" %
                         (rank, world, local_world, have))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ["NCCL_DEBUG"] = "WARN"          # keep RCCL's banner and warnings off stdout:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/jh_rccl_%h_%p.log")   # the JSON line is last
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as sk:                   # a free port (two benches on one box must not collide)
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
        if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl":
            raise SystemExit("bench.py: the RCCL process group has %d ranks (backend %s), --gpus is %d" %
                             (dist.get_world_size(), dist.get_backend(), args.gpus))

    from jarvis_hybridnet_amd import _native as N
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import MultiStreamPredictor, NativePredictor

    c = CONFIGS[args.config]
    if args.model_size is None:
        args.model_size = c.get("size", "small")
    if args.time_batch is None:
        args.time_batch = c["time_batch"]
    pipelined = sharded and not args.no_pipeline and args.three_d == "sharded"
    KS = max(1, args.sharded_streams) if pipelined else 1
    K = KS if sharded else (1 if args.graph else max(1, args.streams))
    # Weak scaling: every rank always does the work of `--time-batch` whole frames
    # (T*C images of 2D work, T frames of 3D work).  Ranks form groups of `gs` GPUs
    # that shard the cameras of T*gs frames; world/gs groups run side by side.
    from jarvis_hybridnet_amd.distributed import plan_groups
    gs, n_groups = plan_groups(world, c["C"])
    grank, gidx = rank % gs, rank // gs
    T = args.time_batch * gs                       # frames per group and step
    calib = S.ring_calibration(c["C"], c["W"], c["H"], c["focal"])
    size = args.model_size
    sd_c = S.efficienttrack_weights(size, 1, c["seeds"][0])
    sd_h = S.hybridnet_weights(size, c["J"], c["seeds"][1])
    # T_distinct seeded frame sets (different subject positions, hence different detections, crop
    # windows and voxel-grid centres in every slot of a time batch); frame 0 is the fixture case
    n_distinct = min(T, int(os.environ.get("JH_BENCH_DISTINCT", str(args.time_batch))))
    distinct = [S.blob_frames(calib, c["W"], c["H"], c["J"], c["seeds"][2] + i)[0] for i in range(n_distinct)]

    def device_frames(cam_lo, cam_n, frames=T):
        """(frames, cam_n, 3, H, W) on the GPU, assembled there from the distinct frames so
        the host never holds T copies (T = 128 frames per group at 4 GPUs per group)."""
        base = torch.stack([d[cam_lo:cam_lo + cam_n] for d in distinct]).to(dev)
        idx = torch.arange(frames, device=dev) % len(distinct)
        return base[idx].contiguous()

    def common_kw(model=size, frames=T):
        return dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                    roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
                    mean=S.MEAN, std=S.STD, time_batch=frames, center_model=model, kp_model=model)
    common = common_kw()
    calib_dev = [t.to(dev) for t in calib]
    step_events = []                               # per stream: one event per step (timed region)
    if not sharded:
        msp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, **common), streams=K, timing=True)
        msp.set_calibration(*calib_dev)
        pred = msp.preds[0]                        # the per-kernel profile and the side legs use one
        fr = device_frames(0, c["C"])
        outs = [(torch.empty((T, c["J"], 3), device=dev), torch.empty((T, c["J"]), device=dev),
                 torch.empty((T,), device=dev, dtype=torch.int32)) for _ in range(K)]
        out = outs[0]
        torch.cuda.synchronize()

        def step(record=False):
            # one step = one time batch per stream (K * T frame sets)
            for i in range(K):
                if K > 1:
                    msp.forward(fr, outs[i])
                    if record:
                        step_events[i].append(msp.last_event)
                else:
                    pred.forward(fr, outs[i])
                    if record:
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record()
                        step_events[0].append(ev)
            return outs
    else:
        from jarvis_hybridnet_amd.distributed import ShardedPredictor, camera_range
        groups = [dist.new_group(list(range(g * gs, (g + 1) * gs))) for g in range(n_groups)]
        cam_lo, cam_n = camera_range(c["C"], grank, gs)
        fr = device_frames(cam_lo, cam_n)

        def make_sharded(three_d, count):
            ps, ss = [], []
            for _ in range(count):
                t3 = T if three_d == "rank0" else T // gs
                p_ = NativePredictor(sd_c, sd_h, time_batch_3d=t3, cam_lo=cam_lo, cam_n=cam_n, **common)
                p_.set_calibration(*calib_dev)
                ps.append(p_)
                ss.append(ShardedPredictor(p_, num_cameras=c["C"], num_joints=c["J"], time_batch=T,
                                           heat_shape=(p_.Hh, p_.Hh, p_.Jp), rank=grank, world=gs,
                                           device=dev, exchange=args.exchange, group=groups[gidx],
                                           three_d=three_d))
            return ps, ss
        preds, shs = make_sharded(args.three_d, KS)
        pred, sh = preds[0], shs[0]
        sh_streams = [torch.cuda.Stream() for _ in range(KS)]
        torch.cuda.synchronize()

        def step(record=False):
            return [sh.step(fr)]

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    step_events = [[] for _ in range(K if not sharded else 1)]
    run = (lambda: step(True)) if not sharded else step
    if args.graph and not sharded:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        run = g.replay
        run()
    if pipelined:
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
    if not sharded and not args.graph:             # t = 0 marks for the per-step percentiles
        for i in range(K):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(msp.streams[i] if K > 1 else torch.cuda.current_stream())
            step_events[i].append(ev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    if pipelined:
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
    res = [[r.clone() for r in o] for o in res]    # the output buffers are reused below
    valid = sum(int(o[2].sum().item()) for o in res) * (n_groups * K if sharded else 1)
    rank_fail = None
    if sharded:
        # every rank holds the (T,) validity vector of its group's time batch, and every group runs the same seeded
        # frame sets: the counts must agree on ALL ranks, or a rank has computed something else (exit code 4 below)
        vt = torch.tensor([valid], device=dev, dtype=torch.int64)
        vl = [torch.zeros_like(vt) for _ in range(world)]
        dist.all_gather(vl, vt)
        valid_per_rank = [int(v.item()) for v in vl]
        if len(set(valid_per_rank)) != 1:
            rank_fail = "valid_frames_per_step differs between the ranks: %s" % valid_per_rank
    frames_per_step = T * K * n_groups
    fps = frames_per_step * args.steps / dt

    line = {
        "metric": c["metric"],
        "value": fps, "unit": "multi-view frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded blob frames, ring calibration, random-init weights)",
        "config": {"workload": c["workload"] if size == c.get("size", "small") else
                   c["workload"].replace("%s/%s" % ((c.get("size", "small"),) * 2), "%s/%s" % (size, size)),
                   "cameras": c["C"], "frame": [c["H"], c["W"]], "joints": c["J"],
                   "grid": int(c["roi"] / c["spacing"]), "time_batch": T, "streams": K,
                   "frames_per_step": frames_per_step, "valid_frames_per_step": valid,
                   "parallelism": ("single GPU, %d time batches in flight on %d HIP streams" % (K, K))
                   if not sharded else
                   "%d group(s) x %d GPUs: camera-sharded 2D (%d cams/GPU) + RCCL %s of heatmaps%s + %s; "
                   "%d such pipeline(s) per rank" % (
                       n_groups, gs, c["C"] // gs, shs[0].exchange,
                       " (overlapped with the next step's CenterDetect)" if pipelined else "",
                       "3D stage on rank 0 of the group" if args.three_d == "rank0" else "frame-sharded 3D", KS),
                   "launches_per_step": int(pred.launches) * K, "launches_per_time_batch": int(pred.launches),
                   "hipgraph": bool(args.graph)},
    }
    if not sharded and not args.graph and step_events[0]:
        # distribution over the timed steps.  One sample = the time one stream needs
        # for one time batch while the other streams' batches share the GPU (HIP events).
        ms = []
        for evs in step_events:
            ms += [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        p = percentiles(ms)
        line["step_time_ms"] = dict(p, note="per time batch of %d frames on one of %d concurrent streams "
                                    "(HIP events); frames/s = streams * time_batch / ms" % (T, K))
        line["frames_per_s_median"] = K * T / p["median"] * 1e3
        line["frames_per_s_p10_p90"] = [K * T / p["p90"] * 1e3, K * T / p["p10"] * 1e3]

    parity_fail = False
    P = max(1, args.profile_passes)
    if sharded and rank != 0:
        for _ in range(P + 1):                     # rank 0 profiles P+1 steps: keep the collectives matched
            step()
    if rank == 0:
        # ---- per-kernel roofline, HIP events around every launch of ONE stream's time batch
        # (kernels timed alone; at N > 1: rank 0's share of the camera-sharded step).  Median
        # over P passes per launch position, after one discarded profiled pass.
        prof_fn = (lambda: step()) if sharded else (lambda: pred.forward(fr, out))
        N.profile(prof_fn)
        recs = []
        for _ in range(P):
            recs += N.profile(prof_fn)
        table, total_ms, ex_flops = kernel_table(recs, P, int(os.environ.get("JH_BENCH_TOP", "10")))
        top = table[0]
        T_prof = T // gs if sharded else T
        roof = {k: top[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                    "launches_per_step", "share_of_step")}
        roof["traffic"] = None
        roof["time_batch"] = T_prof
        if top["bound"] == "mfma":
            roof["algorithmic_equiv"] = top["algorithmic_equiv"]
            roof["note"] = ("achieved = FLOPs the matrix cores execute (channel padding included%s) / HIP-event "
                            "duration; algorithmic_equiv = the same launch counted as the direct convolution it "
                            "computes" % (", Winograd F(2x2,3x3) x 3 z taps: 12 of the direct algorithm's 27 "
                                          "multiplies" if "wino" in top["kernel"] else ""))
        roof["algorithmic_bytes"] = top["algorithmic_bytes_per_launch"]
        # HBM bytes of that kernel: rocprofv3 cannot run inside this process, so the PMC passes
        # (tools/pmc_traffic.sh: this bench command under --pmc FETCH_SIZE / WRITE_SIZE) are
        # committed under profiles/ together with the SHA-256 of the csrc/ tree they were taken
        # on.  They are reported -- scaled to this run's time batch -- only when this run's
        # sources are that tree; otherwise `traffic` stays null and says why.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_traffic import csrc_sha256
            pmc = json.load(open(os.path.join(ROOT, PMC_TRAFFIC)))
            if pmc.get("csrc_sha256") != csrc_sha256():
                roof["traffic_note"] = "%s was measured on another csrc/ tree (sha256 %s...)" % (
                    PMC_TRAFFIC, str(pmc.get("csrc_sha256"))[:12])
            elif (pmc.get("config", "cfg3"), pmc.get("model_size", "small")) != (args.config, size):
                # several kernel names carry no shape (norm_apply, reproject_gather, ...): byte counts of another
                # workload must not be reported under them
                roof["traffic_note"] = "%s was measured on %s / %s, this run is %s / %s" % (
                    PMC_TRAFFIC, pmc.get("config", "cfg3"), pmc.get("model_size", "small"), args.config, size)
            elif top["kernel"] in pmc:
                e = pmc[top["kernel"]]
                roof["traffic"] = e["hbm_bytes_per_launch"] * T_prof / pmc["time_batch"]
                roof["traffic_source"] = "%s (rocprofv3 --pmc passes of `%s`, csrc sha256 %s...)" % (
                    PMC_TRAFFIC, pmc.get("command", "bench.py"), pmc["csrc_sha256"][:12])
                for row in table:                     # same figure for the other measured kernels
                    if row["kernel"] in pmc and isinstance(pmc[row["kernel"]], dict):
                        row["traffic"] = pmc[row["kernel"]]["hbm_bytes_per_launch"] * T_prof / pmc["time_batch"]
            else:
                roof["traffic_note"] = "no PMC entry for %s in %s" % (top["kernel"], PMC_TRAFFIC)
        except (OSError, ValueError, KeyError, ImportError) as e:
            roof["traffic_note"] = "no usable %s: %r" % (PMC_TRAFFIC, e)
        line["roofline"] = roof
        line["kernels"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in row.items()
                            if k not in ("algorithmic_flops_per_launch", "algorithmic_bytes_per_launch")}
                           for row in table]
        line["kernel_time_ms_per_time_batch"] = total_ms
        if not sharded:
            # the whole step against the matrix-core peak: FLOPs the MFMA kernels execute per step (K time batches)
            # over the step's wall time -- what the non-MFMA kernels, stalls and launch gaps leave of the peak
            e2e = ex_flops * K / (dt / args.steps) / 1e12
            line["roofline_end_to_end"] = {
                "bound": "mfma", "achieved": e2e, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": e2e / PEAK_F32_MFMA_TFLOPS,
                "executed_gflop_per_frame": ex_flops / T / 1e9,
                "algorithmic_gflop_per_frame": sum(r[2] for r in recs[:len(recs) // P]) / T / 1e9,
                "note": "executed matrix-core FLOPs of one step (channel padding included, Winograd at 12/27 of the "
                        "direct count) / ms_per_step; the remainder is HBM-bound kernels, stalls and launch gaps"}

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
        fps_res_k = None
        if K > 1:                                  # the headline's form (K time batches in flight), uint8 frames
            for _ in range(2):
                for i in range(K):
                    msp.forward(dev_u8, outs[i])
            msp.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                for i in range(K):
                    msp.forward(dev_u8, outs[i])
            msp.synchronize()
            torch.cuda.synchronize()
            fps_res_k = T * K * 6 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(10):
            dev_u8.copy_(host, non_blocking=True)
            pred.forward(dev_u8, out)
        torch.cuda.synchronize()
        fps_pcie = T * 10 / (time.perf_counter() - t0)
        line["uint8_ingest"] = {"frames_per_s_resident": fps_res,
                                "frames_per_s_resident_%d_streams" % K: fps_res_k,
                                "frames_per_s_incl_pcie_h2d_serial": fps_pcie,
                                "bytes_per_frame": int(host[0].numel()),
                                "note": "one stream unless named; pinned host uint8 BGR -> HBM inside the timed region"}
        # ---- the SHIPPED driver: predict3D_frames itself, fed a generator of numpy uint8 BGR frame sets as a
        # decoder yields them, data3D.csv written to /dev/shm; pinned staging by a thread pool, upload on a copy
        # stream, K time batches in flight (jarvis_hybridnet_amd/prediction/_ingest.py).  PCIe-inclusive: never `value`.
        try:
            import shutil
            from types import SimpleNamespace as NS
            from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
            from jarvis_hybridnet_amd.prediction.predict3D import frame_row, predict3D_frames
            from jarvis_hybridnet_amd.prediction._ingest import release_ingest_buffers
            cfg_ns = NS(PARENT_DIR="/nonexistent", PROJECT_NAME="bench",
                        DATASET=NS(DATASET_ROOT_DIR="x", MEAN=S.MEAN, STD=S.STD),
                        CENTERDETECT=NS(MODEL_SIZE=size, NUM_JOINTS=1, IMAGE_SIZE=c["center"]),
                        KEYPOINTDETECT=NS(MODEL_SIZE=size, NUM_JOINTS=c["J"], BOUNDING_BOX_SIZE=c["bbox"]),
                        HYBRIDNET=NS(NUM_CAMERAS=c["C"], ROI_CUBE_SIZE=c["roi"], GRID_SPACING=c["spacing"]))
            jp = JarvisPredictor3D(cfg_ns, sd_c, sd_h)
            sets = [u8[i].numpy() for i in range(len(distinct))]

            def decoded(n):
                for i in range(n):
                    yield sets[i % len(sets)]
            Kd = max(1, args.streams)
            odir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", "jh_bench_driver_%d" % os.getpid())
            predict3D_frames(jp, decoded(T * Kd), *calib_dev, cfg_ns, odir, time_batch=T, streams=Kd)   # plans + pinning
            nd = T * Kd * 8
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = predict3D_frames(jp, decoded(nd), *calib_dev, cfg_ns, odir, time_batch=T, streams=Kd)
            ddt = time.perf_counter() - t0
            import csv as _csv
            rows = list(_csv.reader(open(os.path.join(odir, "data3D.csv"))))
            pred.forward(dev_u8, out)                            # the resident path on the same bytes
            torch.cuda.synchronize()
            rp, rc, rv = [t.cpu() for t in out]
            same = len(rows) == nd and all(
                rows[t] == [str(v) for v in frame_row(rp[t] if int(rv[t]) else None, rc[t] if int(rv[t]) else None, c["J"])]
                for t in range(min(T, nd)))
            line["driver"] = {"frames_per_s": got / ddt, "frames": got, "time_batch": T, "streams": Kd,
                              "rows_equal_resident_forward": bool(same),
                              "pipeline_slots": next(iter(jp._ingest_cache.values())).slots,
                              "note": "predict3D_frames() end to end: generator of numpy uint8 BGR frame sets -> pinned "
                                      "staging (thread pool) -> H2D on a copy stream -> MultiStreamPredictor -> data3D.csv "
                                      "on /dev/shm; PCIe bound = bytes_per_frame / host-to-device bandwidth"}
            release_ingest_buffers(jp)
            shutil.rmtree(odir, ignore_errors=True)
            del jp
        except Exception as e:                                   # noqa: BLE001 -- a side leg never loses the headline
            line["driver"] = {"error": repr(e)}
        del dev_u8, host

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle (port of the reference) on the host cores,
        # same workload, bounded sample of whole frames
        from oracle import hybridnet_oracle as O
        cores = usable_cores()
        torch.set_num_threads(cores)
        kw = dict(center_size=c["center"], bbox=c["bbox"], roi_cube_size=c["roi"],
                  grid_spacing=c["spacing"], mean=S.MEAN, std=S.STD, center_model=size, kp_model=size,
                  chunk=5)
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
            # The oracle runs on THIS host's CPU.  torch's CPU kernels differ in the last bit between CPU models,
            # which flips a few of the reference's own reprojection gather indices (a truncation of u/2, v/2 within an
            # ulp of an integer) against the fixtures' host; the HIP path's indices equal the fixture host's bit for bit
            # (tests/test_hip_stages.py::test_reprojection).  So besides the raw distance the oracle is re-run from the
            # gather on with the HIP indices substituted: that figure is host independent and must meet the 1e-3 mm bar
            # (bench.py exits non-zero otherwise); the flips are listed with their distance to the truncation boundary.
            pts0 = res[0][0][0].cpu()
            line["parity_max_abs_mm_vs_host_oracle"] = (pts0 - ref[0][0]).abs().max().item()
            try:
                from types import SimpleNamespace as NS
                from jarvis_hybridnet_amd.hybridnet.repro_layer import ReprojectionLayer
                inter = {}
                with torch.no_grad():
                    O.predictor3d_forward(sd_c, sd_h, distinct[0], *calib, intermediates=inter, **kw)
                c3i, chm = inter["center3d"].int()[None], inter["center_hm"][None]
                layer = ReprojectionLayer(NS(HYBRIDNET=NS(GRID_SPACING=c["spacing"], ROI_CUBE_SIZE=c["roi"],
                                                          NUM_CAMERAS=c["C"]),
                                             KEYPOINTDETECT=NS(BOUNDING_BOX_SIZE=c["bbox"])))
                idx = layer.gather_indices(inter["heatmaps_padded"].to(dev), c3i.to(dev), chm.to(dev),
                                           *[t[None] for t in calib_dev]).cpu()
                with torch.no_grad():
                    hp = O.host_parity(sd_h, inter, idx, pts0[None], ref[0], calib, c["roi"], c["spacing"], c["bbox"],
                                       chunk=5)
                line["parity_max_abs_mm_vs_host_oracle_same_indices"] = hp["same_indices_mm"]
                line["host_oracle_index_flips"] = {
                    "flips": hp["flips"], "of": hp["of"],
                    "max_dist_to_truncation_boundary": max([r["dist_to_integer"] for r in hp["flip_voxels"]] or [0.0]),
                    "voxels": [dict(voxel=r["voxel"], half_u=round(r["half_u"], 7), half_v=round(r["half_v"], 7),
                                    dist=r["dist_to_integer"]) for r in hp["flip_voxels"][:8]],
                    "note": "gather indices of the oracle run on this host's CPU that differ from the HIP path's (= the "
                            "reference fixtures'); each is a truncation tie of the reference's own float32 arithmetic "
                            "(dist = |u/2 or v/2 - nearest integer| on this host).  parity_..._same_indices = the "
                            "oracle re-run from the gather on with the HIP indices: the host-independent figure"}
                parity_fail = hp["same_indices_mm"] >= 1e-3
                del idx, layer
            except Exception as e:                                  # noqa: BLE001
                line["host_oracle_index_flips"] = {"error": repr(e)[:200]}
    fixture_tag = c.get("fixture", args.config) if size == c.get("size", "small") else None
    if rank == 0 and world == 1 and fixture_tag is not None:
        # frame 0 of this workload is a fixture case of tests/golden/predictor.npz, i.e. the
        # output of the imported upstream reference on the same input
        import numpy as np
        gpath = os.path.join(ROOT, "tests", "golden", "predictor.npz")
        if os.path.isfile(gpath):
            gold = np.load(gpath)[fixture_tag + ".points3D"]
            line["parity_max_abs_mm_vs_reference_fixture"] = float(
                np.abs(res[0][0][0].cpu().numpy() - gold[0]).max())

    if rank == 0 and not sharded and T >= 3:
        # ---- side check (outside the timed region): one frame set of the batch replaced by an empty (all
        # zero) one must not disturb its neighbours (time-batch slots are independent instances).  Whether the
        # empty set itself is reported invalid depends on the weights: with these random-init networks the
        # centre heatmaps exceed the detection threshold on any input (the (None, None) path of
        # jarvis3D.py:157,187-190 is pinned by the fixture case cfg2_none, whose weights are scaled for it).
        try:
            fr2 = fr.clone()
            fr2[1].zero_()
            chk = [t.clone() for t in pred.forward(fr2, None)]
            base = [t.clone() for t in pred.forward(fr, None)]
            torch.cuda.synchronize()
            keep = [i for i in range(T) if i != 1]
            line["invalid_frame_check"] = {
                "valid_flag_of_empty_frame": int(chk[2][1].item()),
                "other_frames_bit_equal": bool(torch.equal(chk[0][keep], base[0][keep]) and
                                               torch.equal(chk[2][keep], base[2][keep]))}
            del fr2, chk, base
        except Exception as e:
            line["invalid_frame_check"] = {"error": repr(e)[:200]}

    if rank == 0 and not sharded and not args.no_reduced_precision and pred.precision == "f32":
        # ---- separately labelled reduced-precision line (BASELINE configs[1] is worded "bf16"; the
        # reference's own fast path is half precision, jarvis3D.py:93,107,122): V2V's 3x3x3 convolutions
        # on the bf16 matrix cores with split operands (csrc/conv3d_bf16x3.hip), everything else as in
        # the headline.  Same frames, same method, half the steps; never the headline `value`.
        try:
            # (precision is a property of the predictor, jh_predictor_config.precision: these coexist with the
            # fp32 predictors of the headline in this process)
            rp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, precision="bf16x3", **common), streams=K)
            rp.set_calibration(*calib_dev)
            routs = [(torch.empty((T, c["J"], 3), device=dev), torch.empty((T, c["J"]), device=dev),
                      torch.empty((T,), device=dev, dtype=torch.int32)) for _ in range(K)]
            for _ in range(2):
                for i in range(K):
                    rp.forward(fr, routs[i])
            torch.cuda.synchronize()
            nst = max(2, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(nst):
                for i in range(K):
                    rp.forward(fr, routs[i])
            torch.cuda.synchronize()
            rdt = time.perf_counter() - t0
            N.profile(lambda: rp.preds[0].forward(fr, routs[0]))
            rtab, _, _ = kernel_table(N.profile(lambda: rp.preds[0].forward(fr, routs[0])), 1, 200)
            rk = [r for r in rtab if "bf16x3" in r["kernel"]]
            red = {"mode": "bf16x3", "value": T * K * nst / rdt, "unit": "multi-view frames/s",
                   "ms_per_step": 1e3 * rdt / nst, "steps": nst,
                   "dtype": "bf16x3 (fp32 operands split into two bf16 terms, three bf16 MFMAs per product, fp32 "
                            "accumulation) for the 3x3x3 stride-1 convolutions of V2V; f32 everywhere else",
                   "parity_max_abs_mm_vs_f32_mode": (routs[0][0] - res[0][0]).abs().max().item(),
                   # 3D MPJPE against the fp32 mode over the T frame sets of one time batch (mm)
                   "mpjpe_mm_vs_f32_mode": (routs[0][0] - res[0][0]).norm(dim=-1).mean().item(),
                   "kernels": [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in r.items()
                                if k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                         "launches_per_step", "algorithmic_equiv")} for r in rk]}
            gpath = os.path.join(ROOT, "tests", "golden", "predictor.npz")
            if os.path.isfile(gpath) and fixture_tag is not None:
                import numpy as np
                gold = np.load(gpath)[fixture_tag + ".points3D"]
                red["parity_max_abs_mm_vs_reference_fixture"] = float(
                    np.abs(routs[0][0][0].cpu().numpy() - gold[0]).max())
            line["reduced_precision"] = red
            del rp, routs
        except Exception as e:                      # the side line never costs the headline
            line["reduced_precision"] = {"error": repr(e)[:300]}

    if rank == 0 and not sharded and not args.no_secondary:
        # ---- the reference's own call pattern: ONE multi-view frame set per forward (batch 1),
        # stream-ordered back to back (no host sync inside the loop)
        try:
            p1 = NativePredictor(sd_c, sd_h, **common_kw(size, 1))
            p1.set_calibration(*calib_dev)
            f1 = fr[:1].contiguous()
            o1 = (torch.empty((1, c["J"], 3), device=dev), torch.empty((1, c["J"]), device=dev),
                  torch.empty((1,), device=dev, dtype=torch.int32))
            lat = {}
            for mode, on in (("graph", True), ("eager", False)):
                p1.graph_replay = on
                for _ in range(10):
                    p1.forward(f1, o1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    p1.forward(f1, o1)
                torch.cuda.synchronize()
                lat[mode] = 1e3 * (time.perf_counter() - t0) / 100
            # the product path (JarvisPredictor3D.forward, one frame set per call as the reference
            # driver calls it) replays a hipGraph; `eager` = the same ~150 launches one by one
            line["single_frame_latency_ms"] = lat["graph"]
            line["single_frame_latency_ms_eager"] = lat["eager"]
            del p1
        except Exception as e:
            line["single_frame_latency_ms"] = repr(e)[:200]

    if rank == 0 and not sharded and not args.no_secondary and size == "small" and args.config == "cfg3":
        # ---- SURVEY 8d secondary line: the `medium` models (config.py's default size) on the
        # same workload, same method, fewer steps
        try:
            Tm = min(T, 16)
            sd_cm = S.efficienttrack_weights("medium", 1, 50)
            sd_hm = S.hybridnet_weights("medium", c["J"], 51)
            mm = MultiStreamPredictor(lambda: NativePredictor(sd_cm, sd_hm, **common_kw("medium", Tm)),
                                      streams=K)
            mm.set_calibration(*calib_dev)
            frm = device_frames(0, c["C"], Tm)
            for _ in range(2 * K):
                mm.forward(frm)
            torch.cuda.synchronize()
            nst = max(2, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(nst * K):
                mm.forward(frm)
            torch.cuda.synchronize()
            mdt = time.perf_counter() - t0
            sec = {"workload": c["workload"].replace("small/small", "medium/medium"),
                   "value": Tm * K * nst / mdt, "unit": "multi-view frames/s",
                   "ms_per_step": 1e3 * mdt / nst, "time_batch": Tm, "streams": K, "steps": nst, "dtype": "f32"}
            # same measurement method as the headline: HIP events around every launch of one stream's time batch
            mp = mm.preds[0]
            N.profile(lambda: mp.forward(frm))
            mrecs = []
            for _ in range(3):
                mrecs += N.profile(lambda: mp.forward(frm))
            mtab, mtot, mex = kernel_table(mrecs, 3, 5)
            sec["roofline"] = {k: mtab[0][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                      "avg_launch_ms", "launches_per_step", "share_of_step")}
            sec["roofline"]["traffic"] = None
            sec["kernels"] = [{k: (round(v, 5) if isinstance(v, float) else v) for k, v in row.items()
                               if k in ("kernel", "bound", "frac", "ms_per_step", "launches_per_step", "share_of_step")}
                              for row in mtab]
            sec["kernel_time_ms_per_time_batch"] = mtot
            e2e = mex * K / (mdt / nst) / 1e12
            sec["roofline_end_to_end"] = {"bound": "mfma", "achieved": e2e, "peak": PEAK_F32_MFMA_TFLOPS,
                                          "unit": "TFLOP/s", "frac": e2e / PEAK_F32_MFMA_TFLOPS}
            # frame 0 with the fixture's weights (tests/cases.py: cfg3_medium) against the imported reference's output
            import numpy as np
            gpath = os.path.join(ROOT, "tests", "golden", "predictor.npz")
            if os.path.isfile(gpath):
                from tests import cases as _cases
                fc = _cases.PREDICTOR_CASES["cfg3_medium"]
                fp = NativePredictor(S.efficienttrack_weights("medium", 1, fc["cseed"]),
                                     S.hybridnet_weights("medium", c["J"], fc["hseed"]), **common_kw("medium", Tm))
                fp.set_calibration(*calib_dev)
                f0 = S.blob_frames(calib, c["W"], c["H"], c["J"], fc["fseed"])[0].to(dev)
                fo = fp.forward(f0[None].expand(Tm, -1, -1, -1, -1).contiguous())
                torch.cuda.synchronize()
                gold = np.load(gpath)["cfg3_medium.points3D"]
                sec["parity_max_abs_mm_vs_reference_fixture"] = float(np.abs(fo[0][0].cpu().numpy() - gold[0]).max())
                del fp, fo
            line["secondary"] = sec
            del mm, frm
        except Exception as e:                      # the secondary line never costs the headline
            line["secondary"] = {"error": repr(e)[:200]}

    if sharded:
        # ---- per-stage timeline of three pipelined submits on every rank (HIP events on the streams the stages
        # run on, ms since the first submit): what DESIGN.md section 5's prediction of the overlap is checked against
        # without a second run.  Collectives inside: every rank runs this, in the same order.
        mine = None
        try:
            sh.trace = []
            torch.cuda.synchronize()
            for _ in range(3):
                sh.submit(fr)
            sh.flush()
            torch.cuda.synchronize()
            t00 = sh.trace[0][1]
            mine = [[lab, round(t00.elapsed_time(ev), 3)] for lab, ev in sh.trace]
        except Exception as e:                                      # noqa: BLE001
            mine = {"error": repr(e)[:200]}
        sh.trace = None
        # every rank reaches this gather whatever happened above (a rank that failed contributes its error string:
        # nobody is left alone inside a collective)
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        line["stage_ms_per_rank"] = allr
        line["rccl"] = {"world": dist.get_world_size(), "group_size": gs, "groups": n_groups,
                        "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                        "exchange": shs[0].exchange, "three_d": args.three_d,
                        "valid_frames_per_step_per_rank": valid_per_rank}
        assert line["rccl"]["world"] == args.gpus == world
    if sharded and (world > 1 or os.environ.get("JH_BENCH_SIDE_LEGS")) and not args.no_side_legs:
        line.update(side_legs(args, c, common, sd_c, sd_h, calib_dev, device_frames, dev, dist, world, T, gs,
                              make_sharded if args.three_d == "sharded" else None, fr, barrier))
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)       # the ONE JSON line, last thing on stdout
        if rank_fail:
            print("bench.py: " + rank_fail, file=sys.stderr)
            raise SystemExit(4)
        if parity_fail:
            print("bench.py: 3D keypoints differ from the host oracle (same gather indices) by >= 1e-3 mm",
                  file=sys.stderr)
            raise SystemExit(3)


def side_legs(args, c, common, sd_c, sd_h, calib_dev, device_frames, dev, dist, world, T, gs, make_sharded, fr,
              barrier):
    """Multi-GPU side measurements next to the camera-sharded `value`: (1) the
    literal north-star placement -- heatmaps all-gathered, 3D stage on rank 0 -- and (2) the
    frame-parallel upper bound `replicas_only`.  Failure-safe: set-up happens under try on
    every rank, then ONE unconditional all_reduce tells every rank whether all are ready, so no
    rank is ever left alone inside a collective; a failed leg reports an error string."""
    from jarvis_hybridnet_amd._predictor import MultiStreamPredictor, NativePredictor
    out = {}

    def all_ok(ok):
        flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    # ---- (1) 3D stage on rank 0 of each group (unpipelined: rank 0's V2V is the critical path)
    if make_sharded is not None:
        sh0, err = None, None
        try:
            _, ss = make_sharded("rank0", 1)
            sh0 = ss[0]
        except Exception as e:
            err = repr(e)[:200]
        if all_ok(sh0 is not None):
            for _ in range(2):
                sh0.step(fr)
            barrier()
            t0 = time.perf_counter()
            nst = max(2, args.steps // 2)
            for _ in range(nst):
                sh0.step(fr)
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out["three_d_rank0"] = {"value": T * (world // gs) * nst / tt.item(), "unit": "multi-view frames/s",
                                    "note": "BASELINE configs[3] placement: RCCL all-gather of heatmaps, "
                                            "reprojection + V2V + soft-argmax of all %d frames on rank 0 of "
                                            "each group (Amdahl-bound by construction)" % T}
        else:
            out["three_d_rank0"] = {"error": err or "set-up failed on another rank"}
        del sh0
        torch.cuda.empty_cache()

    # ---- (2) frame-parallel upper bound: every rank runs the whole path on its own frames
    rp, err, Tb, Kr = None, None, args.time_batch, max(1, args.streams)
    try:
        rp = MultiStreamPredictor(lambda: NativePredictor(sd_c, sd_h, **dict(common, time_batch=Tb)), streams=Kr)
        rp.set_calibration(*calib_dev)
        rfr = device_frames(0, c["C"], Tb)
        for _ in range(2 * Kr):
            rp.forward(rfr)
        torch.cuda.synchronize()
    except Exception as e:
        err, rp = repr(e)[:200], None
    if all_ok(rp is not None):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps * Kr):
            rp.forward(rfr)
        barrier()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out["replicas_only"] = {"value": Tb * Kr * world * args.steps / tt.item(), "unit": "multi-view frames/s",
                                "note": "frame-parallel upper bound: each rank runs all %d cameras of its own "
                                        "frames (%d streams x %d frames per step), no collective" % (
                                            c["C"], Kr, Tb)}
    else:
        out["replicas_only"] = {"error": err or "set-up failed on another rank"}
    return out


if __name__ == "__main__":
    main()
