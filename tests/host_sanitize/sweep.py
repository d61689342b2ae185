#!/usr/bin/env python3
"""Sweep of the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only).

Runs inside `LD_PRELOAD=<libclang_rt.asan> JH_LIBRARY_PATH=tests/host_sanitize/build/libjarvis_hip_san.so` (started by
tests/test_host_sanitize.py).  The library is the real csrc/ compiled --cuda-host-only against a malloc-backed HIP
stand-in (hip_stub.cpp): plan building (weight packing, tile / table builders, arena sizing, every upload and clear),
the launch arithmetic of every forward (grids, LDS sizes, the tile-form and column-block rules) and the C ABI's argument
checks all execute; kernels do not (there is no device code in this build), launch configurations the hardware would
refuse are reported.  Any sanitizer finding aborts the process.

Swept: V2V plans over grids 16..128 step 4 x joints 1..64; the reprojection launcher over cameras 2..48 x grids x
joints x crop sizes; EfficientTrack plans over the three model sizes x joints x image sides x batch classes; whole
predictors (create + calibration + forward in both time-batch classes, sharded camera ranges, uint8 ingest, graph
capture path) over seeded random configurations; jh_params_* with ragged inputs.
"""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from jarvis_hybridnet_amd import _native as N  # noqa: E402
from jarvis_hybridnet_amd import arch, synthetic as S  # noqa: E402
from jarvis_hybridnet_amd._predictor import NativePredictor  # noqa: E402

assert "san" in os.path.basename(N.LIB_PATH), "run against the sanitizer build (JH_LIBRARY_PATH)"
lib = N.lib()
for f in ("jh_stub_launches", "jh_stub_rejected"):
    getattr(lib, f).restype = ctypes.c_long
QUICK = os.environ.get("JH_SAN_QUICK") == "1"
BUDGET_BYTES = 0.5e9                      # per plan: the stand-in's "device" memory is host memory
refused = []                              # configurations the library declines with a message (not findings)
t00 = time.time()


def ok(rc, what, may_refuse=()):
    if rc != 0:
        msg = lib.jh_last_error().decode()
        if any(m in msg for m in may_refuse):
            refused.append((what, msg))
            return False
        raise SystemExit("FAILED %s: %s" % (what, msg))
    return True


def rand_state(spec, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(tuple(s), generator=g) * 0.1).contiguous() for k, s in spec}


# ---------------------------------------------------------------- V2V plans + forward
def sweep_v2v():
    n = 0
    joints = [1, 2, 3, 5, 8, 13, 23, 30, 46, 64]
    grids = list(range(16, 129, 4))
    if QUICK:
        joints, grids = [3, 23, 64], list(range(16, 129, 12))
    for G in grids:
        for J in joints:
            Jp = (J + 7) // 8 * 8
            for T in (1, 9):
                if T * 2 * Jp * (G // 2) ** 3 * 4 * 12 > BUDGET_BYTES:      # ~12 activations of the 2J-wide stage
                    continue
                sd = rand_state(arch.v2v_params(J), 7)
                params = N.Params(sd)
                h = ctypes.c_void_p()
                ok(lib.jh_v2v_create(params.handle, b"", J, T, G, ctypes.byref(h)), "v2v_create J=%d G=%d T=%d" % (J, G, T))
                x = torch.zeros((T, J, G, G, G))
                y = torch.zeros((T, J, G // 2, G // 2, G // 2))
                ok(lib.jh_v2v_forward(h, x.data_ptr(), y.data_ptr(), None), "v2v_forward J=%d G=%d T=%d" % (J, G, T))
                lib.jh_v2v_destroy(h)
                n += 1
    return n


# ---------------------------------------------------------------- reprojection launcher
def sweep_reproject():
    n = 0
    cams = [2, 3, 4, 12, 16, 24, 48]
    grids = list(range(16, 129, 4))
    joints = [1, 5, 8, 12, 23, 30, 40, 64]
    if QUICK:
        cams, grids, joints = [2, 12, 48], list(range(16, 129, 16)), [1, 23, 64]
    for C in cams:
        calib = S.ring_calibration(C, 1280, 1024, 1800.0)
        for G in grids:
            for J in joints:
                for bbox in (64, 256, 320):
                    hs = bbox // 2 + 2
                    if J * G ** 3 * 4 > BUDGET_BYTES or C * J * hs * hs * 4 > BUDGET_BYTES:
                        continue
                    if (C, G, J, bbox) != (cams[0], grids[0], joints[0], 64) and random.random() > (0.5 if QUICK else 0.12):
                        continue                      # a seeded sample of the product
                    ws_bytes = lib.jh_reproject_workspace_bytes(C, J, hs, G)
                    assert ws_bytes > 0
                    ws = torch.zeros((ws_bytes,), dtype=torch.uint8)
                    hm = torch.zeros((C, J, hs, hs))
                    vol = torch.zeros((1, J, G, G, G))
                    idx = torch.zeros((C, G, G, G), dtype=torch.int32)
                    c3 = torch.zeros((3,), dtype=torch.int32)
                    chm = torch.full((C, 2), 400, dtype=torch.int32)
                    ok(lib.jh_reproject_forward(hm.data_ptr(), C, J, hs, c3.data_ptr(), chm.data_ptr(), calib[0].data_ptr(),
                                                calib[1].data_ptr(), calib[2].data_ptr(), G, 2.0, vol.data_ptr(),
                                                idx.data_ptr(), ws.data_ptr(), ws.numel(), None),
                       "reproject C=%d G=%d J=%d hs=%d" % (C, G, J, hs),
                       # (many cameras x more than 32 channels: the voxel-row form's coarse-table tile; a clean refusal)
                       may_refuse=("coarse table tile does not fit LDS",))
                    n += 1
    return n


# ---------------------------------------------------------------- EfficientTrack plans + forward
def sweep_efftrack():
    n = 0
    for size in ("small", "medium", "large"):
        for J in ((1, 23) if QUICK else (1, 2, 12, 23, 30, 64)):
            sd = rand_state(arch.efficienttrack_params(size, J), 3)
            params = N.Params(sd)
            for side in ((64, 320) if QUICK else (64, 128, 192, 256, 320, 384)):
                for nimg in (1, 3, 16):                     # (16 images = the row-streaming class of a 2-camera rig)
                    if nimg * side * side * 4 * 2000 > BUDGET_BYTES * (8 if size == "small" else 4):
                        continue
                    h = ctypes.c_void_p()
                    ok(lib.jh_efftrack_create(params.handle, b"", arch.SIZE_IDS[size], J, nimg, side, side, 1,
                                              ctypes.byref(h)), "efftrack_create %s J=%d %d^2 n=%d" % (size, J, side, nimg))
                    x = torch.zeros((nimg, 3, side, side))
                    r1 = torch.zeros((nimg, J, side // 4, side // 4))
                    r2 = torch.zeros((nimg, J, side // 2, side // 2))
                    ok(lib.jh_efftrack_forward(h, x.data_ptr(), r1.data_ptr(), r2.data_ptr(), None),
                       "efftrack_forward %s J=%d %d^2 n=%d" % (size, J, side, nimg))
                    assert lib.jh_efftrack_launches(h) > 50
                    lib.jh_efftrack_destroy(h)
                    n += 1
    return n


# ---------------------------------------------------------------- whole predictors
def sweep_predictors():
    rng = random.Random(11)
    n = 0
    fixed = [  # the shipped geometries on scaled-down frames / rigs (the stand-in's "device" memory is host memory)
        dict(C=12, J=23, G=64, bbox=128, center=128, size="small", T=1, W=320, H=256),
        dict(C=2, J=23, G=72, bbox=320, center=320, size="medium", T=1, W=416, H=384),
        dict(C=4, J=23, G=72, bbox=192, center=192, size="medium", T=8, W=320, H=256),     # levels 48 / 24 / 12 / 6 / 3
        dict(C=16, J=30, G=96, bbox=128, center=128, size="small", T=1, W=320, H=256),
        dict(C=4, J=23, G=48, bbox=128, center=192, size="large", T=8, W=320, H=256),
    ]
    randoms = []
    for _ in range(2 if QUICK else 14):
        bbox = rng.choice([64, 128, 192] if QUICK else [64, 128, 192, 256, 320])
        T = rng.choice([1, 2, 8, 9])
        randoms.append(dict(C=rng.choice([2, 3, 4] if T >= 8 else [2, 3, 4, 6, 8, 12]),
                            J=rng.choice([1, 3, 8, 12, 23, 30, 40]), G=rng.choice(list(range(16, 100, 4))), bbox=bbox,
                            center=rng.choice([128, 192] if QUICK else [128, 192, 256, 320]),
                            size=rng.choice(["small", "small", "medium", "large"]), T=T,
                            W=max(bbox + 64, rng.choice([320, 416])), H=max(bbox + 32, rng.choice([256, 384]))))
    for c in (fixed[:3] if QUICK else fixed) + randoms:
        sd_c = rand_state(arch.efficienttrack_params(c["size"], 1), 5)
        sd_h = rand_state(arch.hybridnet_params(c["size"], c["J"]), 6)
        calib = S.ring_calibration(c["C"], c["W"], c["H"], 900.0)
        shards = [(0, None)]
        if c["C"] % 2 == 0:
            shards.append((c["C"] // 2, c["C"] // 2))      # a rank's camera share (distributed.py)
        for cam_lo, cam_n in shards:
            kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center"], bbox=c["bbox"],
                      roi_cube_size=float(c["G"] * 2), grid_spacing=2.0, img_h=c["H"], img_w=c["W"], mean=S.MEAN,
                      std=S.STD, center_model=c["size"], kp_model=c["size"], time_batch=c["T"], cam_lo=cam_lo,
                      cam_n=cam_n)
            try:
                p = NativePredictor(sd_c, sd_h, **kw)
            except RuntimeError as e:
                raise SystemExit("FAILED predictor_create %r: %s" % (c, e))
            Cl = p.Cloc
            ok(lib.jh_predictor_set_calibration(p.handle, calib[0].data_ptr(), calib[1].data_ptr(), calib[2].data_ptr(), None),
               "set_calibration")
            T, J = c["T"], c["J"]
            pts, conf, valid = torch.zeros((T, J, 3)), torch.zeros((T, J)), torch.zeros((T,), dtype=torch.int32)
            f32 = torch.zeros((T, Cl, 3, c["H"], c["W"]))
            u8 = torch.zeros((T, Cl, c["H"], c["W"], 3), dtype=torch.uint8)
            if cam_n is None:
                for graph in (0, 1):
                    ok(lib.jh_predictor_set_graph_replay(p.handle, graph), "graph_replay")
                    ok(lib.jh_predictor_forward(p.handle, f32.data_ptr(), pts.data_ptr(), conf.data_ptr(), valid.data_ptr(),
                                                None), "predictor_forward %r" % (c,))
                ok(lib.jh_predictor_forward_u8(p.handle, u8.data_ptr(), pts.data_ptr(), conf.data_ptr(), valid.data_ptr(),
                                               None), "predictor_forward_u8 %r" % (c,))
            else:
                det = torch.zeros((T, Cl, 3))
                det_all = torch.zeros((T, c["C"], 3))
                heat = torch.zeros((T, Cl, p.Hh, p.Hh, p.Jp))
                heat_all = torch.zeros((T, c["C"], p.Hh, p.Hh, p.Jp))
                # (the Python wrappers insist on CUDA tensors: the ABI is called directly)
                ok(lib.jh_predictor_stage_center(p.handle, f32.data_ptr(), det.data_ptr(), None), "stage_center")
                ok(lib.jh_predictor_stage_keypoints(p.handle, f32.data_ptr(), det_all.data_ptr(), heat.data_ptr(), None),
                   "stage_keypoints")
                ok(lib.jh_predictor_stage_3d(p.handle, heat_all.data_ptr(), 0, pts.data_ptr(), conf.data_ptr(),
                                             valid.data_ptr(), None), "stage_3d")
                ok(lib.jh_predictor_stage_keypoints_gathered(p.handle, f32.data_ptr(), 0, det_all.data_ptr(), 2,
                                                             heat.data_ptr(), None), "stage_keypoints_gathered")
                ok(lib.jh_predictor_stage_3d_blocks(p.handle, heat_all.data_ptr(), 2, T, 0, 0, pts.data_ptr(),
                                                    conf.data_ptr(), valid.data_ptr(), None), "stage_3d_blocks")
            p.close()
            n += 1
    return n


# ---------------------------------------------------------------- parameter sets
def sweep_params():
    h = ctypes.c_void_p()
    ok(lib.jh_params_create(ctypes.byref(h)), "params_create")
    for k, numel in ((b"a", 1), (b"a", 7), (b"some.very.long.key." * 20, 3), (b"", 5), (b"zero", 0)):
        t = torch.zeros((max(numel, 1),))
        rc = lib.jh_params_set(h, k, t.data_ptr(), numel)
        assert rc in (0, 1)
    lib.jh_params_destroy(h)
    # a network built from a parameter set that misses keys must refuse, not read out of bounds
    sd = rand_state(arch.v2v_params(5), 1)
    sd.pop(sorted(sd)[0])
    params = N.Params(sd)
    h = ctypes.c_void_p()
    assert lib.jh_v2v_create(params.handle, b"", 5, 1, 16, ctypes.byref(h)) != 0
    # ... and one whose tensor is too short
    sd = rand_state(arch.v2v_params(5), 1)
    k0 = sorted(sd)[0]
    sd[k0] = sd[k0].flatten()[:-1].contiguous()
    params = N.Params(sd)
    assert lib.jh_v2v_create(params.handle, b"", 5, 1, 16, ctypes.byref(h)) != 0
    return 3


if __name__ == "__main__":
    random.seed(5)
    counts = {}
    for name, fn in (("params", sweep_params), ("v2v", sweep_v2v), ("reproject", sweep_reproject),
                     ("efficienttrack", sweep_efftrack), ("predictors", sweep_predictors)):
        t0 = time.time()
        counts[name] = fn()
        print("%s: %d cases, %.1f s" % (name, counts[name], time.time() - t0), flush=True)
    rej = lib.jh_stub_rejected()
    print("launch configurations accepted %d, rejected %d; total %.1f s" % (lib.jh_stub_launches(), rej, time.time() - t00))
    for what, msg in refused:
        print("declined: %s -- %s" % (what, msg))
    if rej:
        raise SystemExit("launch configurations the hardware would refuse: %d" % rej)
    print("SANITIZER SWEEP OK")
