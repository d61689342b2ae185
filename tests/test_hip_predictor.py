"""GPU: HybridNetBackbone.forward and JarvisPredictor3D.forward (HIP) against
the CPU oracle and the reference's golden outputs, incl. the (None, None) path
and the time-batched throughput form."""
import os
from types import SimpleNamespace as NS

import pytest
import torch

from tests import cases
from tests.gpu_util import cuda, max_err, rel_err, report

pytestmark = pytest.mark.gpu


def make_cfg(c, center_size=256):
    from jarvis_hybridnet_amd import synthetic as S
    return NS(PARENT_DIR="/nonexistent", PROJECT_NAME="none",
              DATASET=NS(DATASET_ROOT_DIR="x", MEAN=S.MEAN, STD=S.STD),
              CENTERDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=1, IMAGE_SIZE=center_size),
              KEYPOINTDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=c["J"], BOUNDING_BOX_SIZE=c["bbox"]),
              HYBRIDNET=NS(NUM_CAMERAS=c["C"], ROI_CUBE_SIZE=c["roi"], GRID_SPACING=c["spacing"]))


@pytest.mark.parametrize("tag", ["cfg2", "cfg3", "cfg5"])
def test_hybridnet_backbone(tag, golden):
    from jarvis_hybridnet_amd.hybridnet.hybridnet import HybridNet
    c = cases.HYBRID_CASES[tag]
    inp = cases.hybrid_inputs(tag)
    net = HybridNet("inference", make_cfg(c), inp["sd_hybrid"]).model
    fin, hm, pts, conf = net(cuda(inp["crops"]), torch.tensor([c["W"], c["H"]]),
                             cuda(inp["center_hm"]), cuda(inp["center3d"]), cuda(inp["cam"]),
                             cuda(inp["intr"]), cuda(inp["dist"]))
    torch.cuda.synchronize()
    g = golden("hybridnet")
    rp, rc = torch.from_numpy(g[tag + ".points3D"]), torch.from_numpy(g[tag + ".confidences"])
    ep, ec = max_err(pts, rp), max_err(conf, rc)
    from tests.util import check_summary
    report("hybridnet", tag=tag, points_mm=ep, conf=ec)
    assert ep < 1e-3, "3D keypoints must be within 1e-3 mm of the reference"
    assert ec < 1e-4
    check_summary(g, tag + ".heatmaps_padded", hm.cpu(), rtol=1e-3, atol=0.05)
    check_summary(g, tag + ".heatmap_final", fin.cpu(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["cfg2", "cfg3", "cfg2_none"])
def test_predictor3d(tag, golden):
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    c = cases.PREDICTOR_CASES[tag]
    inp = cases.predictor_inputs(tag)
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    pts, conf = pred(cuda(inp["imgs"]), cuda(inp["cam"]), cuda(inp["intr"]), cuda(inp["dist"]))
    torch.cuda.synchronize()
    g = golden("predictor")
    dbg = pred.native(c["H"], c["W"]).debug("cuda")
    if c.get("expect_none"):
        assert pts is None and conf is None
        return
    # integer path: exact
    preds = torch.from_numpy(g[tag + ".preds"]).reshape(c["C"], 2)
    assert torch.equal(dbg["det"][0, :, :2].cpu().long(), preds)
    assert torch.equal(dbg["center_hm"][0].cpu(), torch.from_numpy(g[tag + ".center_hm"]))
    assert torch.equal(dbg["center3d_int"][0].cpu(),
                       torch.from_numpy(g[tag + ".center3d"]).int())
    e3 = (dbg["center3d"][0].cpu() - torch.from_numpy(g[tag + ".center3d"])).abs().max().item()
    ep = max_err(pts, torch.from_numpy(g[tag + ".points3D"]))
    ec = max_err(conf, torch.from_numpy(g[tag + ".confidences"]))
    report("predictor3d", tag=tag, points_mm=ep, conf=ec, center3d_mm=e3)
    assert ep < 1e-3, "3D keypoints must be within 1e-3 mm of the reference"
    assert ec < 1e-4


def test_predictor3d_time_batch():
    """T independent frames in one call == T single calls (bit-for-bit: the
    kernels treat the batch dimension as independent instances)."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    calib = (inp["cam"], inp["intr"], inp["dist"])
    frames = [inp["imgs"]] + [S.blob_frames(calib, c["W"], c["H"], c["J"], s)[0] for s in (60, 61)]
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    dev = [cuda(t) for t in calib]
    singles = [pred(cuda(f), *dev) for f in frames]
    pts, conf, valid = pred.forward_batch(cuda(torch.stack(frames)), *dev)
    torch.cuda.synchronize()
    for t, (p, q) in enumerate(singles):
        assert int(valid[t]) == (p is not None)
        if p is not None:
            assert (pts[t] - p[0]).abs().max().item() < 1e-4
            assert (conf[t] - q[0]).abs().max().item() < 1e-6


def test_sharded_stages_emulated_two_ranks():
    """The camera-sharded stage API on ONE GPU: two NativePredictors own half of
    the cameras each, the two exchanges of distributed.py are done by hand
    (concatenation = what the collectives deliver).  Result must equal the
    single-predictor forward bit-for-bit (sharding only moves data; InstanceNorm
    is per image, so per-camera batching is order independent)."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    calib = (inp["cam"], inp["intr"], inp["dist"])
    T, C, J, world = 2, c["C"], c["J"], 2
    frames = cuda(torch.stack([inp["imgs"], S.blob_frames(calib, c["W"], c["H"], J, 60)[0]]))
    common = dict(num_cameras=C, num_joints=J, center_size=c["center_size"], bbox=c["bbox"],
                  roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
                  mean=S.MEAN, std=S.STD, time_batch=T)
    dev = [cuda(t) for t in calib]
    full = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **common)
    full.set_calibration(*dev)
    rp, rc, rv = [t.clone() for t in full.forward(frames)]
    Cl, T3 = C // world, T // world
    ranks = []
    for r in range(world):
        p = NativePredictor(inp["sd_center"], inp["sd_hybrid"], time_batch_3d=T3, cam_lo=r * Cl,
                            cam_n=Cl, **common)
        p.set_calibration(*dev)
        ranks.append(p)
    local = [frames[:, r * Cl:(r + 1) * Cl].contiguous() for r in range(world)]
    dets = []
    for r, p in enumerate(ranks):
        d = torch.empty((T, Cl, 3), device="cuda")
        p.stage_center(local[r], d)
        dets.append(d)
    det_all = torch.cat(dets, 1).contiguous()                       # exchange 1
    heats = []
    for r, p in enumerate(ranks):
        h = torch.empty((T, Cl, p.Hh, p.Hh, p.Jp), device="cuda")
        p.stage_keypoints(local[r], det_all, h)
        heats.append(h)
    heat_all = torch.cat(heats, 1).contiguous()                     # exchange 2
    for r, p in enumerate(ranks):
        pts = torch.empty((T3, J, 3), device="cuda")
        conf = torch.empty((T3, J), device="cuda")
        valid = torch.empty((T3,), device="cuda", dtype=torch.int32)
        p.stage_3d(heat_all[r * T3:(r + 1) * T3].contiguous(), r * T3, pts, conf, valid)
        torch.cuda.synchronize()
        assert torch.equal(valid, rv[r * T3:(r + 1) * T3])
        ep = (pts - rp[r * T3:(r + 1) * T3]).abs().max().item()
        ec = (conf - rc[r * T3:(r + 1) * T3]).abs().max().item()
        report("sharded_emulated", rank=r, points_mm=ep, conf=ec)
        assert ep < 1e-4 and ec < 1e-6


def test_predictor3d_uint8_ingest():
    """frames as uint8 BGR (C,H,W,3) straight from the decoder.
    Must equal the fp32 path fed the reference driver's conversion
    (predict3D.py:79-80) bit for bit -- the conversion is the same arithmetic, only
    done inside the resize / crop kernels."""
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    u8 = (inp["imgs"].permute(0, 2, 3, 1)[..., [2, 1, 0]] * 255).round().to(torch.uint8).contiguous()
    ref_in = cuda(u8).float().permute(0, 3, 1, 2)[:, [2, 1, 0]] / 255.
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    dev = [cuda(inp[k]) for k in ("cam", "intr", "dist")]
    p0, c0 = pred(ref_in.contiguous(), *dev)
    p1, c1 = pred.forward_uint8(cuda(u8), *dev)
    torch.cuda.synchronize()
    assert p0 is not None and p1 is not None
    assert torch.equal(p0, p1) and torch.equal(c0, c1)
    pts, conf, valid = pred.forward_batch(cuda(torch.stack([u8, u8])), *dev)
    torch.cuda.synchronize()
    assert int(valid.sum()) == 2 and (pts[1] - p0[0]).abs().max().item() < 1e-4


@pytest.mark.parametrize("tag", list(cases.PREDICTOR2D_CASES))
def test_predictor2d(tag, golden):
    """JarvisPredictor2D.forward vs the reference's fixtures.
    points2D is an integer path (argmax indices): bit-exact."""
    from jarvis_hybridnet_amd.prediction.jarvis2D import JarvisPredictor2D
    c = cases.PREDICTOR2D_CASES[tag]
    inp = cases.predictor2d_inputs(tag)
    cfg = make_cfg(dict(J=c["J"], bbox=c["bbox"], C=1, roi=32, spacing=2), c["center_size"])
    pred = JarvisPredictor2D(cfg, inp["sd_center"], inp["sd_kp"])
    pts, conf = pred(cuda(inp["img"]))
    torch.cuda.synchronize()
    g = golden("predictor2d")
    if c.get("expect_none"):
        assert pts is None and conf is None
        return
    assert torch.equal(pts.cpu(), torch.from_numpy(g[tag + ".points2D"]))
    ec = max_err(conf, torch.from_numpy(g[tag + ".confidences"]))
    report("predictor2d", tag=tag, conf=ec)
    assert ec < 1e-5
    # batch form: two copies of the image -> two identical rows
    p2, c2, v2 = pred.forward_batch(cuda(torch.cat([inp["img"], inp["img"]])))
    torch.cuda.synchronize()
    assert int(v2.sum()) == 2 and torch.equal(p2[0], p2[1]) and torch.equal(p2[0].long().cpu(), pts.cpu())


def test_predict3d_frames_writes_csv(tmp_path, golden):
    """frames in (uint8 BGR as decoded, and fp32), data3D.csv out."""
    import csv
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    from jarvis_hybridnet_amd.prediction.predict3D import predict3D_frames
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    cfg = make_cfg(c, c["center_size"])
    cfg.KEYPOINT_NAMES = ["k%d" % i for i in range(c["J"])]
    pred = JarvisPredictor3D(cfg, inp["sd_center"], inp["sd_hybrid"])
    u8 = (inp["imgs"].permute(0, 2, 3, 1)[..., [2, 1, 0]] * 255).round().to(torch.uint8).numpy()
    dev = [cuda(inp[k]) for k in ("cam", "intr", "dist")]
    n = predict3D_frames(pred, [inp["imgs"], u8, u8], *dev, cfg,
                         str(tmp_path), NS(recording_path="r", dataset_name="d", frame_start=0,
                                           number_frames=3))
    rows = list(csv.reader(open(tmp_path / "data3D.csv")))
    assert n == 3 and len(rows) == 5 and len(rows[2]) == 4 * c["J"]
    assert rows[0][:4] == ["k0"] * 4 and rows[1][:4] == ["x", "y", "z", "confidence"]
    gold = torch.from_numpy(golden("predictor")["cfg2.points3D"])[0]
    got = torch.tensor([float(v) for v in rows[2]]).view(c["J"], 4)
    assert (got[:, :3] - gold).abs().max() < 1e-3
    got8 = torch.tensor([float(v) for v in rows[3]]).view(c["J"], 4)
    assert (got8[:, :3] - gold).abs().max() < 20.0    # 8-bit quantised input, random-weight nets
    assert rows[3] == rows[4]
    assert os.path.isfile(tmp_path / "info.yaml")
    # throughput form: groups of 2 frame sets per launch sequence (short last group padded)
    n = predict3D_frames(pred, [u8, u8, u8], *dev, cfg, str(tmp_path / "tb"), time_batch=2)
    rows2 = list(csv.reader(open(tmp_path / "tb" / "data3D.csv")))
    assert n == 3 and len(rows2) == 5 and rows2[2] == rows2[3] == rows2[4] == rows[3]
    # ... and with three such groups in flight on three HIP streams (7 frame sets: 4 groups)
    n = predict3D_frames(pred, [u8] * 7, *dev, cfg, str(tmp_path / "ms"), time_batch=2, streams=3)
    rows3 = list(csv.reader(open(tmp_path / "ms" / "data3D.csv")))
    assert n == 7 and len(rows3) == 9 and all(r == rows[3] for r in rows3[2:])


def test_error_paths_are_loud():
    """No silent fallbacks: bad inputs surface as exceptions carrying the library's message
    (jh_last_error), as the reference's own failures surface as exceptions."""
    from jarvis_hybridnet_amd import _native as N
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd.efficienttrack.model import EfficientTrackBackbone
    from jarvis_hybridnet_amd.hybridnet.v2vnet import V2VNet
    # image side that is not a multiple of 64
    net = EfficientTrackBackbone(None, "small", 3)
    net.load_state_dict(S.efficienttrack_weights("small", 3, 1), strict=True)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        net(torch.zeros(1, 3, 100, 100, device="cuda"))
    # CPU tensors are rejected, not copied
    with pytest.raises(RuntimeError, match="needs CUDA"):
        net(torch.zeros(1, 3, 128, 128))
    # a state dict with a missing key
    sd = S.v2v_weights(3, 2)
    sd.pop(sorted(sd)[0])
    v = V2VNet(3, 3)
    with pytest.raises((RuntimeError, KeyError)):
        v.load_state_dict(sd, strict=True)
        v(torch.zeros(1, 3, 16, 16, 16, device="cuda"))
    # grid side that V2V cannot halve twice
    v2 = V2VNet(3, 3)
    v2.load_state_dict(S.v2v_weights(3, 2), strict=True)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        v2(torch.zeros(1, 3, 18, 18, 18, device="cuda"))
    # after an error the library keeps working
    y = v2(torch.rand(1, 3, 16, 16, 16, device="cuda"))
    torch.cuda.synchronize()
    assert y.shape == (1, 3, 8, 8, 8) and bool(torch.isfinite(y).all())
    assert N.lib().jh_abi_version() >= 1


def test_multi_stream_predictor_equals_single_stream():
    """MultiStreamPredictor (K independent time batches in flight on K HIP streams, one plan
    set each) returns, batch for batch, what a single predictor returns."""
    from jarvis_hybridnet_amd._predictor import MultiStreamPredictor, NativePredictor
    from jarvis_hybridnet_amd import synthetic as S
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center_size"], bbox=c["bbox"],
              roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
              mean=S.MEAN, std=S.STD, time_batch=2)
    calib = [cuda(inp[k]) for k in ("cam", "intr", "dist")]
    single = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
    single.set_calibration(*calib)
    msp = MultiStreamPredictor(lambda: NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw), streams=3)
    msp.set_calibration(*calib)
    imgs = cuda(inp["imgs"])
    batches = [torch.stack([imgs, imgs.flip(0)]), torch.stack([imgs.flip(0), imgs]),
               torch.stack([imgs, imgs]), torch.stack([imgs.roll(1, 0), imgs])]
    got = [msp.forward(b) for b in batches]            # 4 batches over 3 streams: stream 0 is reused
    msp.synchronize()
    got = [[t.clone() for t in g] for g in got[1:]]    # batch 0's buffers were not reused (own `out`)
    for b, g in zip(batches[1:], got):
        ref = single.forward(b)
        torch.cuda.synchronize()
        for x, y in zip(g, ref):
            assert torch.equal(x, y)
