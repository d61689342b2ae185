"""Generate the committed golden vectors (BUILD CONTAINER ONLY).

Imports the upstream reference from /root/reference through the stub recipe
in `_ref_import.py`, runs it on the seeded synthetic inputs of
`jarvis_hybridnet_amd.synthetic`, checks that the repo's CPU oracle reproduces
every reference tensor BIT-FOR-BIT, and writes small `.npz` fixtures (full
tensors where small, strided samples + float64 checksums where large).

    python tests/golden/make_golden.py [case ...]
    GOLDEN_TAGS=ex72,cfg3_medium python tests/golden/make_golden.py predictor   # only these tags,
                                              # merged into the existing .npz / .json of the case

The fixtures are data (inputs are re-derivable from seeds; outputs are the
reference's).  No reference source is copied into the repository.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as R  # noqa: E402

R.install()
import torch  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

from jarvis.efficienttrack.efficienttrack import EfficientTrack  # noqa: E402
from jarvis.efficienttrack.model import EfficientTrackBackbone  # noqa: E402
from jarvis.hybridnet.hybridnet import HybridNet  # noqa: E402
from jarvis.hybridnet.model import HybridNetBackbone  # noqa: E402
from jarvis.hybridnet.repro_layer import ReprojectionLayer  # noqa: E402
from jarvis.hybridnet.v2vnet import V2VNet  # noqa: E402
from jarvis.prediction.jarvis3D import JarvisPredictor3D  # noqa: E402
from jarvis.utils.reprojection import ReprojectionTool  # noqa: E402

from jarvis_hybridnet_amd import synthetic as S  # noqa: E402
from oracle import hybridnet_oracle as O  # noqa: E402
from tests import cases  # noqa: E402


def summary(t):
    """Strided sample + float64 checksums of a large tensor."""
    a = t.detach().double().numpy()
    flat = a.reshape(-1)
    step = max(1, flat.size // 4096)
    return dict(sample=flat[::step].astype(np.float32 if t.is_floating_point()
                                           else np.int64),
                step=np.int64(step), sum=np.float64(flat.sum()),
                abssum=np.float64(np.abs(flat).sum()),
                shape=np.array(t.shape, dtype=np.int64))


def put(out, name, t, full):
    if full:
        out[name] = t.detach().numpy()
    else:
        for k, v in summary(t).items():
            out[name + "." + k] = v


def must_equal(a, b, what):
    if not torch.equal(a, b):
        d = (a.double() - b.double()).abs().max().item()
        raise SystemExit("oracle != reference for %s (max abs diff %g)" % (what, d))


ONLY = set(filter(None, os.environ.get("GOLDEN_TAGS", "").split(",")))


def tags(table):
    """The cases of `table` to (re)generate: all, or those named in GOLDEN_TAGS."""
    return {k: v for k, v in table.items() if not ONLY or k in ONLY}.items()


def merge_json(name, new):
    path = os.path.join(HERE, name)
    if ONLY and os.path.exists(path):
        old = json.load(open(path))
        old.update(new)
        new = old
    with open(path, "w") as f:
        json.dump(new, f, indent=1)


def save(name, out):
    path = os.path.join(HERE, name + ".npz")
    if ONLY and os.path.exists(path):
        old = dict(np.load(path))
        old.update(out)
        out = old
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------

def case_efficienttrack():
    out = {}
    for tag, (size, J, N, hw, wseed, xseed) in cases.EFFTRACK_CASES.items():
        sd = S.efficienttrack_weights(size, J, wseed)
        x = cases.efftrack_input(N, hw, xseed)
        ref = EfficientTrackBackbone(None, size, J).eval()
        ref.load_state_dict(sd, strict=True)
        with torch.no_grad():
            r1, r2 = ref(x)
            o1, o2 = O.efficienttrack_forward(sd, x, size)
        must_equal(r1, o1, tag + ".res1")
        must_equal(r2, o2, tag + ".res2")
        full = r2.numel() <= 1 << 18
        put(out, tag + ".res1", r1, full)
        put(out, tag + ".res2", r2, full)
        print(tag, "ok", tuple(r2.shape), float(r2.abs().max()))
    save("efficienttrack", out)


def case_state_spec():
    """Key/shape layout of the reference modules (data, not source)."""
    spec = {}
    for size in ("small", "medium", "large"):
        for J in (1, 23):
            m = EfficientTrackBackbone(None, size, J)
            spec["efficienttrack.%s.%d" % (size, J)] = [
                [k, list(v.shape)] for k, v in m.state_dict().items()]
    cfg = R.make_cfg(num_cameras=4, num_joints=23, roi=32, spacing=2)
    m = HybridNetBackbone(cfg)
    spec["hybridnet.small.23"] = [[k, list(v.shape)]
                                  for k, v in m.state_dict().items()]
    for size, J in (("small", 23),):
        assert [[k, list(s)] for k, s in O.hybridnet_state_spec(size, J)] == \
            spec["hybridnet.small.23"]
    with open(os.path.join(HERE, "state_spec.json"), "w") as f:
        json.dump(spec, f, indent=0)
    print("wrote state_spec.json")


def case_reprojection():
    out = {}
    for tag, (C, J, G, spacing, bbox, W, H, focal, seed) in tags(cases.REPRO_CASES):
        cfg = R.make_cfg(num_cameras=C, num_joints=J, roi=G * spacing,
                         spacing=spacing, bbox=bbox)
        inp = cases.repro_inputs(tag)
        layer = ReprojectionLayer(cfg)
        hs = layer.heatmap_size
        with torch.no_grad():
            grid = layer.grid + inp["center3d"][0]
            ref_idx = layer.reprojectPoints(grid, inp["cam"][0], inp["intr"][0],
                                            inp["dist"][0], inp["center_hm"][0])
            t = time.time()
            if J * C * G ** 3 <= 4e8:
                ref_vol = layer(inp["hm_pad"], inp["center3d"], inp["center_hm"],
                                inp["cam"], inp["intr"], inp["dist"])
            else:   # too large to materialise in one go: run per joint block
                ref_vol = torch.cat([
                    layer(inp["hm_pad"][:, :, j:j + 5], inp["center3d"],
                          inp["center_hm"], inp["cam"], inp["intr"], inp["dist"])
                    for j in range(0, J, 5)], 1)
            print(tag, "reference layer %.2fs" % (time.time() - t))
            ovol, oidx = O.reprojection_forward(
                inp["hm_pad"], inp["center3d"], inp["center_hm"], inp["cam"],
                inp["intr"], inp["dist"], G * spacing, spacing, chunk=5,
                return_idx=True)
        must_equal(ref_idx, oidx, tag + ".idx")
        must_equal(ref_vol, ovol, tag + ".vol")
        assert int(ref_idx.max()) < hs * hs and int(ref_idx.min()) >= 0
        full = ref_vol.numel() <= 1 << 16
        put(out, tag + ".idx", ref_idx, full)
        put(out, tag + ".vol", ref_vol, full)
        if not full and ref_idx.numel() <= 4e6:
            # the complete integer index field, delta-coded along the last axis
            a = ref_idx.numpy().astype(np.int16)
            out[tag + ".idx_delta16"] = np.diff(a, axis=-1, prepend=np.int16(0)).astype(np.int16)
        frac = float((ref_vol > 1).float().mean())
        print(tag, "ok", tuple(ref_vol.shape), "max %.1f  frac>1 %.3f" %
              (float(ref_vol.max()), frac))
        assert frac > 0.01, "degenerate reprojection case"
        if tag.endswith("_edge"):
            # the crop clamp is active: the subject's true projection lies outside the clamped crop centre's
            # reach for several cameras, and a large share of those cameras' voxels take a border index
            uv = torch.from_numpy(S.project(cases.subject_geometry(C, W, H, focal, bbox, seed)[3][None]
                                            .double().numpy(), inp["cam"][0], inp["intr"][0], inp["dist"][0]))[:, 0]
            chm = inp["center_hm"][0]
            nx = int((uv[:, 0].int() != chm[:, 0]).sum())
            ny = int((uv[:, 1].int() != chm[:, 1]).sum())
            col, row = ref_idx % hs, ref_idx // hs
            border = ((col == 0) | (col == hs - 1) | (row == 0) | (row == hs - 1)).float().flatten(1).mean(1)
            print(tag, "cameras clamped in x %d, in y %d; per-camera share of border indices" % (nx, ny),
                  [round(float(b), 3) for b in border])
            assert nx >= 2 and ny >= 1 and float(border.max()) > 0.3
            out[tag + ".clamped_xy"] = np.array([nx, ny], dtype=np.int64)
            out[tag + ".border_share"] = border.numpy()
    save("reprojection", out)


def case_reprojection_hash():
    """64-bit hash of every camera plane of the reference's gather-index field
    (ReprojectionLayer.reprojectPoints) for all reprojection cases, incl. the 14 M indices of
    cfg5 that are too many to commit in full (tests/util.py::index_plane_hashes)."""
    from tests.util import index_plane_hashes
    out = {}
    for tag, (C, J, G, spacing, bbox, W, H, focal, seed) in tags(cases.REPRO_CASES):
        cfg = R.make_cfg(num_cameras=C, num_joints=J, roi=G * spacing, spacing=spacing, bbox=bbox)
        inp = cases.repro_inputs(tag)
        layer = ReprojectionLayer(cfg)
        with torch.no_grad():
            grid = layer.grid + inp["center3d"][0]
            ref_idx = layer.reprojectPoints(grid, inp["cam"][0], inp["intr"][0], inp["dist"][0],
                                            inp["center_hm"][0])
        out[tag] = dict(n=int(ref_idx.numel()), planes=index_plane_hashes(ref_idx))
        print(tag, tuple(ref_idx.shape), out[tag]["planes"][:2])
    merge_json("reprojection_index_hashes.json", out)


def case_v2v():
    out = {}
    for tag, (J, G, wseed, xseed) in tags(cases.V2V_CASES):
        sd = S.v2v_weights(J, wseed)
        x = cases.v2v_input(J, G, xseed)
        ref = V2VNet(J, J).eval()
        ref.load_state_dict(sd, strict=True)
        with torch.no_grad():
            r = ref(x)
            o = O.v2v_forward(sd, x)
        must_equal(r, o, tag)
        put(out, tag + ".out", r, r.numel() <= 1 << 18)
        # tail on the same tensor (hybridnet/model.py:73-88)
        cfg = R.make_cfg(num_cameras=2, num_joints=J, roi=G * 2, spacing=2)
        hb = HybridNetBackbone.__new__(HybridNetBackbone)
        torch.nn.Module.__init__(hb)
        hb.grid_spacing = torch.tensor(cfg.HYBRIDNET.GRID_SPACING)
        hb.grid_size = torch.tensor(cfg.HYBRIDNET.ROI_CUBE_SIZE)
        hb.softplus = torch.nn.Softplus()
        n = int(hb.grid_size / hb.grid_spacing / 2)
        hb.xx, hb.yy, hb.zz = torch.meshgrid(torch.arange(n), torch.arange(n),
                                             torch.arange(n), indexing="ij")
        center = torch.tensor([[35, -58, 549]], dtype=torch.int32)
        # run the tail lines of the reference forward by calling it with the
        # front half replaced: effTrack / reproLayer / v2vNet stubs
        hb.effTrack = lambda im: (None, torch.zeros(2, J, 4, 4))
        hb.reproLayer = lambda *a: x * 255.
        hb.v2vNet = lambda v: r
        fin, _, pts, conf = HybridNetBackbone.forward(
            hb, torch.zeros(1, 2, 3, 8, 8), torch.tensor([8, 8]), None, center,
            None, None, None)
        ofin, opts, oconf = O.softargmax_tail(r, center, G * 2, 2)
        must_equal(fin, ofin, tag + ".final")
        must_equal(pts, opts, tag + ".points")
        must_equal(conf, oconf, tag + ".conf")
        out[tag + ".points"] = pts.numpy()
        out[tag + ".conf"] = conf.numpy()
        put(out, tag + ".final", fin, False)
        print(tag, "ok", tuple(r.shape), "conf", float(conf.min()), float(conf.max()))
    save("v2v", out)


def case_geometry():
    out = {}
    for tag, (C, W, H, focal, seed) in cases.GEOM_CASES.items():
        cam, intr, dist = S.ring_calibration(C, W, H, focal)
        pts2d, maxvals, p3d = cases.geom_inputs(tag)
        tool = ReprojectionTool()
        tool.device = "cpu"
        tool.cameraMatrices, tool.intrinsicMatrices = cam, intr
        tool.distortionCoefficients = dist
        with torch.no_grad():
            r_rec = tool.reconstructPoint(pts2d.clone(), maxvals)
            r_rep = tool.reprojectPoint(p3d.clone())
            o_rec = O.reconstruct_point(pts2d.clone(), maxvals, cam, intr, dist)
            o_rep = O.reproject_point(p3d.clone(), cam, intr, dist)
        must_equal(r_rec, o_rec, tag + ".reconstruct")
        must_equal(r_rep, o_rep, tag + ".reproject")
        out[tag + ".reconstruct"] = r_rec.numpy()
        out[tag + ".reproject"] = r_rep.numpy()
        print(tag, "ok", r_rec.tolist())
    save("geometry", out)


def _predictor(cfg, sd_center, sd_hybrid, tmp):
    pc, ph = os.path.join(tmp, "c.pth"), os.path.join(tmp, "h.pth")
    torch.save(sd_center, pc)
    torch.save(sd_hybrid, ph)
    pred = JarvisPredictor3D.__new__(JarvisPredictor3D)
    torch.nn.Module.__init__(pred)
    # the reference constructor (jarvis3D.py:20-46) with trt_mode='off'
    JarvisPredictor3D.__init__(pred, cfg, pc, ph, "off")
    return pred


def case_predictor():
    out = {}
    meta = {}
    for tag, c in tags(cases.PREDICTOR_CASES):
        size = c.get("size", "small")
        cfg = R.make_cfg(num_cameras=c["C"], num_joints=c["J"], roi=c["roi"],
                         spacing=c["spacing"], bbox=c["bbox"],
                         center_size=c["center_size"], center_model=size, kp_model=size)
        inp = cases.predictor_inputs(tag)
        with tempfile.TemporaryDirectory() as tmp:
            pred = _predictor(cfg, inp["sd_center"], inp["sd_hybrid"], tmp)
        # capture the integer path by wrapping hybridNet.forward
        seen = {}
        hyb_fwd = pred.hybridNet.forward

        def spy(imgs, img_size, centerHM, center3D, camM, K, D):
            seen["center_hm"] = centerHM[0].clone()
            seen["center3d_int"] = center3D[0].clone()
            res = hyb_fwd(imgs, img_size, centerHM, center3D, camM, K, D)
            seen["heatmaps_padded"] = res[1]
            seen["heatmap_final"] = res[0]
            return res
        pred.hybridNet.forward = spy
        t = time.time()
        with torch.no_grad():
            pts, conf = pred(inp["imgs"], inp["cam"], inp["intr"], inp["dist"])
        print(tag, "reference forward %.1fs" % (time.time() - t))
        inter = {}
        with torch.no_grad():
            opts, oconf = O.predictor3d_forward(
                inp["sd_center"], inp["sd_hybrid"], inp["imgs"], inp["cam"],
                inp["intr"], inp["dist"], center_size=c["center_size"],
                bbox=c["bbox"], roi_cube_size=c["roi"],
                grid_spacing=c["spacing"], mean=S.MEAN, std=S.STD, chunk=5,
                center_model=size, kp_model=size, intermediates=inter)
        if "n_detect" in c:
            assert inter["n_detect"] == c["n_detect"], (tag, inter["n_detect"])
        if c.get("expect_none"):
            assert pts is None and conf is None and opts is None
            out[tag + ".none"] = np.int64(1)
            out[tag + ".n_detect"] = np.int64(inter["n_detect"])
            out[tag + ".maxvals"] = inter["maxvals"].numpy()
            meta[tag] = dict(n_detect=inter["n_detect"],
                             maxvals_255=[round(float(v) * 255, 3) for v in inter["maxvals"].flatten()])
            print(tag, "ok (None path, n_detect=%d)" % inter["n_detect"])
            continue
        must_equal(pts, opts, tag + ".points3D")
        must_equal(conf, oconf, tag + ".confidences")
        must_equal(seen["center_hm"], inter["center_hm"], tag + ".center_hm")
        must_equal(seen["center3d_int"], inter["center3d"].int(), tag + ".c3d")
        must_equal(seen["heatmaps_padded"], inter["heatmaps_padded"], tag + ".hm")
        # margins that make the integer path robust to last-bit differences
        hm = inter["center_heatmap"]
        flat = hm.view(hm.shape[0], -1)
        top2 = flat.topk(2, dim=1)[0]
        margin = ((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs()).min().item()
        c3 = inter["center3d"]
        frac3 = (c3 - c3.trunc()).abs()
        m3 = torch.minimum(frac3, 1 - frac3).min().item()
        rp = O.reproject_point(c3.unsqueeze(0), inp["cam"], inp["intr"], inp["dist"])
        fr = (rp - rp.trunc()).abs()
        mr = torch.minimum(fr, 1 - fr).min().item()
        # which cameras the crop clamp (jarvis3D.py:163-166) moved: the reference's own centerHMs against the
        # unclamped integer reprojection of its centre
        raw = rp.int()
        hw = c["bbox"] // 2
        clamps = (int((raw[:, 0] < hw).sum()), int((raw[:, 0] > c["W"] - hw).sum()),
                  int((raw[:, 1] < hw).sum()), int((raw[:, 1] > c["H"] - hw).sum()))
        assert int((raw != seen["center_hm"]).any(1).sum()) == int(((raw[:, 0] < hw) | (raw[:, 0] > c["W"] - hw) |
                                                                  (raw[:, 1] < hw) | (raw[:, 1] > c["H"] - hw)).sum())
        if "clamps" in c:
            assert clamps == tuple(c["clamps"]), (tag, clamps)
        # how far the reference's float32 SVD is from the exact least-squares centre on ITS OWN inputs (the same
        # restatement in float64): the noise level any other correct solver differs by.  The integer paths downstream
        # (center3D.int(), centerHMs) are only meaningful to pin when their margins clear it.
        scale = torch.tensor([c["W"] / float(c["center_size"]), c["H"] / float(c["center_size"])]).float()
        p2 = (inter["preds"].reshape(c["C"], 2) * (scale * 2)).transpose(0, 1)
        c64 = O.reconstruct_point(p2.double(), inter["maxvals"].double(), inp["cam"].double(), inp["intr"].double(),
                                  inp["dist"].double())
        noise3 = (c3.double() - c64).abs().max().item()
        noise_px = (rp - O.reproject_point(c64.float().unsqueeze(0), inp["cam"], inp["intr"], inp["dist"])).abs().max().item()
        assert m3 > 2 * noise3 and mr > 2 * noise_px, \
            "ill-conditioned case: integer margins %.4f mm / %.4f px within the float32 SVD's own noise %.4f mm / %.4f px" \
            % (m3, mr, noise3, noise_px)
        meta[tag] = dict(argmax_margin=margin, center3d_int_margin=m3,
                         center_hm_int_margin=mr, n_detect=inter["n_detect"],
                         svd_noise_mm=noise3, svd_noise_px=noise_px,
                         center3d=c3.tolist(), clamped_xlo_xhi_ylo_yhi=list(clamps),
                         maxvals_255=[round(float(v) * 255, 3) for v in inter["maxvals"].flatten()])
        print(tag, "margins", meta[tag])
        assert margin > 1e-3 and m3 > 2e-3 and mr > 2e-3, "fragile case: change seed"
        assert float(c3.abs().max()) < 1e4
        out[tag + ".points3D"] = pts.numpy()
        out[tag + ".confidences"] = conf.numpy()
        out[tag + ".preds"] = inter["preds"].numpy()
        out[tag + ".maxvals"] = inter["maxvals"].numpy()
        out[tag + ".n_detect"] = np.int64(inter["n_detect"])
        out[tag + ".center3d"] = inter["center3d"].numpy()
        out[tag + ".center_hm"] = inter["center_hm"].numpy()
        put(out, tag + ".heatmaps_padded", seen["heatmaps_padded"], False)
        put(out, tag + ".heatmap_final", seen["heatmap_final"], False)
        print(tag, "ok", pts[0, :2].tolist(), conf[0, :3].tolist())
    save("predictor", out)
    merge_json("predictor_meta.json", meta)


def case_hybridnet():
    """HybridNetBackbone.forward with geometrically meaningful centres."""
    out = {}
    for tag, c in tags(cases.HYBRID_CASES):
        cfg = R.make_cfg(num_cameras=c["C"], num_joints=c["J"], roi=c["roi"],
                         spacing=c["spacing"], bbox=c["bbox"])
        inp = cases.hybrid_inputs(tag)
        with tempfile.TemporaryDirectory() as tmp:
            ph = os.path.join(tmp, "h.pth")
            torch.save(inp["sd_hybrid"], ph)
            net = HybridNet("inference", cfg, ph).model
        with torch.no_grad():
            rf, rh, rp, rc = net(inp["crops"], torch.tensor([c["W"], c["H"]]),
                                 inp["center_hm"], inp["center3d"], inp["cam"],
                                 inp["intr"], inp["dist"])
            of, oh, op, oc = O.hybridnet_forward(
                inp["sd_hybrid"], "small", c["roi"], c["spacing"], inp["crops"],
                inp["center_hm"], inp["center3d"], inp["cam"], inp["intr"],
                inp["dist"], chunk=5)
        for a, b, n in ((rf, of, "final"), (rh, oh, "hm"), (rp, op, "pts"),
                        (rc, oc, "conf")):
            must_equal(a, b, tag + "." + n)
        out[tag + ".points3D"] = rp.numpy()
        out[tag + ".confidences"] = rc.numpy()
        put(out, tag + ".heatmap_final", rf, False)
        put(out, tag + ".heatmaps_padded", rh, False)
        print(tag, "ok", rp[0, 0].tolist(), rc[0, :3].tolist())
    save("hybridnet", out)


def case_predictor2d():
    from jarvis.prediction.jarvis2D import JarvisPredictor2D
    out, meta = {}, {}
    for tag, c in cases.PREDICTOR2D_CASES.items():
        cfg = R.make_cfg(num_cameras=1, num_joints=c["J"], bbox=c["bbox"],
                         center_size=c["center_size"])
        inp = cases.predictor2d_inputs(tag)
        with tempfile.TemporaryDirectory() as tmp:
            pc, pk = os.path.join(tmp, "c.pth"), os.path.join(tmp, "k.pth")
            torch.save(inp["sd_center"], pc)
            torch.save(inp["sd_kp"], pk)
            pred = JarvisPredictor2D(cfg, pc, pk, "off")
        with torch.no_grad():
            pts, conf = pred(inp["img"])
            inter = {}
            opts, oconf = O.predictor2d_forward(
                inp["sd_center"], inp["sd_kp"], inp["img"], center_size=c["center_size"],
                bbox=c["bbox"], mean=S.MEAN, std=S.STD, intermediates=inter)
        if c.get("expect_none"):
            assert pts is None and opts is None
            out[tag + ".none"] = np.int64(1)
            print(tag, "ok (None path)")
            continue
        must_equal(pts, opts, tag + ".points2D")
        must_equal(conf, oconf, tag + ".confidences")
        kh = inter["kp_heatmap"]
        top2 = kh.flatten(2).topk(2, dim=2)[0][0]
        margin = ((top2[:, 0] - top2[:, 1]) / top2[:, 0].abs()).min().item()
        meta[tag] = dict(kp_argmax_margin=margin, center_hm=inter["center_hm"].tolist())
        assert margin > 1e-3, "fragile keypoint argmax: change seed"
        out[tag + ".points2D"] = pts.numpy()
        out[tag + ".confidences"] = conf.numpy()
        out[tag + ".center_hm"] = inter["center_hm"].numpy()
        print(tag, "ok", pts[:3].tolist(), conf[:3].tolist(), "margin %.4f" % margin)
    save("predictor2d", out)
    with open(os.path.join(HERE, "predictor2d_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def case_calibration():
    """Calibration files -> ReprojectionTool tensors (utils/reprojection.py:16-46,93-111).
    cv2 is absent here, so the reference's TorchCamera.get_mat_from_file is served by an
    independent parser (PyYAML with an !!opencv-matrix constructor); everything after
    the file read is the reference's own code."""
    import yaml
    from jarvis.utils import reprojection as RP

    class Loader(yaml.SafeLoader):
        pass

    def opencv_matrix(loader, node):
        m = loader.construct_mapping(node, deep=True)
        return np.array(m["data"], dtype=np.float64).reshape(m["rows"], m["cols"])
    Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", opencv_matrix)

    def get_mat(self, filepath, node):
        text = open(filepath).read().replace("%YAML:1.0", "%YAML 1.1", 1)
        return yaml.load(text, Loader=Loader)[node]
    RP.TorchCamera.get_mat_from_file = get_mat

    cdir = os.path.join(HERE, "calib")
    os.makedirs(cdir, exist_ok=True)
    names = ["Camera_A", "Camera_B", "Camera_C", "Camera_D"]
    for name, (Rm, T, Kt, dist) in zip(names, S.ring_cameras(4, 640, 512, 900.0)):
        S.write_opencv_yaml(os.path.join(cdir, name + ".yaml"), Rm, T, Kt, dist)
    tool = RP.ReprojectionTool(cdir, {n: n + ".yaml" for n in names}, "cpu")
    out = dict(cameraMatrices=tool.cameraMatrices.numpy(),
               intrinsicMatrices=tool.intrinsicMatrices.numpy(),
               distortionCoefficients=tool.distortionCoefficients.numpy())
    # the files describe the same rig as synthetic.ring_calibration
    cam, intr, dist = S.ring_calibration(4, 640, 512, 900.0)
    assert (tool.cameraMatrices - cam).abs().max() < 1e-2
    if os.path.isdir("/root/reference/datasets/Example_Dataset/calib_params/12Cam_Ralph"):
        # sanity on the reference's real 12-camera calibration (files do not travel)
        rd = "/root/reference/datasets/Example_Dataset/calib_params/12Cam_Ralph"
        files = sorted(f for f in os.listdir(rd) if f.endswith(".yaml"))
        real = RP.ReprojectionTool(rd, {f[:-5]: f for f in files}, "cpu")
        from jarvis_hybridnet_amd.utils.reprojection import ReprojectionTool as Mine
        mine = Mine(rd, {f[:-5]: f for f in files}, "cpu")
        for a, b in ((real.cameraMatrices, mine.cameraMatrices),
                     (real.intrinsicMatrices, mine.intrinsicMatrices),
                     (real.distortionCoefficients, mine.distortionCoefficients)):
            must_equal(a, b, "real 12-camera calibration")
        print("real 12-camera calibration: loader == reference (%d cameras)" % len(files))
    save("calibration", out)


def case_csv():
    """data3D.csv wire format (prediction/predict3D.py:64-70,87-97,141-146)."""
    import csv
    import io
    from jarvis.prediction.predict3D import create_header
    g = dict(np.load(os.path.join(HERE, "predictor.npz")))
    cfg = R.ns(KEYPOINT_NAMES=["joint%d" % i for i in range(23)])
    buf = io.StringIO()
    writer = csv.writer(buf, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
    create_header(writer, cfg)
    pts = torch.from_numpy(g["cfg2.points3D"])
    conf = torch.from_numpy(g["cfg2.confidences"])
    # the reference's row loop (predict3D.py:88-97)
    row = []
    for point, c in zip(pts.squeeze(), conf.squeeze().cpu().numpy()):
        row = row + point.tolist() + [c]
    writer.writerow(row)
    row = []
    for i in range(23 * 4):
        row = row + ["NaN"]
    writer.writerow(row)
    with open(os.path.join(HERE, "data3D_expected.csv"), "w", newline="") as f:
        f.write(buf.getvalue())
    print("wrote data3D_expected.csv (%d bytes)" % len(buf.getvalue()))


def case_csv2d():
    """data2D.csv wire format (prediction/predict2D.py:71-109,120-125): header by the
    reference's own create_header, rows by its row loop over the reference predictor's outputs
    of the predictor2d fixture, one 'NaN' row."""
    import csv
    import io
    from jarvis.prediction.predict2D import create_header
    g = dict(np.load(os.path.join(HERE, "predictor2d.npz")))
    J = 12
    cfg = R.ns(KEYPOINT_NAMES=["joint%d" % i for i in range(J)],
               KEYPOINTDETECT=R.ns(NUM_JOINTS=J))
    buf = io.StringIO()
    writer = csv.writer(buf, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
    create_header(writer, cfg)
    for tag in ("cam0_j12", "cam2_j12", None):
        if tag is not None:
            # predict2D.py:98-103
            points2D = torch.from_numpy(g[tag + ".points2D"]).cpu().numpy()
            confidences = torch.from_numpy(g[tag + ".confidences"]).cpu().numpy()
            row = []
            for i, point in enumerate(points2D):
                row = row + point.tolist() + [confidences[i]]
            writer.writerow(row)
        else:
            # predict2D.py:105-109
            row = []
            for i in range(cfg.KEYPOINTDETECT.NUM_JOINTS * 3):
                row = row + ["NaN"]
            writer.writerow(row)
    with open(os.path.join(HERE, "data2D_expected.csv"), "w", newline="") as f:
        f.write(buf.getvalue())
    print("wrote data2D_expected.csv (%d bytes)" % len(buf.getvalue()))


def case_analysis():
    """analysis/analyze.py:22-96 run for real with its collaborators (project manager,
    Dataset3D, predictor, calibration loader) replaced by seeded stand-ins: the three CSV
    files it writes are the wire-format fixture for jarvis_hybridnet_amd.analysis."""
    import shutil
    import jarvis.analysis.analyze as A
    J = 23
    samples, preds = cases.analysis_samples(J)
    out_root = tempfile.mkdtemp()

    class FakeProject:
        parent_dir = out_root
        cfg = R.ns(PROJECTS_ROOT_PATH="projects", KEYPOINTDETECT=R.ns(NUM_JOINTS=J),
                   DATALOADER_NUM_WORKERS=0)

        def load(self, name):
            return True

        def get_cfg(self):
            return self.cfg

    class FakeDataset(torch.utils.data.Dataset):
        image_ids = list(range(len(samples)))

        def __init__(self, **kw):
            pass

        def __len__(self):
            return len(samples)

        def __getitem__(self, i):
            return samples[i]

    class FakePredictor:
        def __init__(self, *a):
            self.i = 0

        def __call__(self, imgs, camM, K, D):
            assert imgs.dtype == torch.float32 and imgs.shape[1] == 3
            self.i += 1
            return preds[self.i - 1], None
    tool = R.ns(cameraMatrices=torch.zeros(2, 4, 3), intrinsicMatrices=torch.zeros(2, 3, 3),
                distortionCoefficients=torch.zeros(2, 1, 5))
    A.ProjectManager, A.Dataset3D, A.JarvisPredictor3D = FakeProject, FakeDataset, FakePredictor
    A.load_reprojection_tools = lambda cfg, cameras_to_use=None: {"calibA": tool}
    real_loader = A.DataLoader
    A.DataLoader = lambda ds, **kw: real_loader(ds, **dict(kw, pin_memory=False))
    A.analyze_validation_data("golden")
    adir = os.path.join(out_root, "projects", "golden", "analysis")
    run = os.path.join(adir, os.listdir(adir)[0])
    dst = os.path.join(HERE, "analysis")
    os.makedirs(dst, exist_ok=True)
    for f in ("frame_names.csv", "points_HybridNet.csv", "points_GroundTruth.csv"):
        shutil.copy(os.path.join(run, f), os.path.join(dst, f))
        print("wrote analysis/%s (%d bytes)" % (f, os.path.getsize(os.path.join(dst, f))))
    shutil.rmtree(out_root)


ALL = dict(analysis=case_analysis, reprojection_hash=case_reprojection_hash, csv2d=case_csv2d, calibration=case_calibration, csv=case_csv, predictor2d=case_predictor2d, state_spec=case_state_spec, efficienttrack=case_efficienttrack,
           reprojection=case_reprojection, v2v=case_v2v, geometry=case_geometry,
           hybridnet=case_hybridnet, predictor=case_predictor)

def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


if __name__ == "__main__":
    with open(os.path.join(HERE, "environment.json"), "w") as f:
        json.dump(dict(cpu=cpu_model(), torch=torch.__version__), f)
    names = sys.argv[1:] or list(ALL)
    for n in names:
        t0 = time.time()
        ALL[n]()
        print("== %s done in %.1fs" % (n, time.time() - t0))
