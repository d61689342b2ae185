"""GPU: HybridNetBackbone.forward and JarvisPredictor3D.forward (HIP) against
the CPU oracle and the reference's golden outputs, incl. the (None, None) path
and the time-batched throughput form."""
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


@pytest.mark.parametrize("tag", ["cfg2", "cfg3"])
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
