"""GPU: the callers either side of the hot path with the REAL HIP predictor -- the validation-analysis loop
(jarvis/analysis/analyze.py:54-96) and the reference's acceleration seam by attribute assignment
(jarvis/prediction/jarvis3D.py:64-69)."""
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from tests import cases
from tests.gpu_util import cuda, report
from tests.test_hip_predictor import make_cfg

pytestmark = pytest.mark.gpu


def test_analyze_frames_with_the_hip_predictor(tmp_path, golden):
    """analyze_frames over five seeded frame sets through a DataLoader(batch_size=1), predictions by
    JarvisPredictor3D (HIP): the not-detected frame set is left out of all three files, row 0 of points_HybridNet.csv
    is the reference's own output for that frame set (fixture cfg2_partial) within 1e-3 mm, every row carries exactly
    the float32 values a direct call of the predictor returns, points_GroundTruth.csv the samples' keypoints."""
    from torch.utils.data import DataLoader
    from jarvis_hybridnet_amd.analysis.analyze import analyze_frames
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    c, inp, samples, frames = cases.analysis_gpu_samples()
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    tool = NS(cameraMatrices=inp["cam"], intrinsicMatrices=inp["intr"], distortionCoefficients=inp["dist"])
    loader = DataLoader(samples, batch_size=1, shuffle=False)
    seen, done = analyze_frames(pred, loader, {"ringA": tool}, str(tmp_path), c["J"], num_frame_sets=len(samples))
    assert (seen, done) == (5, sum(cases.ANALYSIS_GPU_VALID))
    names = open(tmp_path / "frame_names.csv").read().split()
    keep = [i for i, v in enumerate(cases.ANALYSIS_GPU_VALID) if v]
    assert names == ["Frame_%03d.jpg" % i for i in keep]
    net = np.loadtxt(tmp_path / "points_HybridNet.csv", delimiter=",").reshape(len(keep), c["J"], 3)
    gt = np.loadtxt(tmp_path / "points_GroundTruth.csv", delimiter=",").reshape(len(keep), c["J"], 3)
    e0 = float(np.abs(net[0] - golden("predictor")["cfg2_partial.points3D"][0]).max())
    report("analyze_frames_hip", row0_vs_reference_fixture_mm=e0)
    assert e0 < 1e-3
    dev = [cuda(t) for t in (inp["cam"], inp["intr"], inp["dist"])]
    for row, i in enumerate(keep):
        pts, _ = pred(cuda(frames[i]), *dev)
        assert pts is not None
        # numpy.savetxt's %.18e round-trips float32 exactly
        assert np.array_equal(net[row].astype(np.float32), pts[0].cpu().numpy()), i
        assert np.array_equal(gt[row], samples[i][1])
    for i, v in enumerate(cases.ANALYSIS_GPU_VALID):
        if not v:
            assert pred(cuda(frames[i]), *dev) == (None, None)


@pytest.mark.parametrize("slot", ["centerDetect", "hybridNet.effTrack", "hybridNet.v2vNet", "hybridNet"])
def test_replaced_submodule_is_refused(slot):
    """The reference's trt_mode seam ASSIGNS compiled callables to these attributes (jarvis3D.py:64-69).  The native
    forward is built from the weights of its own modules, so a foreign callable would be ignored: it must raise, with
    the attribute's name, in every forward form -- and work again once a native module is back in the slot."""
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    dev = [cuda(t) for t in (inp["cam"], inp["intr"], inp["dist"])]
    x = cuda(inp["imgs"])
    ref = pred(x, *dev)
    owner, name = (pred, slot) if "." not in slot else (pred.hybridNet, slot.split(".")[1])
    native = getattr(owner, name)
    setattr(owner, name, torch.nn.Identity())            # what `torch.jit.load(...)` would put there
    for call in (lambda: pred(x, *dev), lambda: pred.forward_batch(x[None], *dev),
                 lambda: pred.forward_uint8((x.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous(), *dev)):
        with pytest.raises(RuntimeError, match=slot.replace(".", r"\.") + " has been replaced"):
            call()
    if slot.startswith("hybridNet."):
        with pytest.raises(RuntimeError, match=name + " has been replaced"):
            pred.hybridNet(cuda(torch.zeros(1, c["C"], 3, c["bbox"], c["bbox"])), None,
                           cuda(torch.zeros(1, c["C"], 2, dtype=torch.int32)),
                           cuda(torch.zeros(1, 3, dtype=torch.int32)), dev[0][None], dev[1][None], dev[2][None])
    setattr(owner, name, native)
    again = pred(x, *dev)
    assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])
