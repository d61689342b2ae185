"""CPU: weight files on disk -> modules.  Mirrors
jarvis/hybridnet/hybridnet.py:84-97,118-131 and
jarvis/efficienttrack/efficienttrack.py:90-113,165-183: a `.pth` is a plain
OrderedDict[str, Tensor]; `load_weights(path)`; `'latest'` = newest run directory (mtime)
holding a `*_final.pth`; a missing file returns False silently.  No compute call is made
(`.cuda()` is neutralised: there is no GPU on the build machine)."""
import os
import time
from types import SimpleNamespace as NS

import pytest
import torch

from jarvis_hybridnet_amd import synthetic as S

REF = "/root/reference/pretrained/MonkeyHand"


@pytest.fixture(autouse=True)
def no_cuda(monkeypatch):
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)


def make_cfg(root, J=5):
    return NS(PARENT_DIR=str(root), PROJECT_NAME="proj",
              DATASET=NS(DATASET_ROOT_DIR="x", MEAN=S.MEAN, STD=S.STD),
              CENTERDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=1, IMAGE_SIZE=128),
              KEYPOINTDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=J, BOUNDING_BOX_SIZE=128),
              HYBRIDNET=NS(NUM_CAMERAS=3, ROI_CUBE_SIZE=32, GRID_SPACING=2))


def same(module, sd):
    got = module.state_dict()
    return list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_efficienttrack_load_weights_from_disk(tmp_path):
    from jarvis_hybridnet_amd.efficienttrack.efficienttrack import EfficientTrack
    cfg = make_cfg(tmp_path)
    sd = S.efficienttrack_weights("small", 1, 3)
    path = tmp_path / "center.pth"
    torch.save(sd, path)
    net = EfficientTrack("CenterDetectInference", cfg, str(path))
    assert same(net.model, sd)
    assert not any(p.requires_grad for p in net.model.parameters()) and not net.model.training
    v = net.model.weights_version
    # missing file: False, silently, weights untouched (efficienttrack.py:109-110)
    assert net.load_weights(str(tmp_path / "nope.pth")) is False
    assert same(net.model, sd) and net.model.weights_version == v
    # reload bumps the version (native plans are rebuilt)
    sd2 = S.efficienttrack_weights("small", 1, 4)
    torch.save(sd2, path)
    assert net.load_weights(str(path)) is True
    assert same(net.model, sd2) and net.model.weights_version == v + 1
    # strict=False (efficienttrack.py:106): a checkpoint without the dead heads still loads
    part = {k: t for k, t in sd.items() if not k.startswith("final_conv")}
    torch.save(part, path)
    assert net.load_weights(str(path)) is True
    assert torch.equal(net.model.state_dict()["deconv1.weight"], sd["deconv1.weight"])


def test_latest_weights_by_mtime(tmp_path):
    from jarvis_hybridnet_amd.efficienttrack.efficienttrack import EfficientTrack
    from jarvis_hybridnet_amd.hybridnet.hybridnet import HybridNet
    cfg = make_cfg(tmp_path)
    # no models directory at all: 'latest' resolves to nothing, construction still works
    net = EfficientTrack("KeypointDetectInference", cfg, "latest")
    assert net.get_latest_weights() is None
    base = tmp_path / "projects" / "proj" / "models"
    now = time.time()
    sds = {}
    for i, run in enumerate(["Run_A", "Run_B", "Run_C"]):
        d = base / "KeypointDetect" / run
        d.mkdir(parents=True)
        if run != "Run_C":                     # the newest run has no final weights yet
            sds[run] = S.efficienttrack_weights("small", 5, 10 + i)
            torch.save(sds[run], d / "EfficientTrack-small_final.pth")
        os.utime(d, (now + 10 * i, now + 10 * i))
    net = EfficientTrack("KeypointDetectInference", cfg, "latest")
    assert net.get_latest_weights().endswith(os.path.join("Run_B", "EfficientTrack-small_final.pth"))
    assert same(net.model, sds["Run_B"])
    os.utime(base / "KeypointDetect" / "Run_A", (now + 100, now + 100))
    assert net.get_latest_weights().endswith(os.path.join("Run_A", "EfficientTrack-small_final.pth"))
    # HybridNet: strict=True, same 'latest' rule (hybridnet.py:84-97,118-131)
    sdh = S.hybridnet_weights("small", 5, 20)
    d = base / "HybridNet" / "Run_1"
    d.mkdir(parents=True)
    torch.save(sdh, d / "HybridNet-small_final.pth")
    hn = HybridNet("inference", cfg, "latest")
    assert same(hn.model, sdh) and len(sdh) == 188
    assert hn.load_weights(str(tmp_path / "missing.pth")) is False
    bad = dict(sdh)
    bad.pop("v2vNet.output_layer.bias")
    torch.save(bad, d / "broken.pth")
    with pytest.raises(RuntimeError, match="Missing key"):
        hn.load_weights(str(d / "broken.pth"))
    # efficienttrack_weights= seeds only the 2D sub-network (hybridnet/model.py:33-37)
    ek = {k[len("effTrack."):]: v for k, v in S.hybridnet_weights("small", 5, 21).items()
          if k.startswith("effTrack.")}
    torch.save(ek, tmp_path / "kp.pth")
    hn2 = HybridNet("inference", cfg, None, str(tmp_path / "kp.pth"))
    assert same(hn2.model.effTrack, ek)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's checkpoints only exist in the build container")
@pytest.mark.parametrize("name,mode,J", [("EfficientTrack_Center-small.pth", "CenterDetectInference", 1),
                                          ("EfficientTrack_Keypoints-small.pth", "KeypointDetectInference", None)])
def test_reference_checkpoints_load_strict(name, mode, J):
    """The reference's shipped MonkeyHand checkpoints (CUDA-saved OrderedDicts) load with
    strict=True into the build's modules: same keys, same order, same shapes."""
    from jarvis_hybridnet_amd.efficienttrack.model import EfficientTrackBackbone
    sd = torch.load(os.path.join(REF, name), map_location="cpu")
    if J is None:
        J = sd["deconv1.weight"].shape[1]
    m = EfficientTrackBackbone(None, "small", J)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(m.state_dict()) == list(sd)
    assert all(m.state_dict()[k].shape == sd[k].shape and torch.equal(m.state_dict()[k], sd[k].float())
               for k in sd)
    assert len(sd) == 164 and sum(v.numel() for v in sd.values()) > 400000
