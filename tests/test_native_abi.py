"""CPU: the C-ABI library loads and exports every symbol include/jarvis_hip.h
declares; the Python modules carry the reference's parameter layout.  No
compute call is made (there is no GPU on the build machine)."""
import json
import os
import re

import pytest

from jarvis_hybridnet_amd import _native as N

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "jarvis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_header_symbols():
    lib = N.lib()                       # raises if the .so is missing
    names = header_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "libjarvis_hip.so lacks " + name
    assert names == N.symbols(), "ctypes table and header disagree"
    assert lib.jh_abi_version() == N.ABI_VERSION == 4
    assert lib.jh_last_error() is not None


def test_header_is_valid_c_and_links(tmp_path):
    """include/jarvis_hip.h in a plain C translation unit (gcc -std=c11 -Wall -Werror), with
    static asserts on the jh_predictor_config layout that _native.PredictorConfig mirrors;
    the program is linked against libjarvis_hip.so and calls the GPU-free entry points."""
    import ctypes
    import subprocess
    N.lib()
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(N.LIB_PATH)
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic",
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_smoke.c"),
                    "-o", exe, "-L", libdir, "-ljarvis_hip", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.split() == ["abi", str(N.ABI_VERSION), "config", str(ctypes.sizeof(N.PredictorConfig))]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libjarvis_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.lib()


def test_cpu_tensor_is_rejected():
    import torch
    with pytest.raises(RuntimeError, match="needs CUDA"):
        N.dev(torch.zeros(3))


def test_module_state_dict_layout():
    from types import SimpleNamespace as NS
    from jarvis_hybridnet_amd.efficienttrack.model import EfficientTrackBackbone
    from jarvis_hybridnet_amd.hybridnet.model import HybridNetBackbone
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "state_spec.json")))
    for size in ("small", "medium", "large"):
        m = EfficientTrackBackbone(None, size, 23)
        got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert got == spec["efficienttrack.%s.23" % size]
    cfg = NS(DATASET=NS(DATASET_ROOT_DIR="x"),
             KEYPOINTDETECT=NS(MODEL_SIZE="small", NUM_JOINTS=23, BOUNDING_BOX_SIZE=256),
             HYBRIDNET=NS(NUM_CAMERAS=4, ROI_CUBE_SIZE=32, GRID_SPACING=2))
    hb = HybridNetBackbone(cfg)
    got = [[k, list(v.shape)] for k, v in hb.state_dict().items()]
    assert got == spec["hybridnet.small.23"]
    # reference checkpoints load with strict=True
    sd = {k: v.clone() for k, v in hb.state_dict().items()}
    hb.load_state_dict(sd, strict=True)
