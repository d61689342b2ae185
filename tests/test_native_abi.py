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


def test_roofline_pricing_takes_the_binding_floor():
    """bench.price: a convolution is priced against the HIGHER of its MFMA and HBM floors (the few-channel 1x1 layers are
    HBM-bound); everything else against HBM.  Pure arithmetic, no GPU."""
    import importlib
    import sys
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    bench = importlib.import_module("bench")
    # conv2d_k1s1_16x8@128 at 384 images: 604 MB, 3.2 GFLOP algorithmic, 0.116 ms
    px = 384 * 128 * 128
    row = bench.price("conv2d_k1s1_16x8@128", 2.0 * px * 16 * 8, 4.0 * px * (16 + 8), 0.116e-3)
    assert row["bound"] == "hbm" and 0.6 < row["frac"] < 0.7 and row["frac_mfma"] < 0.3
    assert abs(row["frac"] - row["frac_hbm"]) < 1e-12
    # the Winograd V2V convolution: MFMA-bound, executed FLOPs = 12 / 27 of the direct count with 46 -> 48 / 48 padding
    vox = 32 * 32 ** 3
    row = bench.price("conv3d_k3s1wino_46x46@32", 2.0 * vox * 46 * 46 * 27, 4.0 * vox * 92, 0.566e-3)
    assert row["bound"] == "mfma" and 0.6 < row["frac"] < 0.7 and row["algorithmic_equiv"] > bench.PEAK_F32_MFMA_TFLOPS
    # a kernel that is no convolution: HBM, whatever its FLOPs
    row = bench.price("bifpn_node_56x56@64", 1e12, 880e6, 0.217e-3)
    assert row["bound"] == "hbm" and "frac_mfma" not in row
