"""CPU: the data formats on either side of the hot path:
OpenCV-YAML calibration files -> ReprojectionTool tensors, and the data3D.csv /
info.yaml wire format.  Host logic only; no compute call."""
import csv
import io
import os
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CALIB = os.path.join(HERE, "golden", "calib")


def test_calibration_loader_matches_reference(golden):
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd.utils.reprojection import ReprojectionTool, read_opencv_yaml
    names = ["Camera_A", "Camera_B", "Camera_C", "Camera_D"]
    tool = ReprojectionTool(CALIB, {n: n + ".yaml" for n in names}, device="cpu")
    g = golden("calibration")
    # fixtures = output of the reference's TorchCamera / ReprojectionTool on these files
    assert np.array_equal(tool.cameraMatrices.numpy(), g["cameraMatrices"])
    assert np.array_equal(tool.intrinsicMatrices.numpy(), g["intrinsicMatrices"])
    assert np.array_equal(tool.distortionCoefficients.numpy(), g["distortionCoefficients"])
    assert tool.cameraMatrices.shape == (4, 4, 3) and tool.num_cameras == 4
    assert list(tool.cameras) == names
    m = read_opencv_yaml(os.path.join(CALIB, "Camera_A.yaml"))
    assert m["R"].shape == (3, 3) and m["T"].shape == (3, 1) and m["intrinsicMatrix"][2, 2] == 1.0
    # the files describe the rig of synthetic.ring_calibration
    cam, intr, dist = S.ring_calibration(4, 640, 512, 900.0)
    assert (tool.cameraMatrices - cam).abs().max() < 1e-2


def test_load_reprojection_tools(tmp_path):
    import json
    import shutil
    from jarvis_hybridnet_amd.utils.reprojection import get_repro_tool, load_reprojection_tools
    root = tmp_path / "datasets" / "ds3d"
    (root / "annotations").mkdir(parents=True)
    (root / "calib_params" / "rig").mkdir(parents=True)
    for f in os.listdir(CALIB):
        shutil.copy(os.path.join(CALIB, f), root / "calib_params" / "rig" / f)
    cal = {"rig": {f[:-5]: "calib_params/rig/" + f for f in sorted(os.listdir(CALIB))}}
    json.dump({"calibrations": cal}, open(root / "annotations" / "instances_val.json", "w"))
    cfg = NS(PARENT_DIR=str(tmp_path), DATASET=NS(DATASET_ROOT_DIR="datasets", DATASET_3D="ds3d"))
    tools = load_reprojection_tools(cfg, device="cpu")
    assert list(tools) == ["rig"] and tools["rig"].num_cameras == 4
    sub = load_reprojection_tools(cfg, cameras_to_use=["Camera_A", "Camera_C"], device="cpu")
    assert sub["rig"].num_cameras == 2
    assert get_repro_tool(cfg, None, device="cpu").num_cameras == 4
    assert get_repro_tool(cfg, "rig", device="cpu").num_cameras == 4


def test_csv_wire_format(golden, tmp_path):
    from jarvis_hybridnet_amd.prediction.predict3D import create_header, create_info_file, frame_row
    g = golden("predictor")
    cfg = NS(KEYPOINT_NAMES=["joint%d" % i for i in range(23)])
    buf = io.StringIO()
    writer = csv.writer(buf, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
    create_header(writer, cfg)
    writer.writerow(frame_row(torch.from_numpy(g["cfg2.points3D"]),
                              torch.from_numpy(g["cfg2.confidences"]), 23))
    writer.writerow(frame_row(None, None, 23))
    expected = open(os.path.join(HERE, "golden", "data3D_expected.csv"), newline="").read()
    assert buf.getvalue() == expected            # byte-identical to the reference's rows
    params = NS(output_dir=str(tmp_path), recording_path="/rec/a", dataset_name=None,
                frame_start=5, number_frames=10)
    create_info_file(params)
    text = open(tmp_path / "info.yaml").read().splitlines()
    assert text == ["recording_path: /rec/a", "dataset_name: null", "frame_start: 5",
                    "number_frames: 10"]
