"""ReprojectionTool on MI355X (mirrors jarvis/utils/reprojection.py:16-166).

Calibration tensors use the reference's storage: cameraMatrices (C,4,3),
intrinsicMatrices (C,3,3) with the principal point in row 2,
distortionCoefficients (C,1,5).  `reprojectPoint` / `reconstructPoint` run in
libjarvis_hip; calibration files (OpenCV FileStorage YAML) are parsed here
without cv2.
"""
import json
import os
import re

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N

_MAT = re.compile(r"^(\w+):\s*!!opencv-matrix\s*\n\s*rows:\s*(\d+)\s*\n\s*cols:\s*(\d+)\s*\n\s*dt:\s*(\w+)\s*\n"
                  r"\s*data:\s*\[(.*?)\]", re.S | re.M)


def read_opencv_yaml(path):
    """{name: float64 ndarray (rows, cols)} of every `!!opencv-matrix` node of an
    OpenCV FileStorage YAML file (what cv2.FileStorage(...).getNode(name).mat() returns,
    reprojection.py:109-111)."""
    text = open(path).read()
    out = {}
    for name, rows, cols, dt, data in _MAT.findall(text):
        vals = [float(v) for v in data.replace("\n", " ").split(",") if v.strip()]
        out[name] = np.array(vals, dtype=np.float64).reshape(int(rows), int(cols))
    if not out:
        raise ValueError("no !!opencv-matrix nodes in " + path)
    return out


class TorchCamera(nn.Module):
    """reprojection.py:93-111: position, rotationMatrix, intrinsicMatrix,
    distortionCoeffccients (sic) and cameraMatrix = ([R; T] @ K)^T."""

    def __init__(self, name, calib_path, device="cuda"):
        super().__init__()
        self.name = name
        mats = read_opencv_yaml(calib_path)
        self.position = torch.from_numpy(mats["T"]).float().to(device)
        self.rotationMatrix = torch.from_numpy(mats["R"]).float().to(device)
        self.intrinsicMatrix = torch.from_numpy(mats["intrinsicMatrix"]).float().to(device)
        self.distortionCoeffccients = torch.from_numpy(
            mats["distortionCoefficients"]).float().to(device)
        self.distortionCoefficients = self.distortionCoeffccients
        self.cameraMatrix = torch.transpose(torch.matmul(
            torch.cat((self.rotationMatrix, self.position.reshape(1, 3)), axis=0),
            self.intrinsicMatrix), 0, 1).float().to(device)


class ReprojectionTool(nn.Module):
    def __init__(self, root_dir=None, calib_paths=None, device="cuda"):
        super().__init__()
        self.device = device
        if calib_paths is not None:
            self.cameras = {cam: TorchCamera(cam, os.path.join(root_dir, calib_paths[cam]), device)
                            for cam in calib_paths}
            self.camera_list = [self.cameras[cam] for cam in self.cameras]
            self.num_cameras = len(self.camera_list)
            self.cameraMatrices = torch.stack(
                [c.cameraMatrix.transpose(0, 1) for c in self.camera_list]).to(device)
            self.intrinsicMatrices = torch.stack(
                [c.intrinsicMatrix for c in self.camera_list]).to(device)
            self.distortionCoefficients = torch.stack(
                [c.distortionCoeffccients for c in self.camera_list]).to(device)
        else:
            self.cameraMatrices = torch.tensor(0)
            self.intrinsicMatrices = torch.tensor(0)
            self.distortionCoefficients = torch.tensor(0)

    def reprojectPoint(self, point3D):
        """point3D (P,3) -> (C,2) for P == 1 (the reference squeezes), else (C,2,P)."""
        pts = N.dev(point3D)
        C = self.cameraMatrices.shape[0]
        uv = torch.empty((C, pts.shape[0], 2), device=pts.device, dtype=torch.float32)
        N.check(N.lib().jh_reproject_point(
            N.ptr(pts), pts.shape[0], C, N.ptr(N.dev(self.cameraMatrices)),
            N.ptr(N.dev(self.intrinsicMatrices)), N.ptr(N.dev(self.distortionCoefficients)),
            N.ptr(uv), N.stream()))
        return uv.permute(0, 2, 1).squeeze()

    def reconstructPoint(self, points, maxvals):
        """points (2,C) pixels, maxvals (C,1,1) weights -> (3,) mm."""
        pts = N.dev(points)
        C = pts.shape[1]
        out = torch.empty(3, device=pts.device, dtype=torch.float32)
        ws = N.workspace(N.lib().jh_reconstruct_workspace_bytes(C), pts.device)
        N.check(N.lib().jh_reconstruct_point(
            N.ptr(pts), N.ptr(N.dev(maxvals.reshape(-1))), C, N.ptr(N.dev(self.cameraMatrices)),
            N.ptr(N.dev(self.intrinsicMatrices)), N.ptr(N.dev(self.distortionCoefficients)),
            N.ptr(out), ws.data_ptr(), ws.numel(), N.stream()))
        return out


def load_reprojection_tools(cfg, cameras_to_use=None, device="cuda"):
    """{calibration set name: ReprojectionTool} from the dataset's
    annotations/instances_val.json (reprojection.py:148-166)."""
    dataset_dir = os.path.join(cfg.PARENT_DIR, cfg.DATASET.DATASET_ROOT_DIR, cfg.DATASET.DATASET_3D)
    with open(os.path.join(dataset_dir, "annotations", "instances_val.json")) as f:
        data = json.load(f)
    tools = {}
    for calib in data["calibrations"]:
        paths = {cam: p for cam, p in data["calibrations"][calib].items()
                 if cameras_to_use is None or cam in cameras_to_use}
        tools[calib] = ReprojectionTool(dataset_dir, paths, device)
    return tools


def get_repro_tool(cfg, dataset_name, device="cuda"):
    """reprojection.py:115-145: the tool of a named calibration set, the only one, or the
    first one; a directory name selects calibration files inside that directory."""
    tools = load_reprojection_tools(cfg, device=device)
    if dataset_name is not None and dataset_name not in tools:
        if os.path.isdir(dataset_name):
            dataset_dir = os.path.join(cfg.PARENT_DIR, cfg.DATASET.DATASET_ROOT_DIR,
                                       cfg.DATASET.DATASET_3D)
            with open(os.path.join(dataset_dir, "annotations", "instances_val.json")) as f:
                data = json.load(f)
            first = list(data["calibrations"].keys())[0]
            paths = {cam: p.split("/")[-1] for cam, p in data["calibrations"][first].items()}
            return ReprojectionTool(dataset_name, paths, device)
        return None
    if len(tools) == 1 or (len(tools) > 1 and dataset_name is None):
        return tools[list(tools.keys())[0]]
    if len(tools) > 1:
        return tools[dataset_name]
    return None
