"""Deterministic synthetic inputs for parity tests and the benchmark.

There is no network on the build or GPU machines, so every measurement and
parity case uses seeded random-init weights of the reference architecture,
a seeded ring calibration and Gaussian-blob frames.
Nothing here is arithmetic of the hot path itself.
"""
import math

import numpy as np
import torch

from . import arch

MEAN = [0.485, 0.456, 0.406]
STD = [0.229, 0.224, 0.225]


def _fill(spec, seed, deconv_std=1.2):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in spec:
        leaf = key.split(".")[-1]
        short = key.rsplit(".", 1)[0].split(".")[-1]
        if len(shape) == 1 and leaf != "bias":
            t = torch.rand(shape, generator=g) + 0.5           # fusion weights
        elif leaf == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif short == "deconv1":
            t = torch.randn(shape, generator=g) * deconv_std
        elif short == "output_layer":
            t = torch.randn(shape, generator=g)
        elif short == "block" and "decoder_upsample1" in key:
            fan = shape[0]                                      # ConvT: in-channels
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
        else:
            fan = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
        sd[key] = t.float().contiguous()
    return sd


def efficienttrack_weights(model_size, num_joints, seed, deconv_std=1.2):
    return _fill(arch.efficienttrack_params(model_size, num_joints), seed, deconv_std)


def v2v_weights(num_joints, seed):
    return _fill(arch.v2v_params(num_joints), seed)


def hybridnet_weights(model_size, num_joints, seed, deconv_std=1.2):
    return _fill(arch.hybridnet_params(model_size, num_joints), seed, deconv_std)


def ring_cameras(num_cameras, width, height, focal=1800.0, radius=1500.0, k1=-0.05, k2=0.01):
    """Per-camera (R (3,3), T (3,), Kt (3,3), dist (5,)) in the convention of the
    reference's calibration files (row-vector: x_cam = X @ R + T, pixel_h = x_cam @ Kt)."""
    cams = []
    for i in range(num_cameras):
        th = 2.0 * math.pi * i / num_cameras
        ph = 0.3 * math.sin(3.0 * th)
        pos = radius * np.array([math.cos(th) * math.cos(ph),
                                 math.sin(th) * math.cos(ph), math.sin(ph)])
        z = -pos / np.linalg.norm(pos)
        x = np.cross(z, np.array([0.0, 0.0, 1.0]))
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        rwc = np.stack([x, y, z], 0)
        kt = np.array([[focal, 0, 0], [0, focal, 0], [width / 2.0, height / 2.0, 1.0]])
        cams.append((rwc.T, -pos @ rwc.T, kt, np.array([k1, k2, 0.0, 0.0, 0.0])))
    return cams


def write_opencv_yaml(path, R, T, Kt, dist):
    """One camera in the OpenCV FileStorage YAML layout the reference reads
    (datasets/*/calib_params/*/*.yaml: intrinsicMatrix, distortionCoefficients, R, T)."""
    def mat(name, a, rows, cols):
        data = ", ".join(repr(float(v)) for v in np.asarray(a, dtype=np.float64).reshape(-1))
        return "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: d\n   data: [ %s ]\n" % (
            name, rows, cols, data)
    with open(path, "w") as f:
        f.write("%YAML:1.0\n---\n")
        f.write(mat("intrinsicMatrix", Kt, 3, 3))
        f.write(mat("distortionCoefficients", dist, 1, 5))
        f.write(mat("R", R, 3, 3))
        f.write(mat("T", T, 3, 1))


def ring_calibration(num_cameras, width, height, focal=1800.0, radius=1500.0,
                     k1=-0.05, k2=0.01):
    """Cameras on a ring looking at the origin, in the reference's storage
    convention (utils/reprojection.py:33-39,105-107): cameraMatrices (C,4,3) =
    [R;T] @ Kt, intrinsicMatrices (C,3,3) = Kt with the principal point in row
    2, distortionCoefficients (C,1,5) with k1,k2 in the first two slots."""
    cam = np.zeros((num_cameras, 4, 3))
    intr = np.zeros((num_cameras, 3, 3))
    dist = np.zeros((num_cameras, 1, 5))
    for i in range(num_cameras):
        th = 2.0 * math.pi * i / num_cameras
        ph = 0.3 * math.sin(3.0 * th)
        pos = radius * np.array([math.cos(th) * math.cos(ph),
                                 math.sin(th) * math.cos(ph), math.sin(ph)])
        z = -pos / np.linalg.norm(pos)
        x = np.cross(z, np.array([0.0, 0.0, 1.0]))
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        rwc = np.stack([x, y, z], 0)            # rows = camera axes in world
        R = rwc.T
        T = -pos @ rwc.T
        kt = np.array([[focal, 0, 0], [0, focal, 0], [width / 2.0, height / 2.0, 1.0]])
        cam[i] = np.concatenate([R, T[None]], 0) @ kt
        intr[i] = kt
        dist[i, 0, 0], dist[i, 0, 1] = k1, k2
    return (torch.from_numpy(cam).float(), torch.from_numpy(intr).float(),
            torch.from_numpy(dist).float())


def project(points, cam, intr, dist):
    """float64 pinhole + 2-term radial projection of (P,3) points -> (C,P,2)."""
    cam, intr, dist = (t.double().numpy() for t in (cam, intr, dist))
    ph = np.concatenate([points, np.ones((points.shape[0], 1))], 1)
    out = np.zeros((cam.shape[0], points.shape[0], 2))
    for c in range(cam.shape[0]):
        q = ph @ cam[c]
        u = q[:, 0] / q[:, 2] - intr[c, 2, 0]
        v = q[:, 1] / q[:, 2] - intr[c, 2, 1]
        r2 = (u / intr[c, 0, 0]) ** 2 + (v / intr[c, 1, 1]) ** 2
        d = 1 + (dist[c, 0, 0] + dist[c, 0, 1] * r2) * r2
        out[c, :, 0] = u * d + intr[c, 2, 0]
        out[c, :, 1] = v * d + intr[c, 2, 1]
    return out


def blob_frames(calib, width, height, num_joints, seed, sigma=6.0, amp=0.9):
    """(C,3,H,W) fp32 RGB frames in [0,1]: dim noise plus one Gaussian blob per
    joint at the projection of a random 3D skeleton.  Returns (frames,
    joints3d (J,3), centre (3,))."""
    cam, intr, dist = calib
    rng = np.random.RandomState(seed)
    centre = rng.uniform(-100, 100, 3)
    joints = centre + rng.uniform(-40, 40, (num_joints, 3))
    uv = project(joints, cam, intr, dist)
    C = cam.shape[0]
    g = torch.Generator().manual_seed(seed)
    frames = torch.rand((C, 3, height, width), generator=g) * 0.05
    r = int(4 * sigma)
    ax = torch.arange(-r, r + 1, dtype=torch.float32)
    for c in range(C):
        for j in range(num_joints):
            u, v = uv[c, j]
            iu, iv = int(round(u)), int(round(v))
            if iu - r < 0 or iv - r < 0 or iu + r >= width or iv + r >= height:
                continue
            gx = torch.exp(-((ax + iu - float(u)) ** 2) / (2 * sigma * sigma))
            gy = torch.exp(-((ax + iv - float(v)) ** 2) / (2 * sigma * sigma))
            blob = amp * gy[:, None] * gx[None, :]
            col = 0.5 + 0.5 * torch.tensor([math.sin(j), math.cos(2 * j), math.sin(3 * j + 1)])
            frames[c, :, iv - r:iv + r + 1, iu - r:iu + r + 1] += col.view(3, 1, 1) * blob
    return frames.clamp_(0, 1), joints, centre


def smooth_heatmaps(num_cameras, num_joints, size, seed):
    """Smooth positive (C,J,size,size) heatmap field in 0..255 units used by
    stage-level reprojection tests."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32),
                            torch.arange(size, dtype=torch.float32), indexing="ij")
    out = torch.zeros(num_cameras, num_joints, size, size)
    for c in range(num_cameras):
        for j in range(num_joints):
            cx, cy = (torch.rand(2, generator=g) * size).tolist()
            s = 4.0 + 10.0 * torch.rand(1, generator=g).item()
            a = 60.0 + 190.0 * torch.rand(1, generator=g).item()
            out[c, j] = a * torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return out + torch.rand(out.shape, generator=g) * 2.0
