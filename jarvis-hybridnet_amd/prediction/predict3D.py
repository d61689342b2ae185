"""Output side of the predict3D driver (mirrors jarvis/prediction/predict3D.py:64-70,
87-97,141-155): `data3D.csv` rows and `info.yaml`, so that the
reference's visualisation / analysis tools consume the results unchanged.

Video decoding (cv2.VideoCapture) and project management are outside the hot path:
`predict3D_frames` takes any iterable of decoded multi-view frame sets instead.
"""
import csv
import itertools
import json
import os
import re

import torch

_PLAIN = re.compile(r"^[A-Za-z0-9_./\\][A-Za-z0-9_./\\ +=,@%-]*$")
_RESERVED = {"", "~", "null", "true", "false", "yes", "no", "on", "off", "y", "n"}


def yaml_scalar(val):
    """One YAML scalar.  Numbers and None as such; strings plain when that is unambiguous,
    double-quoted (JSON escapes are valid YAML) otherwise -- a `recording_path` containing
    ': ', '#', quotes or a leading '-' / '*' must still load as the same string."""
    if val is None:
        return "null"
    if isinstance(val, bool):
        return "true" if val else "false"
    if isinstance(val, (int, float)):
        return repr(val)
    s = str(val)
    numberlike = re.match(r"^[-+.]?[0-9]", s) is not None or s.lower() in (".inf", ".nan")
    if (_PLAIN.match(s) and s.lower() not in _RESERVED and not numberlike and s == s.strip()
            and ": " not in s and " #" not in s and not s.endswith(":")):
        return s
    return json.dumps(s)


def create_header(writer, cfg):
    """Two header rows: every joint name four times, then x,y,z,confidence per joint."""
    joints = list(itertools.chain.from_iterable(itertools.repeat(x, 4) for x in cfg.KEYPOINT_NAMES))
    coords = ["x", "y", "z", "confidence"] * len(cfg.KEYPOINT_NAMES)
    writer.writerow(joints)
    writer.writerow(coords)


def create_info_file(params, keys=("recording_path", "dataset_name", "frame_start", "number_frames")):
    """info.yaml with the four keys the reference writes (predict3D.py:149-155; a block
    mapping in the reference's key order; strings quoted where YAML needs it)."""
    with open(os.path.join(params.output_dir, "info.yaml"), "w") as f:
        for key in keys:
            f.write("%s: %s\n" % (key, yaml_scalar(getattr(params, key))))


def frame_row(points3D, confidences, num_joints):
    """One CSV row: [x, y, z, confidence] per joint, or 'NaN' x 4J when the predictor
    returned (None, None).  Same element types as the reference (Python floats from
    `.tolist()`, numpy float32 confidences), hence the same text."""
    if points3D is None:
        return ["NaN"] * (num_joints * 4)
    row = []
    for point, conf in zip(points3D.squeeze(), confidences.squeeze().cpu().numpy()):
        row = row + point.tolist() + [conf]
    return row


def predict3D_frames(predictor, frame_sets, cameraMatrices, intrinsicMatrices,
                     distortionCoefficients, cfg, output_dir, params=None, time_batch=1, streams=1):
    """Run `predictor` over an iterable of multi-view frame sets -- (C,H,W,3) uint8 BGR
    arrays / tensors exactly as cv2 delivers them, or (C,3,H,W) fp32 RGB -- and write
    data3D.csv (+ info.yaml when `params` is given).  Returns the number of frames.

    time_batch > 1 groups that many consecutive frame sets into one launch sequence
    (`forward_batch`, the throughput form the bench measures); rows are written in frame
    order and are the same as with time_batch = 1 -- bit for bit up to time_batch 7, to about
    1e-5 mm from 8 on (the high-resolution BiFPN nodes then run in their row-streaming form,
    DESIGN.md section 1; a row does not depend on its position in the group or on `streams`).
    A short last group is padded with its last frame set and the padding rows are dropped.
    streams > 1 (with time_batch > 1) keeps that many groups in flight on as many HIP streams;
    rows still come out in frame order and are identical to the streams = 1 run."""
    os.makedirs(output_dir, exist_ok=True)
    if params is not None:
        params.output_dir = output_dir
        create_info_file(params)
    J = cfg.KEYPOINTDETECT.NUM_JOINTS
    calib = (cameraMatrices, intrinsicMatrices, distortionCoefficients)
    n = 0
    with open(os.path.join(output_dir, "data3D.csv"), "w", newline="") as f:
        writer = csv.writer(f, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
        names = getattr(cfg, "KEYPOINT_NAMES", [])
        if len(names) == J:
            create_header(writer, cfg)

        inflight = []                               # (outputs, event, real) of submitted groups

        def drain(keep):
            while len(inflight) > keep:
                (pts, conf, valid), ev, real = inflight.pop(0)
                ev.synchronize()
                pts, conf, valid = pts.cpu(), conf.cpu(), valid.cpu()
                for t in range(real):
                    ok = int(valid[t]) != 0
                    writer.writerow(frame_row(pts[t] if ok else None, conf[t] if ok else None, J))

        def flush(group):
            real = len(group)
            group = group + [group[-1]] * (time_batch - real)
            x = torch.stack(group).cuda()
            if streams > 1:
                h, w = (x.shape[2], x.shape[3]) if x.dtype == torch.uint8 else (x.shape[3], x.shape[4])
                msp = predictor.native_streams(h, w, time_batch, streams)
                msp.set_calibration(*calib)
                drain(streams - 1)                  # the stream about to be reused must be idle
                res = msp.forward(x)
                inflight.append((res, msp.last_event, real))
                return real
            pts, conf, valid = predictor.forward_batch(x, *calib)
            pts, conf, valid = pts.cpu(), conf.cpu(), valid.cpu()
            for t in range(real):
                ok = int(valid[t]) != 0
                writer.writerow(frame_row(pts[t] if ok else None, conf[t] if ok else None, J))
            return real

        group = []
        for frames in frame_sets:
            x = torch.as_tensor(frames)
            if time_batch > 1:
                if group and (x.dtype != group[0].dtype or x.shape != group[0].shape):
                    n += flush(group)
                    drain(0)
                    group = []
                group.append(x)
                if len(group) == time_batch:
                    n += flush(group)
                    group = []
                continue
            if x.dtype == torch.uint8:
                pts, conf = predictor.forward_uint8(x.cuda(), *calib)
            else:
                pts, conf = predictor(x.cuda(), *calib)
            writer.writerow(frame_row(pts, conf, J))
            n += 1
        if group:
            n += flush(group)
        drain(0)
    return n
