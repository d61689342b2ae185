"""Output side of the predict2D driver (mirrors jarvis/prediction/predict2D.py:22-27,
60-125): `data2D.csv` rows -- three columns per joint -- and
`info.yaml`, so that the reference's visualisation tools consume the results unchanged.

Video decoding (cv2.VideoCapture) and project management are outside the hot path:
`predict2D_frames` takes any iterable of decoded frames, `predict2D_recordings` a mapping
of recording path -> frames and applies the reference's per-video file naming.
"""
import csv
import itertools
import os

import torch

from .predict3D import create_info_file as _create_info_file


def create_header(writer, cfg):
    """Two header rows: every joint name three times, then x,y,confidence per joint
    (predict2D.py:120-125)."""
    joints = list(itertools.chain.from_iterable(itertools.repeat(x, 3) for x in cfg.KEYPOINT_NAMES))
    coords = ["x", "y", "confidence"] * len(cfg.KEYPOINT_NAMES)
    writer.writerow(joints)
    writer.writerow(coords)


def create_info_file(params):
    """info.yaml of a 2D prediction run: recording_path, frame_start, number_frames
    (predict2D.py:22-27 -- no dataset_name, unlike the 3D driver)."""
    _create_info_file(params, keys=("recording_path", "frame_start", "number_frames"))


def csv_filename(recording_path, multiple_videos):
    """`data2D.csv`, prefixed with the video's base name (up to its first '.') when a directory
    of recordings is processed (predict2D.py:64-68)."""
    name = "data2D.csv"
    if multiple_videos:
        name = "%s_%s" % (recording_path.split(os.sep)[-1].split(".")[0], name)
    return name


def frame_row(points2D, confidences, num_joints):
    """One CSV row: [x, y, confidence] per joint (x, y Python ints from `.tolist()` of the
    int64 pixel coordinates, numpy float32 confidences: the reference's element types, hence
    its text), or 'NaN' x 3J when the predictor returned (None, None) (predict2D.py:96-109)."""
    if points2D is None:
        return ["NaN"] * (num_joints * 3)
    pts = points2D.cpu().numpy()
    conf = confidences.cpu().numpy()
    row = []
    for i, point in enumerate(pts):
        row = row + point.tolist() + [conf[i]]
    return row


def predict2D_frames(predictor, frames, cfg, output_dir, params=None, time_batch=1,
                     csv_name="data2D.csv"):
    """Run `predictor` (JarvisPredictor2D) over an iterable of frames -- (H,W,3) uint8 BGR arrays
    / tensors exactly as cv2 delivers them, or (3,H,W) fp32 RGB -- and write `csv_name`
    (+ info.yaml when `params` is given).  Returns the number of frames.

    time_batch > 1 groups that many consecutive frames into one launch sequence
    (`forward_batch`); rows are written in frame order and are the same as with
    time_batch = 1.  A short last group is padded with its last frame and the padding rows
    are dropped."""
    os.makedirs(output_dir, exist_ok=True)
    if params is not None:
        params.output_dir = output_dir
        create_info_file(params)
    J = cfg.KEYPOINTDETECT.NUM_JOINTS
    n = 0
    with open(os.path.join(output_dir, csv_name), "w", newline="") as f:
        writer = csv.writer(f, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
        names = getattr(cfg, "KEYPOINT_NAMES", [])
        if len(names) == J:
            create_header(writer, cfg)

        def flush(group):
            real = len(group)
            x = torch.stack(group + [group[-1]] * (time_batch - real)).cuda()
            pts, conf, valid = predictor.forward_batch(x)
            pts, conf, valid = pts.long().cpu(), conf.cpu(), valid.cpu()
            for t in range(real):
                ok = int(valid[t]) != 0
                writer.writerow(frame_row(pts[t] if ok else None, conf[t] if ok else None, J))
            return real

        group = []
        for frame in frames:
            x = torch.as_tensor(frame)
            if time_batch > 1:
                if group and (x.dtype != group[0].dtype or x.shape != group[0].shape):
                    n += flush(group)
                    group = []
                group.append(x)
                if len(group) == time_batch:
                    n += flush(group)
                    group = []
                continue
            if x.dtype == torch.uint8:
                # the reference driver's own conversion (predict2D.py:93-94) is what the uint8
                # entry point fuses into the resize / crop kernels
                pts, conf, valid = predictor.forward_batch(x.unsqueeze(0).cuda())
                ok = int(valid[0].item()) != 0
                pts, conf = (pts[0].long(), conf[0]) if ok else (None, None)
            else:
                pts, conf = predictor(x.unsqueeze(0).cuda())
            writer.writerow(frame_row(pts, conf, J))
            n += 1
        if group:
            n += flush(group)
    return n


def predict2D_recordings(predictor, recordings, cfg, output_dir, params=None, time_batch=1):
    """`recordings`: {recording path: iterable of frames}.  One CSV per recording, named as the
    reference names them: `data2D.csv` for a single file, `<video>_data2D.csv` per video of a
    directory (predict2D.py:49-68).  Returns {csv file name: number of frames}."""
    multiple = len(recordings) > 1 or bool(getattr(params, "multiple_videos", False))
    done = {}
    for i, (path, frames) in enumerate(recordings.items()):
        name = csv_filename(path, multiple)
        done[name] = predict2D_frames(predictor, frames, cfg, output_dir, params if i == 0 else None,
                                      time_batch, name)
    return done
