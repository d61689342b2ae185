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
                     csv_name="data2D.csv", frame_spec=None):
    """Run `predictor` (JarvisPredictor2D) over an iterable of frames -- (H,W,3) uint8 BGR arrays
    / tensors exactly as cv2 delivers them, or (3,H,W) fp32 RGB -- and write `csv_name`
    (+ info.yaml when `params` is given).  Returns the number of frames.

    Frames are staged in pinned memory by a thread pool as the iterator yields them and uploaded per
    time batch on a copy stream while the previous batch is computed (`_ingest.FramePipeline`, the
    same pipeline as predict3D_frames; an item may be a callable `fill(dst)` decoding in place, with
    `frame_spec=((H,W,3), torch.uint8)`).

    time_batch > 1 groups that many consecutive frames into one launch sequence
    (`forward_batch`); rows are written in frame order and are the same as with
    time_batch = 1.  A short last group is padded with its last frame and the padding rows
    are dropped."""
    from ._ingest import host_outputs, pipeline_for
    from .predict3D import _as_host
    os.makedirs(output_dir, exist_ok=True)
    if params is not None:
        params.output_dir = output_dir
        create_info_file(params)
    J = cfg.KEYPOINTDETECT.NUM_JOINTS
    time_batch = max(1, int(time_batch))
    n = 0
    with open(os.path.join(output_dir, csv_name), "w", newline="") as f:
        writer = csv.writer(f, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
        names = getattr(cfg, "KEYPOINT_NAMES", [])
        if len(names) == J:
            create_header(writer, cfg)

        def emit(outs, real):
            pts, conf, valid = outs
            pts = pts.long()
            for t in range(real):
                ok = int(valid[t]) != 0
                writer.writerow(frame_row(pts[t] if ok else None, conf[t] if ok else None, J))

        ring = {}                                               # pinned host copies of the outputs, per slot

        def submit(x, slot):
            # the reference driver's own conversion of uint8 frames (predict2D.py:93-94) is what the
            # uint8 entry point fuses into the resize / crop kernels
            res = predictor.forward_batch(x)
            ev = None
            if x.is_cuda:
                res = host_outputs(ring, slot, res)
                ev = torch.cuda.Event()
                ev.record()
            return res, ev

        pipe, key = None, None
        try:
            for frame in frames:
                if not callable(frame):
                    frame = frame if torch.is_tensor(frame) and frame.is_cuda else _as_host(frame)
                    k = (frame.dtype, tuple(frame.shape), torch.is_tensor(frame))
                else:
                    k = key if key is not None else ("fill",)
                if pipe is None or k != key:
                    if pipe is not None:
                        n += pipe.finish()
                    pipe, key = pipeline_for(predictor, frame, time_batch, 1, submit, emit, frame_spec), k
                pipe.push(frame)
            if pipe is not None:
                n += pipe.finish()
        except BaseException:
            from ._ingest import release_ingest_buffers
            release_ingest_buffers(predictor)           # an aborted run leaves work in flight on the cached buffers
            raise
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
