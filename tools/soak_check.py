"""Soak check at the bench's scale: a time batch of 32 frame sets of configs[2] (384 images per 2D launch)
run repeatedly on three concurrent streams; every frame of every run of every stream must reproduce the first
frame of the first run bit for bit, and lie within 1e-4 mm of the single-frame call's result (bit-equal to it with
JH_NODE_ROWS=0: a time batch of >= 8 frames takes the row-streaming BiFPN nodes, the single-frame call the tile
form)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from jarvis_hybridnet_amd import synthetic as S
from jarvis_hybridnet_amd._predictor import NativePredictor
from tests import cases

CASE = os.environ.get("CASE", "cfg3")
c = cases.PREDICTOR_CASES[CASE]
inp = cases.predictor_inputs(CASE)
T, REPS, STREAMS = int(os.environ.get("T", "32")), int(os.environ.get("REPS", "20")), 3
kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center_size"], bbox=c["bbox"],
          roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"], mean=S.MEAN, std=S.STD,
          center_model=c.get("size", "small"), kp_model=c.get("size", "small"))
dev = [t.cuda() for t in (inp["cam"], inp["intr"], inp["dist"])]
one = inp["imgs"].cuda().unsqueeze(0).contiguous()
p1 = NativePredictor(inp["sd_center"], inp["sd_hybrid"], time_batch=1, **kw)
p1.set_calibration(*dev)
ref = [t.clone() for t in p1.forward(one)]
frames = one.expand(T, *one.shape[1:]).contiguous()
streams = [torch.cuda.Stream() for _ in range(STREAMS)]
preds = []
for s in streams:
    with torch.cuda.stream(s):
        p = NativePredictor(inp["sd_center"], inp["sd_hybrid"], time_batch=T, **kw)
        p.set_calibration(*dev)
        preds.append(p)
torch.cuda.synchronize()
bad = 0
class_bad = 0
first = None
strict = os.environ.get("JH_NODE_ROWS") == "0" or T < 8
for r in range(REPS):
    outs = []
    for s, p in zip(streams, preds):
        with torch.cuda.stream(s):
            outs.append([t.clone() for t in p.forward(frames)])
    torch.cuda.synchronize()
    for o in outs:
        if first is None:
            first = (o[0][0].clone(), o[1][0].clone())
            d = float((first[0] - ref[0][0]).abs().max())
            print("soak: batch vs single-frame call: %.3g mm%s" % (d, " (bit-equal required)" if strict else ""))
            # the two time-batch classes agree to 2-6e-5 mm (small, medium), 1.1-1.8e-4 mm (large): the bound the fixture
            # test holds (tests/test_hip_predictor.py::test_predictor3d_time_batch_8_vs_fixture) -- counted apart from
            # the bit-equality of the runs
            if d > 3e-4 or (strict and not (torch.equal(first[0], ref[0][0]) and torch.equal(first[1], ref[1][0]))):
                class_bad = 1
        for t in range(T):
            if not (torch.equal(o[0][t], first[0]) and torch.equal(o[1][t], first[1])):
                bad += 1
print("soak: %d runs x %d streams x %d frames, mismatching frames: %d%s" % (
    REPS, STREAMS, T, bad, "; batch vs single-frame call OUT OF BOUND" if class_bad else ""))
sys.exit(1 if bad or class_bad else 0)
