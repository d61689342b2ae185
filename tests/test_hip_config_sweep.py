"""GPU: a randomised sweep over configurations no fixture covers (camera counts 2..9, joint counts 1..31, frame sizes
that are not multiples of anything, bounding boxes 128..320, CenterDetect sizes 128..320, grid spacings 1 / 2 / 4 with
16..56 voxel grids, time batches 1 / 2 / 8) against the CPU oracle run on this host (tools/config_sweep.py): validity
must agree, and on frames whose gather indices equal the host oracle's the 3D keypoints must agree to 1e-3 mm."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 7])
def test_config_sweep(seed):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config_sweep.py"), "10", str(seed)], cwd=ROOT,
                         capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0 and "0 of 10 cases failed" in res.stdout, res.stdout[-3000:] + res.stderr[-1500:]


def test_frames_beyond_2gb_per_time_batch():
    """4K cameras, time batch 8: 2.4 GB of fp32 frames in one call (byte offsets past 2^31 inside the frame buffer)
    through the HIP path against the CPU oracle, plus the single-frame and uint8 entry points on the same frames
    (tools/big_frame_check.py)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_frame_check.py"), "3840", "2160", "3", "8"],
                         cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0 and "big frame check: worst" in res.stdout, res.stdout[-3000:] + res.stderr[-1500:]
