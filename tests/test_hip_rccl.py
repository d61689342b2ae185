"""GPU: the multi-GPU code path.

* `test_sharded_predictor_over_rccl`: the REAL `nccl` (= RCCL) backend at world size 1 -- ShardedPredictor in all three
  exchange / placement modes (a one-rank all-to-all, all-gather and broadcast are self-exchanges through RCCL's own
  kernels), and bench.py's camera-sharded leg.
* `test_multi_rank_rccl`: fires BY ITSELF on any box with >= 2 GPUs (world = min(device_count, 4); skipped on the
  one-GPU pool): fresh child processes, one GPU each, RCCL, three pipelined submits + flush, `torch.equal` with the
  unsharded forward and < 1e-3 mm from the reference fixture.
* `test_two_processes_one_gpu_gloo_bridge`: the same worker with two PROCESSES on the one GPU that exists, the
  collectives bounced through host memory and `gloo`: real process concurrency, real asynchronous work handles, real
  HIP stages -- what the in-process emulation of tests/local_comm.py cannot give.
Multi-rank data movement is also covered by the gloo tests (CPU, world 2 and 4) and the emulated ranks of
tests/test_hip_predictor.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests import cases
from tests.gpu_util import cuda, max_err, report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["alltoall", "allgather", "rank0"])
def test_sharded_predictor_over_rccl(mode, nccl_world1, golden):
    """cfg3 rig (12 cameras 1280x1024, 23 keypoints, 64^3), T = 2: ShardedPredictor over
    torch.distributed / nccl == the unsharded forward bit for bit; frame 0 = the reference's
    own output (fixture) within 1e-3 mm; the pipelined submit / flush form gives the same rows."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor
    from jarvis_hybridnet_amd.distributed import ShardedPredictor
    dist = nccl_world1
    assert dist.get_backend() == "nccl"
    c = cases.PREDICTOR_CASES["cfg3"]
    inp = cases.predictor_inputs("cfg3")
    calib = (inp["cam"], inp["intr"], inp["dist"])
    T, C, J = 2, c["C"], c["J"]
    frames = cuda(torch.stack([inp["imgs"], S.blob_frames(calib, c["W"], c["H"], J, 60)[0]]))
    kw = dict(num_cameras=C, num_joints=J, center_size=c["center_size"], bbox=c["bbox"],
              roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
              mean=S.MEAN, std=S.STD, time_batch=T)
    dev = [cuda(t) for t in calib]
    full = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
    full.set_calibration(*dev)
    rp, rc, rv = [t.clone() for t in full.forward(frames)]
    p = NativePredictor(inp["sd_center"], inp["sd_hybrid"], time_batch_3d=T, cam_lo=0, cam_n=C, **kw)
    p.set_calibration(*dev)
    three_d = "rank0" if mode == "rank0" else "sharded"
    sh = ShardedPredictor(p, num_cameras=C, num_joints=J, time_batch=T, heat_shape=(p.Hh, p.Hh, p.Jp),
                          rank=0, world=1, device="cuda", exchange="allgather" if mode == "rank0" else mode,
                          three_d=three_d)
    pts, conf, valid = sh.step(frames)
    assert sh.submit(frames) is None
    again = sh.submit(frames.flip(0).contiguous())
    last = sh.flush()
    torch.cuda.synchronize()
    assert torch.equal(valid, rv) and torch.equal(pts, rp) and torch.equal(conf, rc)
    assert torch.equal(again[0], rp) and torch.equal(again[1], rc)
    assert torch.equal(last[0], rp.flip(0)) and torch.equal(last[1], rc.flip(0))
    e0 = max_err(pts[0], torch.from_numpy(golden("predictor")["cfg3.points3D"])[0])
    report("sharded_rccl_world1", mode=mode, frame0_vs_reference_fixture_mm=e0)
    assert e0 < 1e-3


def test_bench_camera_sharded_leg_over_rccl():
    """bench.py --force-sharded: the multi-GPU leg of the bench (RCCL process group, two
    pipelined camera-sharded pipelines, side legs) on one rank, through the bench contract."""
    env = dict(os.environ, JH_BENCH_SIDE_LEGS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-sharded", "--steps", "2",
                          "--warmup", "1", "--time-batch", "2", "--profile-passes", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["scaling"] == "weak"
    assert "RCCL alltoall" in line["config"]["parallelism"]
    assert line["config"]["valid_frames_per_step"] == line["config"]["frames_per_step"] == 4
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] <= 1.0
    assert "value" in line["three_d_rank0"] and "value" in line["replicas_only"], line


def _run_ranks(world, transport, mode, case="cfg3", timeout=900):
    """Start `world` fresh worker processes (tests/multirank_worker.py), wait for all of them, return their JSON lines.
    On a time-out or a failure the exact child PIDs are killed (never a pattern)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   NCCL_DEBUG="WARN", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"),
                                       "--transport", transport, "--mode", mode, "--case", case],
                                      cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = []
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d exited %s\n%s\n%s" % (r, p.returncode, so[-2000:], se[-3000:])
        lines.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    return lines


def _check_lines(lines, world, transport, mode):
    assert [ln["rank"] for ln in lines] == list(range(world))
    for ln in lines:
        assert ln["ok"] and ln["equals_rank0"] and ln["world"] == world, ln
    checked = [ln for ln in lines if "frame0_vs_reference_fixture_mm" in ln]
    assert checked and checked[0]["rank"] == 0
    for ln in checked:
        assert ln["batch0_equals_unsharded"] and ln["batch1_equals_unsharded"] and ln["batch2_equals_unsharded"], ln
        assert ln["frame0_vs_reference_fixture_mm"] < 1e-3
    report("multirank_" + transport.replace("-", "_"), mode=mode, world=world,
           frame0_vs_reference_fixture_mm=checked[0]["frame0_vs_reference_fixture_mm"])


@pytest.mark.parametrize("mode", ["alltoall", "allgather", "rank0"])
def test_multi_rank_rccl(mode):
    """SURVEY 8(e) on real hardware, without a human: one fresh process per GPU over RCCL / xGMI.  cfg3 (12 cameras
    divide over 2, 3 and 4 ranks), two frame sets per rank, three back-to-back submit()s + flush()."""
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs (%d visible)" % torch.cuda.device_count())
    lines = _run_ranks(world, "rccl", mode)
    assert all(ln["backend"] == "nccl" for ln in lines)
    _check_lines(lines, world, "rccl", mode)


@pytest.mark.parametrize("mode", ["alltoall", "allgather", "rank0"])
def test_two_processes_one_gpu_gloo_bridge(mode):
    """Two processes share GPU 0, each owns six of cfg3's cameras; the exchanges go device -> host -> gloo -> host ->
    device with asynchronous work handles.  Both ranks also run the unsharded forward and compare bit for bit."""
    lines = _run_ranks(2, "gloo-bridge", mode)
    assert all(ln["backend"] == "gloo" for ln in lines)
    _check_lines(lines, 2, "gloo-bridge", mode)
