"""CPU, world_size 2, gloo: the camera-sharded multi-GPU driver
(jarvis_hybridnet_amd/distributed.py) produces exactly what the single-process
path produces.  The compute of each stage is the CPU oracle here (the HIP
stages need a GPU); what is under test is the sharding, the two exchanges, the
camera / frame re-assembly and both exchange modes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from jarvis_hybridnet_amd import synthetic as S
from jarvis_hybridnet_amd.distributed import ShardedPredictor, camera_range
from oracle import hybridnet_oracle as O

C, J, W, H, BBOX, CENTER, ROI, SP, T = 4, 3, 320, 256, 128, 128, 32, 2, 2
JP = 8
KW = dict(center_size=CENTER, bbox=BBOX, roi_cube_size=ROI, grid_spacing=SP, mean=S.MEAN, std=S.STD)


def make_inputs(T=T):
    calib = S.ring_calibration(C, W, H, 450.0)
    sd_c = S.efficienttrack_weights("small", 1, 80)
    sd_h = S.hybridnet_weights("small", J, 81)
    frames = torch.stack([S.blob_frames(calib, W, H, J, 82 + t)[0] for t in range(T)])
    return calib, sd_c, sd_h, frames


class OracleStages:
    """The three stage calls of NativePredictor, computed by the oracle on CPU."""

    def __init__(self, calib, sd_c, sd_h, cam_lo, cam_n):
        self.calib, self.sd_c, self.sd_h, self.lo, self.n = calib, sd_c, sd_h, cam_lo, cam_n
        self.mean = torch.tensor(S.MEAN).view(3, 1, 1)
        self.std = torch.tensor(S.STD).view(3, 1, 1)
        self.state = {}

    def stage_center(self, frames, det):
        with torch.no_grad():
            for t in range(frames.shape[0]):
                small = F.interpolate(frames[t], size=[CENTER, CENTER], mode="bilinear",
                                      align_corners=False)
                hm = O.efficienttrack_forward(self.sd_c, (small - self.mean) / self.std, "small",
                                              want_res1=False)[1]
                flat = hm.view(hm.shape[0], -1)
                m = flat.argmax(1)
                det[t, :, 0] = (m % hm.shape[2]).float()
                det[t, :, 1] = (m // hm.shape[3]).float()
                det[t, :, 2] = flat.gather(1, m[:, None])[:, 0]

    def stage_keypoints(self, frames, det_all, heat):
        cam, intr, dist_ = self.calib
        hw = BBOX // 2
        scale = torch.tensor([W / float(CENTER), H / float(CENTER)]).float()
        with torch.no_grad():
            for t in range(frames.shape[0]):
                preds, maxv = det_all[t, :, :2], det_all[t, :, 2]
                self.state[("valid", t)] = int((maxv > 50).sum() >= 2)
                c3 = O.reconstruct_point((preds * (scale * 2)).transpose(0, 1),
                                         (maxv / 255.).view(-1, 1, 1), cam, intr, dist_)
                chm = O.reproject_point(c3.unsqueeze(0), cam, intr, dist_).int()
                chm[:, 0] = chm[:, 0].clamp(hw, W - hw)
                chm[:, 1] = chm[:, 1].clamp(hw, H - hw)
                self.state[("c3", t)], self.state[("chm", t)] = c3.int(), chm
                crops = torch.stack([
                    frames[t, i, :, int(chm[self.lo + i, 1]) - hw:int(chm[self.lo + i, 1]) + hw,
                           int(chm[self.lo + i, 0]) - hw:int(chm[self.lo + i, 0]) + hw]
                    for i in range(self.n)])
                hm = O.efficienttrack_forward(self.sd_h, (crops - self.mean) / self.std, "small",
                                              "effTrack.", want_res1=False)[1]
                heat[t].zero_()
                heat[t, :, :, :, :J] = hm.permute(0, 2, 3, 1)       # channel-last, Jp padded

    # the layouts the collectives deliver (what the library's *_gathered / *_blocks entry points
    # read in place): re-ordered here with torch, then the plain stage
    def stage_keypoints_gathered(self, frames, det_gathered, n_blocks, heat):
        Tn = frames.shape[0]
        det_all = det_gathered.view(n_blocks, Tn, -1, 3).permute(1, 0, 2, 3).reshape(Tn, C, 3)
        self.stage_keypoints(frames, det_all, heat)

    def stage_3d_blocks(self, heat_blocks, n_blocks, frames_per_block, t_off, t0, pts, conf, valid):
        hb = heat_blocks.view((n_blocks, frames_per_block) + tuple(heat_blocks.shape[1:]))
        T3 = pts.shape[0]
        mine = hb[:, t_off:t_off + T3].permute(1, 0, 2, 3, 4, 5)
        self.stage_3d(mine.reshape((T3, C) + tuple(hb.shape[3:])), t0, pts, conf, valid)

    def stage_3d(self, heat_all, t0, pts, conf, valid):
        cam, intr, dist_ = self.calib
        with torch.no_grad():
            for i in range(heat_all.shape[0]):
                t = t0 + i
                hm = heat_all[i, :, :, :, :J].permute(0, 3, 1, 2)[None]      # (1,C,J,h,w)
                hm_pad = F.pad(hm, [1, 1, 1, 1])
                vol = O.reprojection_forward(hm_pad, self.state[("c3", t)][None],
                                             self.state[("chm", t)][None], cam[None], intr[None],
                                             dist_[None], ROI, SP)
                out = O.v2v_forward(self.sd_h, vol / 255., "v2vNet.")
                _, p, cf = O.softargmax_tail(out, self.state[("c3", t)][None], ROI, SP)
                pts[i], conf[i], valid[i] = p[0], cf[0], self.state[("valid", t)]


def _worker(rank, world, port, exchange, q, T=T, three_d="sharded", gs=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calib, sd_c, sd_h, frames = make_inputs(T if gs is None else T * (world // gs))
    group, grank, gworld, gidx = None, rank, world, 0
    if gs is not None:
        # the 8-GPU layout of bench.py in small: world // gs groups of gs ranks, each group shards the
        # cameras of ITS time batch (every rank creates every group, as torch.distributed requires)
        groups = [dist.new_group(list(range(g * gs, (g + 1) * gs))) for g in range(world // gs)]
        gidx, grank, gworld = rank // gs, rank % gs, gs
        group = groups[gidx]
        frames = frames[gidx * T:(gidx + 1) * T]
    lo, n = camera_range(C, grank, gworld)
    st = OracleStages(calib, sd_c, sd_h, lo, n)
    sh = ShardedPredictor(st, num_cameras=C, num_joints=J, time_batch=T,
                          heat_shape=(BBOX // 2, BBOX // 2, JP), rank=grank, world=gworld,
                          device="cpu", exchange=exchange, three_d=three_d, group=group)
    mine = frames[:, lo:lo + n].contiguous()
    pts, conf, valid = sh.step(mine)
    # pipelined form: batch B (frames in reverse order) is submitted while batch A is in
    # flight; every batch must come out as the unpipelined step computes it
    rev = mine.flip(0).contiguous()
    assert sh.submit(mine) is None
    a = sh.submit(rev)
    b = sh.flush()
    assert sh.flush() is None
    for x, y in zip(a, (pts, conf, valid)):
        assert torch.equal(x, y), "pipelined batch A differs from step()"
    for x, y in zip(b, (pts.flip(0), conf.flip(0), valid.flip(0))):
        assert torch.equal(x, y), "pipelined batch B differs from step() on the same frames"
    if grank == 0:
        q.put((pts.clone(), conf.clone(), valid.clone()) if gs is None else
              (gidx, pts.clone(), conf.clone(), valid.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,world,three_d", [("alltoall", 2, "sharded"),
                                                    ("allgather", 2, "sharded"),
                                                    ("alltoall", 4, "sharded"),
                                                    ("allgather", 2, "rank0")])
def test_camera_sharded_equals_single_process(exchange, world, three_d):
    """world 4 = one camera per rank, the group size the 4- and 8-GPU runs use (12 cameras -> 3
    per rank there).  three_d='rank0' = the literal placement of BASELINE configs[3]: heatmaps
    all-gathered, 3D stage on rank 0, results broadcast."""
    T = 2 if world == 2 else 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, exchange, q, T, three_d))
             for r in range(world)]
    for p in procs:
        p.start()
    pts, conf, valid = q.get(timeout=240)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    calib, sd_c, sd_h, frames = make_inputs(T)
    for t in range(T):
        with torch.no_grad():
            rp, rc = O.predictor3d_forward(sd_c, sd_h, frames[t], *calib, **KW)
        assert int(valid[t]) == (rp is not None)
        if rp is not None:
            # collectives only move data.  (Not bit-equal here only because oneDNN picks
            # other blockings for 2-image than for 4-image CPU batches; the HIP kernels
            # treat images as independent instances, see test_predictor3d_time_batch.)
            assert (pts[t] - rp[0]).abs().max().item() < 1e-3
            assert (conf[t] - rc[0]).abs().max().item() < 1e-5


@pytest.mark.parametrize("three_d", ["sharded", "rank0"])
def test_two_groups_of_two_ranks(three_d):
    """The layout `bench.py --gpus 8` uses for 12 cameras (2 groups of 4 GPUs), in small: world 4 = 2 groups of
    2 ranks, each group camera-shards its own time batch over a sub-group communicator.  three_d='rank0'
    broadcasts from the GROUP's first rank (global rank 2 in the second group)."""
    world, gs = 4, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "alltoall", q, T, three_d, gs))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict((g, (a, b, c)) for g, a, b, c in (q.get(timeout=240) for _ in range(world // gs)))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    calib, sd_c, sd_h, frames = make_inputs(T * (world // gs))
    for g in range(world // gs):
        pts, conf, valid = got[g]
        for t in range(T):
            with torch.no_grad():
                rp, rc = O.predictor3d_forward(sd_c, sd_h, frames[g * T + t], *calib, **KW)
            assert int(valid[t]) == (rp is not None)
            if rp is not None:
                assert (pts[t] - rp[0]).abs().max().item() < 1e-3
                assert (conf[t] - rc[0]).abs().max().item() < 1e-5


def test_plan_groups():
    from jarvis_hybridnet_amd.distributed import plan_groups, frame_range
    assert plan_groups(1, 12) == (1, 1)
    assert plan_groups(2, 12) == (2, 1)
    assert plan_groups(4, 12) == (4, 1)
    assert plan_groups(8, 12) == (4, 2)          # 12 cameras do not divide over 8 GPUs
    assert plan_groups(8, 16) == (8, 1)
    assert plan_groups(5, 12) == (1, 5)
    for w in (1, 2, 4):
        lo = [camera_range(12, r, w) for r in range(w)]
        assert sum(n for _, n in lo) == 12 and [a for a, _ in lo] == [r * (12 // w) for r in range(w)]
        assert [frame_range(64, r, w)[0] for r in range(w)] == [r * (64 // w) for r in range(w)]


@pytest.mark.parametrize("exchange,three_d", [("alltoall", "sharded"), ("allgather", "rank0")])
def test_local_comm_matches_gloo_semantics(exchange, three_d):
    """tests/local_comm.py (threads + copies; what the one-GPU multi-rank emulation of
    tests/test_hip_predictor.py uses instead of RCCL) drives ShardedPredictor to the same result
    as the single-process oracle, like the gloo processes above."""
    from tests.local_comm import LocalWorld
    world, Tn = 2, 2
    calib, sd_c, sd_h, frames = make_inputs(Tn)

    def rank_fn(rank, comm):
        lo, n = camera_range(C, rank, world)
        st = OracleStages(calib, sd_c, sd_h, lo, n)
        sh = ShardedPredictor(st, num_cameras=C, num_joints=J, time_batch=Tn,
                              heat_shape=(BBOX // 2, BBOX // 2, JP), rank=rank, world=world,
                              device="cpu", exchange=exchange, three_d=three_d, comm=comm)
        return sh.step(frames[:, lo:lo + n].contiguous())
    res = LocalWorld(world).run(rank_fn)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)                       # every rank holds the full result
    pts, conf, valid = res[0]
    for t in range(Tn):
        with torch.no_grad():
            rp, rc = O.predictor3d_forward(sd_c, sd_h, frames[t], *calib, **KW)
        assert int(valid[t]) == (rp is not None)
        if rp is not None:
            assert (pts[t] - rp[0]).abs().max().item() < 1e-3
            assert (conf[t] - rc[0]).abs().max().item() < 1e-5
