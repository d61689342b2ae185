"""One rank of a multi-PROCESS run of the camera-sharded hot path (tests/test_hip_rccl.py starts `world` of these as
fresh child processes: a process that has touched the GPU is never re-executed).

    python tests/multirank_worker.py --transport rccl|gloo-bridge --mode alltoall|allgather|rank0 --case cfg3

RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT come from the environment, as under torch.distributed.run.

transport rccl         one GPU per rank, backend `nccl` (= RCCL over xGMI): what `bench.py --gpus N` runs.  Needs
                       world <= torch.cuda.device_count().
transport gloo-bridge  every rank on GPU 0 (the pool's boxes have one): real HIP stages in concurrent processes, real
                       asynchronous work handles, the collectives bounced through host memory and the `gloo` backend
                       (`GlooBridge` below, injected as ShardedPredictor's `comm`).  Data movement and ordering of
                       distributed.py as on N GPUs; only the wire differs.

Every rank: three back-to-back submit()s (frame sets A, B, A) + flush() of a ShardedPredictor over its share of the
cameras; every returned batch must equal, bit for bit, the UNSHARDED forward of the same frames (computed on every
rank in gloo-bridge mode, on rank 0 otherwise) and frame 0 must be within 1e-3 mm of the reference's fixture.
Prints one JSON line and exits 0 on success.
"""
import argparse
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class _BridgeWork:
    def __init__(self, work, out, host_out):
        self.work, self.out, self.host_out = work, out, host_out

    def wait(self):
        """Like ProcessGroupNCCL's Work.wait(): afterwards the CURRENT stream sees the result."""
        self.work.wait()
        self.out.copy_(self.host_out, non_blocking=False)
        return True


class GlooBridge:
    """The collectives ShardedPredictor calls, for device tensors, over a `gloo` process group: device -> host on the
    caller's current stream (the input is complete when the copy returns), gloo collective on the host tensors
    (asynchronous: a real Work handle that completes while the caller goes on), host -> device in wait()."""

    def _run(self, fn, out, inp, group, async_op):
        h_in = inp.cpu()
        h_out = torch.empty(out.shape, dtype=out.dtype)
        w = _BridgeWork(fn(h_out, h_in, group=group, async_op=True), out, h_out)
        if not async_op:
            w.wait()
        return w

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        return self._run(dist.all_gather_into_tensor, out, inp, group, async_op)

    def all_to_all_single(self, out, inp, group=None, async_op=False):
        return self._run(dist.all_to_all_single, out, inp, group, async_op)

    def broadcast(self, tensor, src, group=None, async_op=False):
        h = tensor.cpu()
        w = _BridgeWork(dist.broadcast(h, src, group=group, async_op=True), tensor, h)
        if not async_op:
            w.wait()
        return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--transport", choices=["rccl", "gloo-bridge"], required=True)
    ap.add_argument("--mode", choices=["alltoall", "allgather", "rank0"], required=True)
    ap.add_argument("--case", default="cfg3")
    ap.add_argument("--frames-per-rank", type=int, default=2)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    timeout = datetime.timedelta(seconds=600)
    if args.transport == "rccl":
        assert torch.cuda.device_count() >= world, "one GPU per rank"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
        comm = None                                   # torch.distributed itself
    else:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
        comm = GlooBridge()

    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor
    from jarvis_hybridnet_amd.distributed import ShardedPredictor, camera_range
    from tests import cases
    c = cases.PREDICTOR_CASES[args.case]
    inp = cases.predictor_inputs(args.case)
    calib = (inp["cam"], inp["intr"], inp["dist"])
    C, J = c["C"], c["J"]
    T = args.frames_per_rank * world
    size = c.get("size", "small")
    sets = [inp["imgs"]] + [S.blob_frames(calib, c["W"], c["H"], J, 60 + i)[0] for i in range(T - 1)]
    A = torch.stack(sets).to(dev)
    B = A.flip(0).contiguous()
    kw = dict(num_cameras=C, num_joints=J, center_size=c["center_size"], bbox=c["bbox"],
              roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
              mean=S.MEAN, std=S.STD, time_batch=T, center_model=size, kp_model=size)
    calib_dev = [t.to(dev) for t in calib]
    lo, n = camera_range(C, rank, world)
    three_d = "rank0" if args.mode == "rank0" else "sharded"
    t3 = T if three_d == "rank0" else T // world
    p = NativePredictor(inp["sd_center"], inp["sd_hybrid"], time_batch_3d=t3, cam_lo=lo, cam_n=n, **kw)
    p.set_calibration(*calib_dev)
    sh = ShardedPredictor(p, num_cameras=C, num_joints=J, time_batch=T, heat_shape=(p.Hh, p.Hh, p.Jp),
                          rank=rank, world=world, device=dev, three_d=three_d, comm=comm,
                          exchange="allgather" if args.mode == "rank0" else args.mode)
    mineA, mineB = A[:, lo:lo + n].contiguous(), B[:, lo:lo + n].contiguous()
    got = [sh.submit(mineA), sh.submit(mineB), sh.submit(mineA), sh.flush()]
    assert got[0] is None and sh.flush() is None
    torch.cuda.synchronize()
    out = dict(rank=rank, world=world, transport=args.transport, mode=args.mode, case=args.case, time_batch=T,
               backend=dist.get_backend())
    ok = True
    if rank == 0 or args.transport == "gloo-bridge":
        full = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
        full.set_calibration(*calib_dev)
        refA = [t.clone() for t in full.forward(A)]
        refB = [t.clone() for t in full.forward(B)]
        torch.cuda.synchronize()
        for i, (g, r) in enumerate(zip(got[1:], (refA, refB, refA))):
            same = all(torch.equal(x, y) for x, y in zip(g, r))
            out["batch%d_equals_unsharded" % i] = bool(same)
            ok = ok and same
        assert int(refA[2].sum()) >= T - 1, "the seeded frame sets are expected to be valid"
        gold = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))[args.case + ".points3D"]
        e0 = float(np.abs(got[1][0][0].cpu().numpy() - gold[0]).max())
        out["frame0_vs_reference_fixture_mm"] = e0
        ok = ok and e0 < 1e-3
    # every rank holds the full result: the ranks must agree with each other too (host-side check over gloo / RCCL)
    mine = torch.cat([g[0].flatten() for g in got[1:]]).cpu()
    ref0 = mine.clone()
    if args.transport == "rccl":
        r0 = ref0.to(dev)
        dist.broadcast(r0, 0)
        ref0 = r0.cpu()
    else:
        dist.broadcast(ref0, 0)
    out["equals_rank0"] = bool(torch.equal(mine, ref0))
    ok = ok and out["equals_rank0"]
    out["ok"] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
