"""Camera-sharded multi-GPU execution of the hot path (one process per GPU,
torch.distributed with the RCCL backend over xGMI).

The reference has no distributed code at all; this is
new design for the 8 x MI355X node:

  rank r owns cameras [r*C/N, (r+1)*C/N) for the 2D work of all T frames of a
  time batch, and frames [r*T/N, (r+1)*T/N) for the 3D work.

  stage 1   resize + CenterDetect + argmax on the owned cameras
  exchange  all-gather of (x, y, maxval) per camera           (T*C*12 bytes)
  stage 2   every rank triangulates redundantly (identical inputs, so identical
            results, no broadcast), crops + KeypointDetect on the owned cameras
  exchange  heatmaps: rank r needs all cameras of ITS frames, so the exchange is
            an all-to-all of (T/N, C/N, h, w, Jp) blocks -- 1/N of the volume of
            the all-gather it replaces.  xGMI is a full mesh of point-to-point
            links, so every block travels one hop on its own link.  The literal
            all-gather (every rank receives everything) is kept as
            exchange='allgather' for comparison.
  stage 3   reprojection + V2V + soft-argmax for the owned frames
  gather    (T/N, J, 4) results to every rank (tiny all-gather)

submit() / flush() pipeline consecutive time batches: the bulk exchange of batch i runs under the
CenterDetect stage of batch i+1, and stage 3 of batch i runs on a SECOND HIP stream under stage 2
(KeypointDetect) of batch i+1 -- the two share no buffer (the library keeps two sets of crop centres
for exactly this: a stage-3 call reads the set of the stage-2 call that preceded it in host call
order, include/jarvis_hip.h) and an event pair orders them against the exchanges; step() is
submit() + flush().

three_d='rank0' is the literal placement of BASELINE.json's north star / configs[3]:
"RCCL all-gather of heatmaps, 3D stage on rank 0" -- the heatmaps of ALL frames are
all-gathered, rank 0 alone runs reprojection + V2V + soft-argmax for the whole time batch
and the (T, J, 4) results are broadcast.  It is Amdahl-bound by construction (V2V is 64 % of
the FLOPs of a frame at configs[2]), which is why the default shards
the 3D stage by frame; bench.py reports both.

`comm` is the object the collectives are called on (default: torch.distributed).  Tests
inject an in-process communicator to run several emulated ranks on one GPU.

`stages` is any object with the three methods used below (stage_center,
stage_keypoints_gathered, stage_3d_blocks), which is what lets the world_size-2 gloo test on
CPU drive this file with the oracle as compute.

No torch compute op runs between the stages: the detections and the heatmaps are consumed
in the layout the collectives deliver them in -- (ranks, frames, cameras per rank, ...) --
by the *_gathered / *_blocks entry points of the library, and points + confidences leave the
3D stage inside one flat buffer that is all-gathered as it is.
"""
import torch
import torch.distributed as dist


def plan_groups(world, num_cameras):
    """Ranks form groups of `gs` GPUs that shard the cameras of a time batch; world // gs such
    groups run side by side on different frames.  gs = the largest divisor of the world size
    that divides the camera count (12 cameras: 2 -> 2, 4 -> 4, 8 -> 2 groups of 4)."""
    gs = max(d for d in range(1, world + 1) if world % d == 0 and num_cameras % d == 0)
    return gs, world // gs


def camera_range(num_cameras, rank, world):
    assert num_cameras % world == 0, "cameras must divide evenly over the ranks"
    n = num_cameras // world
    return rank * n, n


def frame_range(time_batch, rank, world):
    assert time_batch % world == 0, "time batch must divide evenly over the ranks"
    n = time_batch // world
    return rank * n, n


class ShardedPredictor:
    def __init__(self, stages, *, num_cameras, num_joints, time_batch, heat_shape, rank, world,
                 device, exchange="alltoall", group=None, three_d="sharded", comm=None):
        """heat_shape = (h, w, Jp) of one camera's channel-last heatmap.
        three_d: 'sharded' (rank r runs the 3D stage of frames [r*T/N, (r+1)*T/N)) or 'rank0'
        (rank 0 runs it for all T frames; `stages` of rank 0 must then be built for
        time_batch_3d = T)."""
        assert three_d in ("sharded", "rank0") and exchange in ("alltoall", "allgather")
        self.st, self.C, self.J, self.T = stages, num_cameras, num_joints, time_batch
        self.rank, self.world, self.group = rank, world, group
        self.three_d = three_d
        # the rank-0 placement needs every frame's heatmaps on rank 0: the all-gather
        self.exchange = exchange = "allgather" if three_d == "rank0" else exchange
        self.comm = comm if comm is not None else dist
        self.cam_lo, self.Cloc = camera_range(num_cameras, rank, world)
        if three_d == "rank0":
            self.t_lo, self.T3 = 0, time_batch
        else:
            self.t_lo, self.T3 = frame_range(time_batch, rank, world)
        f32 = dict(device=device, dtype=torch.float32)
        self.det_local = torch.empty((self.T, self.Cloc, 3), **f32)
        # gather outputs are allocated in the "concatenated along dim 0" form that both
        # RCCL and gloo accept, and viewed as (world, ...) afterwards
        self.det_gather = torch.empty((world * self.T, self.Cloc, 3), **f32)
        self.heat_local = torch.empty((self.T, self.Cloc) + tuple(heat_shape), **f32)
        if exchange == "alltoall":
            self.heat_recv = torch.empty((world * self.T3, self.Cloc) + tuple(heat_shape), **f32)
        else:
            self.heat_recv = torch.empty((world * self.T, self.Cloc) + tuple(heat_shape), **f32)
        # results: [points (T3,J,3) | conf (T3,J)] in ONE flat buffer per rank, written in place by
        # the 3D stage and gathered as it is
        self.n_pts = self.T3 * self.J * 3
        self.res_local = torch.zeros((self.T3 * self.J * 4,), **f32)
        self.pts_local = self.res_local[:self.n_pts].view(self.T3, self.J, 3)
        self.conf_local = self.res_local[self.n_pts:].view(self.T3, self.J)
        self.valid_local = torch.zeros((self.T3,), device=device, dtype=torch.int32)
        n_res = 1 if three_d == "rank0" else world
        self.res_all = torch.empty((n_res * self.T3 * self.J * 4,), **f32)
        self.valid_all = torch.empty((n_res * self.T3,), device=device, dtype=torch.int32)
        self._pending = None                     # exchange of the time batch in flight
        # stage 3 (+ the result collectives) of batch i on its own stream, under stage 2 of batch i+1
        self.cuda = torch.device(device).type == "cuda"
        self.s3d = torch.cuda.Stream(device=device) if self.cuda else None
        self._done3d = None                      # event: stage 3 of the last finished batch has read heat_recv
        self._exchanged = None                   # event: the exchange in flight has completed (read its send buffer)
        self.trace = None                        # list of (label, timing event) while a timeline is being recorded

    def _mark(self, label):
        """Timeline stamp on the current stream (bench.py: stage_ms_per_rank); free when no trace is recorded."""
        if self.trace is not None and self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((label, ev))

    def step(self, frames_local):
        """frames_local (T, Cloc, 3, H, W) -> points (T,J,3), conf (T,J), valid (T)."""
        self.submit(frames_local)
        return self.flush()

    # ---- pipelined form: the bulk exchange of time batch i runs under the CenterDetect
    # stage of time batch i+1 (which touches neither the heatmap buffers nor the crop
    # centres the 3D stage of batch i still needs).  Order of one submit():
    #     stage_center(i+1) || exchange(i)  ->  stage_3d(i) [second stream] || stage_keypoints(i+1)
    #     -> exchange(i+1) started asynchronously once stage_3d(i) has let go of the receive buffer
    def submit(self, frames_local):
        """Start time batch i+1; returns the results of batch i (None on the first call)."""
        W = self.world
        self._mark("submit")
        self.st.stage_center(frames_local, self.det_local)
        self._mark("stage1_end")
        self.comm.all_gather_into_tensor(self.det_gather, self.det_local, group=self.group)
        self._mark("det_gathered")
        prev = self._finish()
        # det_gather is (W, T, Cl, 3): read in place, camera = source rank * Cl + local camera.  heat_local is the
        # SEND buffer of exchange(i): _finish() has made this stream wait for that exchange's completion.
        self.st.stage_keypoints_gathered(frames_local, self.det_gather, W, self.heat_local)
        self._mark("stage2_end")
        if self._done3d is not None:
            # exchange(i+1) overwrites the receive buffer stage_3d(i) reads; the results returned above are
            # stream-ordered for the caller from here on too
            torch.cuda.current_stream().wait_event(self._done3d)
            self._done3d = None
        if self.exchange == "alltoall":
            # block r of the send buffer = my cameras' heatmaps of rank r's frames
            self._pending = self.comm.all_to_all_single(self.heat_recv, self.heat_local,
                                                        group=self.group, async_op=True)
        else:
            self._pending = self.comm.all_gather_into_tensor(self.heat_recv, self.heat_local,
                                                             group=self.group, async_op=True)
        self._mark("exchange_issued")
        return prev

    def flush(self):
        """Results of the last submitted time batch (None if nothing is in flight)."""
        res = self._finish()
        if self._done3d is not None:
            torch.cuda.current_stream().wait_event(self._done3d)
            self._done3d = None
        return res

    def _finish(self):
        """Stage 3 + result collectives of the batch in flight.  On a GPU they are issued on the second stream
        (every rank issues its collectives in the same program order, whatever the stream); the caller's stream
        picks the results up through the `_done3d` event."""
        if self._pending is None:
            return None
        if not self.cuda:
            return self._finish_on_current_stream()
        cur = torch.cuda.current_stream()
        self.s3d.wait_stream(cur)                # (result buffers of the previous batch have been cloned there)
        with torch.cuda.stream(self.s3d):
            res = self._finish_on_current_stream()
            self._done3d = torch.cuda.Event()
            self._done3d.record(self.s3d)
        # exchange(i) has read its send buffer heat_local, which stage 2 of batch i+1 overwrites on the caller's
        # stream next: an explicit dependency (the order of the collectives on the communicator implied it, but that
        # is an implementation detail of ProcessGroupNCCL)
        cur.wait_event(self._exchanged)
        for t in res:
            t.record_stream(cur)                 # allocated on the side stream, consumed on the caller's
        return res

    def _finish_on_current_stream(self):
        W = self.world
        self._pending.wait()                     # the current stream waits for exchange(i)
        self._pending = None
        if self.cuda:
            self._exchanged = torch.cuda.Event()
            self._exchanged.record()
        self._mark("exchange_done")
        if self.three_d == "rank0":
            if self.rank == 0:
                self._run_3d()
            # results of the whole time batch from rank 0 to everybody
            self.comm.broadcast(self.res_local, self._global_rank0(), group=self.group)
            self.comm.broadcast(self.valid_local, self._global_rank0(), group=self.group)
            self._mark("results_end")
            return self._split(self.res_local.view(1, -1), self.valid_local)
        self._run_3d()
        self._mark("stage3_end")
        self.comm.all_gather_into_tensor(self.res_all, self.res_local, group=self.group)
        self.comm.all_gather_into_tensor(self.valid_all, self.valid_local, group=self.group)
        self._mark("results_end")
        return self._split(self.res_all.view(W, -1), self.valid_all)

    def _split(self, res, valid):
        """(ranks, [points | conf]) -> points (T,J,3), conf (T,J), valid (T): fresh tensors (the
        result buffers are reused by the next time batch)."""
        n = res.shape[0] * self.T3
        cf = torch.contiguous_format
        return (res[:, :self.n_pts].clone(memory_format=cf).view(n, self.J, 3),
                res[:, self.n_pts:].clone(memory_format=cf).view(n, self.J), valid.clone())

    def _run_3d(self):
        # heat_recv is (W, frames per block, Cloc, h, w, Jp): camera c = source rank * Cloc + local
        # camera.  Read in place; with the all-gather only frames [t_lo, t_lo + T3) of every block.
        if self.exchange == "alltoall":
            fpb, t_off = self.T3, 0
        else:
            fpb, t_off = self.T, self.t_lo
        self.st.stage_3d_blocks(self.heat_recv, self.world, fpb, t_off, self.t_lo, self.pts_local,
                                self.conf_local, self.valid_local)

    def _global_rank0(self):
        """broadcast() takes the GLOBAL rank of the source: rank 0 of this group."""
        if self.group is None or self.comm is not dist:
            return 0
        return dist.get_global_rank(self.group, 0)
