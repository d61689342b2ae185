"""Python handle of the native predictor (jh_predictor_* in include/jarvis_hip.h)."""
import ctypes

import torch

from . import _native as N
from . import arch


class NativePredictor:
    """Owns the launch plans of CenterDetect, KeypointDetect and V2V plus all
    intermediates for `time_batch` multi-view frames of a fixed size."""

    def __init__(self, center_state, hybrid_state, *, num_cameras, num_joints, center_size, bbox,
                 roi_cube_size, grid_spacing, img_h, img_w, mean, std, center_model="small",
                 kp_model="small", time_batch=1, time_batch_3d=0, cam_lo=0, cam_n=None, precision=None):
        """precision: "f32" | "bf16x3" | "bf16x3_wide" for THIS predictor, or None = the process default
        (`_native.set_precision`, environment JH_PRECISION) at creation time."""
        cam_n = num_cameras if cam_n is None else cam_n
        cfg = N.PredictorConfig(
            num_cameras, num_joints, center_size, bbox, float(roi_cube_size), float(grid_spacing),
            arch.SIZE_IDS[center_model], arch.SIZE_IDS[kp_model], img_h, img_w, time_batch,
            time_batch_3d, cam_lo, cam_n, (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std),
            N.precision_id(precision))
        self.cfg = cfg
        self.T, self.C, self.Cloc, self.J = time_batch, num_cameras, cam_n, num_joints
        self.T3 = time_batch_3d if time_batch_3d > 0 else time_batch
        self.Jp = (num_joints + 7) // 8 * 8
        self.Hh = bbox // 2
        self.handle = ctypes.c_void_p()
        pc = N.Params(center_state) if center_state is not None else None
        ph = N.Params(hybrid_state)
        N.check(N.lib().jh_predictor_create(pc.handle if pc else None, ph.handle,
                                            ctypes.byref(cfg), ctypes.byref(self.handle)))
        self.precision = {v: k for k, v in N.PRECISIONS.items()}[N.lib().jh_predictor_precision(self.handle)]
        self.launches = N.lib().jh_predictor_launches(self.handle)
        self.device_bytes = N.lib().jh_predictor_device_bytes(self.handle)

    def close(self):
        if getattr(self, "handle", None) and N is not None and N._lib is not None:
            N.lib().jh_predictor_destroy(self.handle)
            self.handle = None

    __del__ = close

    # ---- graph replay of the whole forward (default: on for time_batch == 1) ---
    @property
    def graph_replay(self):
        return bool(N.lib().jh_predictor_graph_replay(self.handle))

    @graph_replay.setter
    def graph_replay(self, on):
        N.check(N.lib().jh_predictor_set_graph_replay(self.handle, int(bool(on))))

    # ---- calibration -----------------------------------------------------
    def set_calibration(self, cam, intr, dist):
        N.check(N.lib().jh_predictor_set_calibration(
            self.handle, N.ptr(N.dev(cam)), N.ptr(N.dev(intr)), N.ptr(N.dev(dist)), N.stream()))

    # ---- single-GPU forward ----------------------------------------------
    def forward(self, frames, out=None):
        """frames (T,C,3,H,W) fp32 RGB, or (T,C,H,W,3) uint8 BGR as decoded
        -> points (T,J,3), conf (T,J), valid (T) int32."""
        self._check_frames(frames, self.Cloc)
        dev = frames.device
        if out is None:
            out = (torch.empty((self.T, self.J, 3), device=dev),
                   torch.empty((self.T, self.J), device=dev),
                   torch.empty((self.T,), device=dev, dtype=torch.int32))
        fn = N.lib().jh_predictor_forward_u8 if frames.dtype == torch.uint8 else \
            N.lib().jh_predictor_forward
        N.check(fn(self.handle, N.ptr(frames), N.ptr(out[0]), N.ptr(out[1]), N.ptr(out[2]),
                   N.stream()))
        return out

    def _check_frames(self, frames, cams):
        """Raw pointers cross the C ABI: refuse anything whose bytes would be misread."""
        H, W = self.cfg.img_h, self.cfg.img_w
        if not (torch.is_tensor(frames) and frames.is_cuda and frames.is_contiguous()):
            raise RuntimeError("frames must be a contiguous CUDA (HIP) tensor")
        if frames.dtype == torch.uint8:
            want = (self.T, cams, H, W, 3)
        elif frames.dtype == torch.float32:
            want = (self.T, cams, 3, H, W)
        else:
            raise RuntimeError("frames must be float32 RGB (T,C,3,H,W) or uint8 BGR (T,C,H,W,3); "
                               "got dtype %s" % frames.dtype)
        if tuple(frames.shape) != want:
            raise RuntimeError("frames shape %s, expected %s" % (tuple(frames.shape), want))

    # ---- camera-sharded stages -------------------------------------------
    def stage_center(self, frames, det):
        self._check_frames(frames, self.Cloc)
        fn = N.lib().jh_predictor_stage_center_u8 if frames.dtype == torch.uint8 else \
            N.lib().jh_predictor_stage_center
        N.check(fn(self.handle, N.ptr(frames), N.ptr(det), N.stream()))

    def stage_keypoints(self, frames, det_all, heat):
        self._check_frames(frames, self.Cloc)
        fn = N.lib().jh_predictor_stage_keypoints_u8 if frames.dtype == torch.uint8 else \
            N.lib().jh_predictor_stage_keypoints
        N.check(fn(self.handle, N.ptr(frames), N.ptr(det_all), N.ptr(heat), N.stream()))

    def stage_3d(self, heat_all, t0, points, conf, valid):
        N.check(N.lib().jh_predictor_stage_3d(self.handle, N.ptr(heat_all), t0, N.ptr(points),
                                              N.ptr(conf), N.ptr(valid), N.stream()))

    def stage_keypoints_gathered(self, frames, det_gathered, n_blocks, heat):
        """Stage 2 reading the all-gathered detections (n_blocks, T, C/n_blocks, 3) in place."""
        self._check_frames(frames, self.Cloc)
        N.check(N.lib().jh_predictor_stage_keypoints_gathered(
            self.handle, N.ptr(frames), int(frames.dtype == torch.uint8), N.ptr(det_gathered), n_blocks,
            N.ptr(heat), N.stream()))

    def stage_3d_blocks(self, heat_blocks, n_blocks, frames_per_block, t_off, t0, points, conf, valid):
        """3D stage reading the exchange's receive buffer (n_blocks, frames_per_block, C/n_blocks,
        h, w, Jp) in place: frames t_off .. t_off+T3-1 of every block."""
        N.check(N.lib().jh_predictor_stage_3d_blocks(
            self.handle, N.ptr(heat_blocks), n_blocks, frames_per_block, t_off, t0, N.ptr(points),
            N.ptr(conf), N.ptr(valid), N.stream()))

    def debug(self, device):
        c3f = torch.empty((self.T, 3), device=device)
        c3i = torch.empty((self.T, 3), device=device, dtype=torch.int32)
        chm = torch.empty((self.T, self.C, 2), device=device, dtype=torch.int32)
        det = torch.empty((self.T, self.C, 3), device=device)
        N.check(N.lib().jh_predictor_debug(self.handle, N.ptr(c3f), N.ptr(c3i), N.ptr(chm),
                                           N.ptr(det), N.stream()))
        return dict(center3d=c3f, center3d_int=c3i, center_hm=chm, det=det)

    def hybridnet_forward(self, crops, center_hm, center3d, want_final=True, want_padded=True):
        dev = crops.device
        Gh = int(self.cfg.roi_cube_size / self.cfg.grid_spacing) // 2
        hs = self.Hh + 2
        final = torch.empty((self.T, self.J, Gh, Gh, Gh), device=dev) if want_final else None
        padded = torch.empty((self.T, self.C, self.J, hs, hs), device=dev) if want_padded else None
        pts = torch.empty((self.T, self.J, 3), device=dev)
        conf = torch.empty((self.T, self.J), device=dev)
        N.check(N.lib().jh_predictor_hybridnet_forward(
            self.handle, N.ptr(crops), N.ptr(center_hm), N.ptr(center3d), N.ptr(final),
            N.ptr(padded), N.ptr(pts), N.ptr(conf), N.stream()))
        return final, padded, pts, conf


class MultiStreamPredictor:
    """K independent time batches in flight on K HIP streams.

    Every stream has its own NativePredictor (own launch plans, activations and scratch), so
    consecutive `forward` calls have no hazards between them and the GPU interleaves their
    kernels: the phases in which one batch leaves resources idle (the Winograd V2V kernel runs
    one workgroup per CU and keeps the matrix cores busy about half the time; the 2D layers
    are mostly latency / bandwidth bound) are filled by another batch.  Measured on one
    MI355X at BASELINE configs[2]: 1602 frames/s with 3 streams x 32 frames against 1528 for
    one stream x 64 frames.

    forward() returns the output tensors of the batch it has just ENQUEUED on its stream; call
    synchronize() (or wait on `last_event`) before reading them.
    """

    def __init__(self, make_predictor, streams=3, timing=False):
        self.timing = timing                         # events usable with elapsed_time (bench.py)
        self.preds = [make_predictor() for _ in range(streams)]
        self.streams = [torch.cuda.Stream() for _ in range(streams)]
        self.events = [None] * streams
        self._next = 0
        self._calib = None

    def set_calibration(self, *calib):
        """Calibration of every predictor, written on that predictor's OWN stream (so the copy
        is ordered against the forwards in flight there).  Setting the same tensors again is a
        no-op: drivers may call this once per group of frames."""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in calib)
        if key == self._calib:
            return
        # the key is only meaningful while the keyed tensors are alive: a freed calibration's
        # address is handed out again by the caching allocator (same size, _version 0), and the
        # next recording's tensors would then look like "the same tensors again".  Holding them
        # makes address + version identify the contents.
        self._calib_refs = tuple(calib)
        cur = torch.cuda.current_stream()
        for p, s in zip(self.preds, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                p.set_calibration(*calib)
            for t in calib:
                if t.is_cuda:
                    t.record_stream(s)
        self._calib = key

    def forward(self, frames, out=None, then=None):
        """`then(outputs)`, when given, runs inside the batch's stream context right behind the forward and
        before its event is recorded (the drivers enqueue the device->host copy of the results there); its
        return value replaces the outputs."""
        i = self._next
        self._next = (i + 1) % len(self.preds)
        s = self.streams[i]
        s.wait_stream(torch.cuda.current_stream())       # `frames` may still be in the making
        # The kernels see raw pointers only, so the caching allocator must be told that the
        # side stream uses these blocks: without record_stream a caller that drops `frames`
        # right after this call could get the same memory back for its next batch while the
        # forward in flight here still reads it (it reads the frames twice: resize, then crops).
        frames.record_stream(s)
        for t in (out or ()):
            t.record_stream(s)
        with torch.cuda.stream(s):
            res = self.preds[i].forward(frames, out)
            if then is not None:
                res = then(res)
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record(s)
        self.events[i] = ev
        self.last_event = ev
        return res

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
