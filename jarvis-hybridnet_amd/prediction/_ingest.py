"""Overlapped frame ingest of the predict2D / predict3D drivers.

The reference driver reads the next frame set with 12 threads into one preallocated array and
uploads it per frame (jarvis/prediction/predict3D.py:72-80: joblib `read_images` into `imgs_orig`,
then `torch.from_numpy(imgs_orig).cuda()`).  Here the same three steps run as a pipeline over time
batches of `time_batch` frame sets:

  host fill   decoded frames are copied (or decoded in place, see `push`) into a PINNED staging
              buffer of one time batch by a small thread pool, chunk by chunk, as the iterator
              yields them -- no `torch.stack`, no pageable intermediate;
  upload      one asynchronous host->HBM copy per time batch on a dedicated copy stream;
  compute     the predictor's forward on its own HIP stream(s) (`MultiStreamPredictor`), ordered
              behind the upload by an event;
  rows        results are read back in frame order once their batch's event has completed.

`slots` staging / device buffer pairs (streams + 2) rotate, so the fill of batch i+1 and
the upload of batch i overlap the forward of batch i-1.  Without a GPU (CPU tests of the host logic
with stub predictors) the same pipeline runs with plain host buffers and no streams.
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_CHUNK = 12 << 20                   # bytes per copy job
_PINNED_BUDGET = 12 << 30           # staging memory above which the pipeline keeps only two slots


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _host_array(frames):
    """A decoded frame set as a numpy view (no copy for numpy arrays and CPU tensors)."""
    if torch.is_tensor(frames):
        return frames.detach().numpy() if not frames.is_cuda else None
    return np.asarray(frames)


class FramePipeline:
    """Time batches of frame sets in, `emit(outputs_on_host, n_real)` calls in frame order out.

    submit(x, slot) -> (outputs, event): enqueue the forward of one uploaded time batch `x`
    (time_batch, *frame_shape) and return its output tensors plus the event that marks their
    completion (None on the CPU: outputs are final when submit returns).  Outputs should already be
    HOST tensors when the event fires (an asynchronous device->host copy into pinned memory enqueued
    behind the forward on ITS stream, see `host_outputs`; `slot` < `slots` names the buffer set to use):
    a synchronous `.cpu()` from the calling thread queues behind the multi-gigabyte upload in flight
    (measured: 25 ms per batch at configs[2])."""

    def __init__(self, frame_shape, dtype, time_batch, streams, submit, emit, device, copy_threads=None):
        self.T, self.submit, self.emit = int(time_batch), submit, emit
        self.frame_shape, self.dtype = tuple(frame_shape), dtype
        self.cuda = device is not None and torch.device(device).type == "cuda"
        shape = (self.T,) + self.frame_shape
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        # streams + 2 buffer pairs: while batch g is filled, batches g-1 .. g-streams are uploaded / computed and
        # the rows of batch g-streams-1 are written
        self.slots = int(streams) + 2
        if nbytes * self.slots > _PINNED_BUDGET:
            self.slots = 2
        self.host = [torch.empty(shape, dtype=dtype, pin_memory=self.cuda) for _ in range(self.slots)]
        self.host_np = [h.numpy() for h in self.host]
        if self.cuda:
            self.dev = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.slots)]
            self.copy_stream = torch.cuda.Stream(device=device)
            self.uploaded = [torch.cuda.Event() for _ in range(self.slots)]
            self.consumed = [None] * self.slots
        threads = copy_threads or max(2, min(8, usable_cores() - 2))
        self.pool = ThreadPoolExecutor(max_workers=threads)
        self.frame_bytes = nbytes // self.T
        self.rebind(submit, emit)

    def rebind(self, submit, emit):
        """Start a new run on the same staging buffers (a driver call re-uses the pinned memory of the
        previous one: pinning gigabytes costs seconds)."""
        self.quiesce()
        self.submit, self.emit = submit, emit
        self.group, self.fill, self.jobs = 0, 0, []
        self.inflight = []                      # (outputs, event, n_real) of submitted batches
        self.frames_in = 0
        # where the calling thread spends its time (seconds): waiting for the pool's copies, enqueueing upload +
        # forward, waiting for the oldest batch's event, reading back + emitting rows
        self.stats = dict(fill_wait=0.0, submit=0.0, event_wait=0.0, emit=0.0, batches=0)
        return self

    def quiesce(self):
        """Nothing of an earlier run may still touch the staging buffers when a new run starts filling slot 0: a
        run that ended in an exception (frame iterator, fill callable, submit) leaves pool jobs and enqueued
        uploads / forwards behind.  Wait for the host-side jobs (their exceptions belong to the old run) and for the
        events of the batches in flight; the copy stream's uploads precede those forwards."""
        for j in getattr(self, "jobs", ()):
            try:
                j.result()
            except Exception:                    # noqa: BLE001 -- reported by the run that raised it
                pass
        for _, ev, _ in getattr(self, "inflight", ()):
            if ev is not None:
                ev.synchronize()
        if self.cuda and getattr(self, "copy_stream", None) is not None:
            self.copy_stream.synchronize()

    # ---- host fill ---------------------------------------------------------------------------------
    def _copy_into(self, dst, src):
        """dst (numpy view of the staging buffer) <- src, split over the pool's threads; numpy
        releases the GIL for these copies."""
        if src.flags.c_contiguous and dst.nbytes > _CHUNK:
            d, s = dst.reshape(-1), src.reshape(-1)
            step = max(1, _CHUNK // max(1, d.itemsize))
            for o in range(0, d.size, step):
                self.jobs.append(self.pool.submit(np.copyto, d[o:o + step], s[o:o + step]))
        else:
            self.jobs.append(self.pool.submit(np.copyto, dst, src))

    def push(self, frames):
        """One decoded frame set: a numpy array / CPU tensor of `frame_shape` (copied into the staging
        buffer by the pool), or a callable `fill(dst)` that decodes straight into the numpy view `dst` of
        the pinned buffer (the reference's `read_images(cap, slice, imgs_orig)` pattern: no copy at all;
        it runs on a pool thread)."""
        slot = self.group % self.slots
        dst = self.host_np[slot][self.fill]
        if callable(frames):
            self.jobs.append(self.pool.submit(frames, dst))
        else:
            self._copy_into(dst, frames)
        self.fill += 1
        self.frames_in += 1
        if self.fill == self.T:
            self._launch()

    def _launch(self):
        slot, real = self.group % self.slots, self.fill
        if real < self.T:                       # short last batch: padded with its last frame set
            h = self.host_np[slot]
            for j in self.jobs:
                j.result()
            self.jobs = []
            for t in range(real, self.T):
                self._copy_into(h[t], h[real - 1])
        # rows of the oldest batch while the pool is still copying (its slot is the one filled after this one)
        self.drain(self.slots - 2)
        t0 = time.perf_counter()
        for j in self.jobs:
            j.result()
        self.jobs = []
        t1 = time.perf_counter()
        if self.cuda:
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(self.copy_stream):
                if self.consumed[slot] is not None:          # the forward that last read this device buffer
                    self.copy_stream.wait_event(self.consumed[slot])
                self.dev[slot].copy_(self.host[slot], non_blocking=True)
                self.uploaded[slot].record(self.copy_stream)
            cur.wait_event(self.uploaded[slot])
            outs, ev = self.submit(self.dev[slot], slot)
            self.consumed[slot] = ev
        else:
            outs, ev = self.submit(self.host[slot], slot)
        t2 = time.perf_counter()
        self.stats["fill_wait"] += t1 - t0
        self.stats["submit"] += t2 - t1
        self.stats["batches"] += 1
        self.inflight.append((outs, ev, real))
        self.group += 1
        self.fill = 0
        # the slot filled next belonged to batch group - slots: at most slots - 1 batches are in flight here (the
        # drain above ran before this batch was appended), so that batch's forward (and with it the upload before
        # it) has completed before a pool thread overwrites the host buffer
        assert len(self.inflight) <= self.slots - 1

    # ---- rows ----------------------------------------------------------------------------------------
    def drain(self, keep=0):
        while len(self.inflight) > keep:
            outs, ev, real = self.inflight.pop(0)
            t0 = time.perf_counter()
            if ev is not None:
                ev.synchronize()
            t1 = time.perf_counter()
            self.emit(tuple(o if not o.is_cuda else o.cpu() for o in outs), real)
            self.stats["event_wait"] += t1 - t0
            self.stats["emit"] += time.perf_counter() - t1

    def finish(self):
        """Flush a partial batch and every batch in flight; returns the number of frame sets taken."""
        if self.fill:
            self._launch()
        self.drain(0)
        return self.frames_in

    def close(self):
        self.pool.shutdown(wait=True)


class DevicePipeline:
    """The same interface for frame sets that already live in HBM (CUDA tensors): no staging, the
    time batch is assembled on the device."""

    def __init__(self, time_batch, submit, emit, streams=1):
        self.T, self.submit, self.emit = int(time_batch), submit, emit
        self.group, self.frames_in = [], 0
        self.keep = max(0, int(streams) - 1)    # batches left in flight behind the one just submitted
        self.slots = self.keep + 1              # output-ring slots: one per batch that can be in flight
        self.n, self.inflight = 0, []

    def push(self, frames):
        self.group.append(frames)
        self.frames_in += 1
        if len(self.group) == self.T:
            self._launch()

    def _launch(self):
        real = len(self.group)
        x = torch.stack(self.group + [self.group[-1]] * (self.T - real))
        self.group = []
        # the slot's previous batch has been emitted: at most `keep` batches are in flight at this point
        self.drain(self.keep)
        outs, ev = self.submit(x, self.n % self.slots)
        self.n += 1
        self.inflight.append((outs, ev, real))

    def drain(self, keep=0):
        while len(self.inflight) > keep:
            outs, ev, real = self.inflight.pop(0)
            if ev is not None:
                ev.synchronize()
            self.emit(tuple(o if not o.is_cuda else o.cpu() for o in outs), real)

    def finish(self):
        if self.group:
            self._launch()
        self.drain(0)
        return self.frames_in


def host_outputs(ring, slot, outs):
    """Enqueue, on the CURRENT stream, asynchronous copies of the device tensors `outs` into pinned host tensors
    kept per `slot` in the dict `ring`; returns the host tensors (valid once an event recorded behind this call
    has completed)."""
    if not outs[0].is_cuda:
        return outs
    host = ring.get(slot)
    if host is None or any(h.shape != o.shape or h.dtype != o.dtype for h, o in zip(host, outs)):
        host = ring[slot] = tuple(torch.empty(o.shape, dtype=o.dtype, pin_memory=True) for o in outs)
    for h, o in zip(host, outs):
        h.copy_(o, non_blocking=True)
    return host


def pipeline_for(owner, frames, time_batch, streams, submit, emit, frame_spec=None):
    """The pipeline for frame sets shaped like `frames` (or like `frame_spec` = (shape, torch dtype) when
    `frames` is a fill callable), cached on `owner` (the predictor) so that its pinned buffers are re-used by
    later driver calls."""
    if torch.is_tensor(frames) and frames.is_cuda:
        return DevicePipeline(time_batch, submit, emit, streams)
    if callable(frames):
        if frame_spec is None:
            raise ValueError("fill callables need frame_spec=(shape, dtype)")
        shape, dtype = tuple(frame_spec[0]), frame_spec[1]
    else:
        a = _host_array(frames)
        shape, dtype = tuple(a.shape), torch.from_numpy(np.empty(0, a.dtype)).dtype
    device = "cuda" if torch.cuda.is_available() else None
    cache = owner.__dict__.setdefault("_ingest_cache", {})
    key = (shape, dtype, int(time_batch), int(streams), device)
    pipe = cache.get(key)
    if pipe is None:
        # ONE cached pipeline per predictor: another frame format / time batch / stream count replaces it (an entry
        # holds streams + 2 pinned host buffers and as many HBM buffers of a whole time batch -- 7.5 GB + 7.5 GB at
        # configs[2], T = 32, 3 streams -- plus its thread pool; release_ingest_buffers() frees the last one)
        for old in cache.values():
            old.quiesce()
            old.close()
        cache.clear()
        pipe = cache[key] = FramePipeline(shape, dtype, time_batch, streams, submit, emit, device)
        return pipe
    return pipe.rebind(submit, emit)


def release_ingest_buffers(owner):
    """Free the pinned staging / device buffers cached on a predictor by the drivers."""
    for pipe in owner.__dict__.pop("_ingest_cache", {}).values():
        pipe.quiesce()
        pipe.close()
