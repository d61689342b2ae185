"""An in-process stand-in for the torch.distributed collectives ShardedPredictor uses, so that
N emulated ranks (one Python thread each) can run the REAL distributed.py code path on one GPU
(or on the CPU): the test then exercises the sharding, both exchange modes, the camera / frame
re-assembly and the pipelined submit/flush exactly as an N-GPU job would, with the data
movement done by copies instead of RCCL.

The two barriers of every collective make sure (1) all inputs have been produced (enqueued) before
anybody copies and (2) all copies have been enqueued before any rank goes on to overwrite its input.
Ranks may issue their work on different HIP streams (ShardedPredictor runs stage 3 and the result
collectives on a side stream): every rank therefore publishes an event recorded on ITS current stream
with its input, consumers make their stream wait for it before copying, and after the copies every rank
publishes a second event that the others' streams wait for before they go on -- the stream-level
equivalent of the two host barriers.
"""
import threading

import torch


class _Done:
    def wait(self):
        return True


class LocalWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.done = [None] * world

    def comm(self, rank):
        return LocalComm(self, rank)

    def run(self, fn):
        """fn(rank, comm) on `world` threads; returns the list of results, re-raises the first
        exception (and breaks the barrier so that no thread waits forever)."""
        res, err = [None] * self.world, []

        def body(r):
            try:
                res[r] = fn(r, self.comm(r))
            except BaseException as e:       # noqa: BLE001 -- reported to the caller below
                err.append(e)
                self.barrier.abort()
        ths = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            first = [e for e in err if not isinstance(e, threading.BrokenBarrierError)] or err
            raise first[0]
        return res


class LocalComm:
    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    def _exchange(self, inp):
        ev = None
        if inp.is_cuda:
            ev = torch.cuda.Event()
            ev.record()                              # (on this rank's current stream: its input is complete here)
        self.w.slots[self.rank] = (inp, ev)
        self.w.barrier.wait()
        ins = []
        for t, e in self.w.slots:
            if e is not None:
                torch.cuda.current_stream().wait_event(e)
            ins.append(t)
        return ins

    def _release(self, cuda):
        """All copies enqueued (host barrier) AND executed before any rank's stream goes on."""
        ev = None
        if cuda:
            ev = torch.cuda.Event()
            ev.record()
        self.w.done[self.rank] = ev
        self.w.barrier.wait()
        for e in list(self.w.done):
            if e is not None:
                torch.cuda.current_stream().wait_event(e)
        self.w.barrier.wait()                        # (nobody overwrites `done` before everybody has read it)

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        ins = self._exchange(inp)
        chunks = out.view((self.w.world,) + tuple(inp.shape))
        for r, t in enumerate(ins):
            chunks[r].copy_(t)
        self._release(inp.is_cuda)
        return _Done()

    def all_to_all_single(self, out, inp, group=None, async_op=False):
        """out block r <- rank r's input block [my rank] (equal splits along dim 0)."""
        ins = self._exchange(inp)
        n = inp.shape[0] // self.w.world
        blocks = out.view((self.w.world, n) + tuple(inp.shape[1:]))
        for r, t in enumerate(ins):
            blocks[r].copy_(t.view((self.w.world, n) + tuple(inp.shape[1:]))[self.rank])
        self._release(inp.is_cuda)
        return _Done()

    def broadcast(self, tensor, src, group=None, async_op=False):
        ins = self._exchange(tensor)
        if self.rank != src:
            tensor.copy_(ins[src])
        self._release(tensor.is_cuda)
        return _Done()
