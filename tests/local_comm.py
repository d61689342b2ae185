"""An in-process stand-in for the torch.distributed collectives ShardedPredictor uses, so that
N emulated ranks (one Python thread each) can run the REAL distributed.py code path on one GPU
(or on the CPU): the test then exercises the sharding, both exchange modes, the camera / frame
re-assembly and the pipelined submit/flush exactly as an N-GPU job would, with the data
movement done by copies instead of RCCL.

All ranks issue their GPU work on the same (default) stream, so device-side ordering equals
host issue order; the two barriers of every collective make sure (1) all inputs have been
produced (enqueued) before anybody copies and (2) all copies have been enqueued before any
rank goes on to overwrite its input.
"""
import threading


class _Done:
    def wait(self):
        return True


class LocalWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

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
        self.w.slots[self.rank] = inp
        self.w.barrier.wait()
        return list(self.w.slots)

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        ins = self._exchange(inp)
        chunks = out.view((self.w.world,) + tuple(inp.shape))
        for r, t in enumerate(ins):
            chunks[r].copy_(t)
        self.w.barrier.wait()
        return _Done()

    def all_to_all_single(self, out, inp, group=None, async_op=False):
        """out block r <- rank r's input block [my rank] (equal splits along dim 0)."""
        ins = self._exchange(inp)
        n = inp.shape[0] // self.w.world
        blocks = out.view((self.w.world, n) + tuple(inp.shape[1:]))
        for r, t in enumerate(ins):
            blocks[r].copy_(t.view((self.w.world, n) + tuple(inp.shape[1:]))[self.rank])
        self.w.barrier.wait()
        return _Done()

    def broadcast(self, tensor, src, group=None, async_op=False):
        ins = self._exchange(tensor)
        if self.rank != src:
            tensor.copy_(ins[src])
        self.w.barrier.wait()
        return _Done()
