"""An in-process stand-in for gnntf.sharded.Comm: the P ranks of a vertex partition run as THREADS of one process and
exchange through shared memory.  Test infrastructure: lets one GPU hold all P blocks of a graph at a realistic size and
compare the vertex-block path (real plan, real kernels, exact exchange semantics) with the one-GPU path, which the gloo
tests (two processes sharing a card, staged through the host) can only do at toy sizes."""
import threading

import torch


class ThreadWorld:
    def __init__(self, size):
        self.size = size
        self.gate = threading.Barrier(size)
        self.slots = [None] * size


class ThreadComm:
    group = None
    solo = False

    def __init__(self, world: ThreadWorld, rank: int):
        self.world, self.rank, self.size = world, rank, world.size

    def _sync(self, t=None):
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _gather(self, obj):
        """Every rank deposits ``obj``; returns the list of all deposits (valid until the next collective)."""
        self._sync()
        self.world.slots[self.rank] = obj
        self.world.gate.wait()
        out = list(self.world.slots)
        return out

    def _done(self):
        self._sync()
        self.world.gate.wait()

    def all_reduce(self, t, op=None):
        parts = self._gather(t.clone())
        import torch.distributed as dist
        stacked = torch.stack(parts)
        res = stacked.max(0).values if op == dist.ReduceOp.MAX else stacked.sum(0)
        self._done()
        t.copy_(res)
        return t

    def broadcast(self, t, src=0):
        parts = self._gather(t)
        if self.rank != src:
            t.copy_(parts[src])
        self._done()
        return t

    def all_gather_vec(self, t):
        parts = [x.clone() for x in self._gather(t)]
        self._done()
        return parts

    def exchange(self, send_chunks, recv_chunks):
        self.exchange_pairs([(q, t) for q, t in enumerate(send_chunks) if q != self.rank],
                            [(q, t) for q, t in enumerate(recv_chunks) if q != self.rank])

    def exchange_pairs(self, sends, recvs):
        """Several messages per peer are matched in list order, as a group of point-to-point transfers would."""
        posted = self._gather([(q, t) for q, t in sends if t is not None and t.numel() > 0])
        taken = [0] * self.size
        for q, dst in recvs:
            if dst is None or dst.numel() == 0:
                continue
            mine = [t for to, t in posted[q] if to == self.rank]
            dst.copy_(mine[taken[q]])
            taken[q] += 1
        self._done()

    def barrier(self):
        self.world.gate.wait()

    def alltoallv(self, chunks):
        sends = self._gather(list(chunks))
        out = [sends[q][self.rank].clone() for q in range(self.size)]
        self._done()
        return out


def run_ranks(size, fn):
    """fn(comm) on ``size`` threads; returns the list of results (re-raises the first failure)."""
    world = ThreadWorld(size)
    results, errors = [None] * size, []

    def body(r):
        try:
            results[r] = fn(ThreadComm(world, r))
        except BaseException as e:                 # noqa: BLE001 -- surface it in the main thread
            errors.append(e)
            world.gate.abort()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(size)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
