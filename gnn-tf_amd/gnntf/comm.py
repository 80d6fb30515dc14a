"""The ranks of a vertex partition and how they talk: ``Comm`` (a torch.distributed group or a single process; pairwise exchanges
as ONE batch of point-to-point operations -- on RCCL a single group call, every peer pair on its own xGMI link, no ring -- and the
probe that places the exchange beside the compute stream), and the pv x pf process grid.  The reference has no distributed code
(SURVEY.md section 2.1)."""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist


class Comm:
    """The ranks that share one vertex partition (a torch.distributed group, or a single process).
    A one-rank Comm never touches torch.distributed, so a feature-sliced grid with one vertex block
    per slice has no data-path communication at all."""

    def __init__(self, group=None, solo=False):
        self.group, self.solo = group, solo
        self.rank = 0 if solo else dist.get_rank(group)
        self.size = 1 if solo else dist.get_world_size(group)

    def _staged(self, t):
        # gloo cannot move device tensors point-to-point: such groups (tests, rehearsals of several
        # ranks on one card) stage through host memory.  RCCL groups never take this path.
        return t is not None and t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _global(self, q):
        return q if self.group is None else dist.get_global_rank(self.group, q)

    def all_reduce(self, t, op=None):
        if self.size == 1:
            return t
        op = dist.ReduceOp.SUM if op is None else op
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def broadcast(self, t, src=0):
        """In-place broadcast from group rank ``src``."""
        if self.size == 1:
            return t
        if self._staged(t):
            h = t.cpu()
            dist.broadcast(h, self._global(src), group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, self._global(src), group=self.group)
        return t

    def all_gather_vec(self, t):
        if self.size == 1:
            return [t]
        src = t.cpu() if self._staged(t) else t
        table = [torch.zeros_like(src) for _ in range(self.size)]
        dist.all_gather(table, src, group=self.group)
        return [x.to(t.device) for x in table]

    def exchange(self, send_chunks, recv_chunks):
        """Pairwise exchange: send_chunks[q] goes to group rank q, recv_chunks[q] is filled from it (None / empty: nothing)."""
        self.exchange_pairs([(q, t) for q, t in enumerate(send_chunks) if q != self.rank],
                            [(q, t) for q, t in enumerate(recv_chunks) if q != self.rank])

    def exchange_pairs(self, sends, recvs):
        """``sends`` / ``recvs``: lists of (group rank, tensor); several messages per peer are matched in list order.
        One batch of point-to-point operations (NCCL/RCCL: a single group call, every peer pair on its own xGMI link).
        Stream-ordered on the current stream for RCCL groups."""
        if self.size == 1:
            return
        sends = [(q, t) for q, t in sends if t is not None and t.numel() > 0]
        recvs = [(q, t) for q, t in recvs if t is not None and t.numel() > 0]
        if any(self._staged(t) for _, t in sends + recvs):
            host_recv = [(q, torch.empty(t.shape, dtype=t.dtype)) for q, t in recvs]
            self.exchange_pairs([(q, t.cpu()) for q, t in sends], host_recv)
            for (_, d), (_, h) in zip(recvs, host_recv):
                d.copy_(h)
            return
        ops = [dist.P2POp(dist.irecv, t, self._global(q), self.group) for q, t in recvs]
        ops += [dist.P2POp(dist.isend, t, self._global(q), self.group) for q, t in sends]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def barrier(self):
        if self.size > 1:
            dist.barrier(group=self.group)

    # ---- which streams carry the exchanges ----------------------------------------------------------------------------------
    def tune_overlap(self, device, force=False, lanes=4, groups=4, seconds=20.0):
        """Places the exchange so that transfers run BESIDE the compute stream's kernels.  HIP multiplexes streams onto a few
        hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise), and two streams matter here: the exchange lane, which
        holds the event waits around every RCCL group call, and the stream torch gives the process group for RCCL's own kernels.
        When either shares the compute stream's hardware queue its packets queue up with the compute kernels and a transfer costs
        its whole duration instead of hiding under the other column chunk's SpMM -- measured on MI355X: about one stream in four,
        for both (tools/overlap_probe3.py, profiles/NOTES.md).  Probe: a send / recv of this rank to ITSELF, issued from each of
        a few fresh lane streams, beside a few matrix products of the compute stream; when no lane hides the transfer on some
        rank the group's own stream is the one in the way, and the probe repeats on a fresh process group (up to ``groups``).
        The lane is each rank's own choice (``self.lane_stream``); the group is agreed on (``self.group``).

        Collective, and every decision in it is: a step that fails on ONE rank (an allocation, a new group, a transfer) is agreed
        on by all (all_reduce MIN of an ok flag) before anyone moves on, so no rank is ever left in a collective the others have
        given up -- on failure every rank keeps what was chosen so far (at worst the defaults) together.  ``seconds``: wall-clock
        cap (the slowest rank's clock), checked between groups.  Groups created and not chosen are destroyed.  A no-op for one
        rank, for sub-groups of a process grid, for backends other than "nccl", and when GNX_TUNE_OVERLAP=0.  Returns the probe
        table (kept in ``self.overlap_probe``; ``self.overlap_status`` says what happened, also when the probe bailed out)."""
        if getattr(self, "overlap_probe", None) is not None:
            return self.overlap_probe
        self.overlap_probe, self.lane_stream = [], None
        status = self.overlap_status = {"ran": False, "reason": None, "seconds": 0.0, "groups_tried": 0, "groups_destroyed": 0}
        if os.environ.get("GNX_TUNE_OVERLAP", "1") == "0":
            status["reason"] = "disabled (GNX_TUNE_OVERLAP=0)"
            return self.overlap_probe
        if device.type != "cuda" or self.solo or self.group is not None or not dist.is_initialized() or dist.get_backend() != "nccl":
            status["reason"] = "not applicable (needs the default process group on the nccl backend)"
            return self.overlap_probe
        if self.size == 1 and not force:
            status["reason"] = "one rank"
            return self.overlap_probe
        me = dist.get_rank()
        t_start = time.perf_counter()

        def agree(ok):
            """True when EVERY rank says ok (collective on the default group)."""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def slowest(x):
            t = torch.tensor([x], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # probe buffers first, sized from what is free (the block's entries are on the device already): at most 1 GiB per message,
        # at most 1/16 of the free memory for each of the two
        problem = ""
        try:
            free, _ = torch.cuda.mem_get_info(device)
            floats = int(min(256 << 20, max(free // 64, 1 << 20)))
            A = torch.randn(4096, 4096, device=device)
            B = torch.randn(4096, 4096, device=device)
            C = torch.empty_like(A)
            src = torch.empty(floats, dtype=torch.float32, device=device).normal_()    # about a millisecond of RCCL kernel at 1 GiB
            dst = torch.empty_like(src)
        except Exception as exc:
            problem = repr(exc)[:200]
        if not agree(not problem):
            status["reason"] = "probe buffers could not be allocated on some rank" + (": " + problem if problem else "")
            return self.overlap_probe
        status["message_bytes"] = floats * 4
        HIDES = 0.75                                                       # measured: 0.5-0.6 beside the products when placed well, 0.85-1.1 when not

        def products():
            for _ in range(8):
                torch.mm(A, B, out=C)

        def transfer(lane, group):
            with torch.cuda.stream(lane):
                for _ in range(2):
                    for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, me, group), dist.P2POp(dist.isend, src, me, group)]):
                        req.wait()

        def ms(fn):
            best = None
            for _ in range(3):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - t0) * 1e3
                best = dt if best is None else min(best, dt)
            return best

        streams = [torch.cuda.Stream(device) for _ in range(max(1, lanes))]
        self._probe_lanes = streams                                        # kept alive: a freed stream's queue slot would be handed out again
        chosen = None                                                      # (worst rank's share, group, this rank's lane, labels)
        made = []                                                          # groups this probe created
        for g in range(max(1, groups)):
            group, problem = None, ""
            if g > 0:
                try:
                    group = dist.new_group(list(range(dist.get_world_size())), backend="nccl")
                    made.append(group)
                except Exception as exc:
                    problem = repr(exc)[:200]
            if not agree(not problem):
                status["reason"] = f"group {g} could not be created on some rank" + (": " + problem if problem else "")
                break
            status["groups_tried"] = g + 1
            mine = None
            try:
                transfer(streams[0], group)                                # (first call of a group: communicator and stream set-up, untimed)
                t_c = ms(products)
                for k, lane in enumerate(streams):
                    t_x = ms(lambda: transfer(lane, group))
                    t_both = ms(lambda: (transfer(lane, group), products()))
                    exposed = max(t_both - t_c, 0.0) / max(t_x, 1e-6)      # share of the transfer that did NOT hide
                    self.overlap_probe.append(dict(group=g, lane=k, products_ms=t_c, transfer_ms=t_x, together_ms=t_both, exposed_share=exposed))
                    if mine is None or exposed < mine[0] - 0.1:            # (ties: the first)
                        mine = (exposed, lane, k)
                    if exposed < HIDES:
                        break
            except Exception as exc:                                       # (a transfer to oneself involves no other rank: the others finish theirs)
                problem = repr(exc)[:200]
            if not agree(not problem and mine is not None):
                status["reason"] = f"the probe of group {g} failed on some rank" + (": " + problem if problem else "")
                break
            worst = slowest(mine[0])                                       # the step takes the slowest rank's time
            if chosen is None or worst < chosen[0] - 0.1:
                chosen = (worst, group, mine[1], (g, mine[2]))
            if worst < HIDES:
                break
            if slowest(time.perf_counter() - t_start) > seconds:
                status["reason"] = f"time cap of {seconds:.0f} s reached after group {g}"
                break
        if chosen is not None:
            self.group, self.lane_stream = chosen[1], chosen[2]
            for rec in self.overlap_probe:
                rec["chosen"] = (rec["group"], rec["lane"]) == chosen[3]
            self.overlap_exposed_worst_rank = chosen[0]
            status.update(ran=True, chosen_group=chosen[3][0], chosen_lane=chosen[3][1], exposed_share_worst_rank=chosen[0])
        for group in made:                                                 # (same list, same order on every rank)
            if chosen is None or group is not chosen[1]:
                try:
                    dist.destroy_process_group(group)
                    status["groups_destroyed"] += 1
                except Exception:
                    pass
        del A, B, C, src, dst
        status["seconds"] = round(time.perf_counter() - t_start, 2)
        return self.overlap_probe

    def alltoallv(self, chunks):
        """chunks[q]: a 1-D tensor for group rank q (any length).  Returns what every rank sent to this one."""
        if self.size == 1:
            return [chunks[0]]
        ref = chunks[0]
        counts = torch.tensor([int(c.numel()) for c in chunks], dtype=torch.int64, device=ref.device)
        table = self.all_gather_vec(counts)
        recv = [torch.empty(int(table[q][self.rank]), dtype=ref.dtype, device=ref.device) for q in range(self.size)]
        self.exchange([c.contiguous() for c in chunks], recv)
        recv[self.rank] = chunks[self.rank]
        return recv


def make_grid(world, rank, pv, pf):
    """Process grid of pv vertex blocks x pf feature slices (pv * pf == world); rank = v * pf + f.
    Returns (v, f, Comm of the pv ranks that share feature slice f).  Collective when pv > 1 and pf > 1
    (every rank creates every sub-group, in the same order)."""
    if pv * pf != world:
        raise Exception("make_grid: pv * pf must equal the world size")
    v, f = rank // pf, rank % pf
    if pv == 1:
        return v, f, Comm(solo=True)
    if pf == 1:
        return v, f, Comm(group=None)
    mine = None
    for ff in range(pf):
        grp = dist.new_group([vv * pf + ff for vv in range(pv)])
        if ff == f:
            mine = grp
    return v, f, Comm(group=mine)


def grid_cost_ms(pv, pf, feats, nodes_total, entries_total, link_GBs, halo_frac=0.09):
    """Estimated time of ONE propagation iteration on a pv x pf grid:
      * compute: entries per rank x (6 + 0.53 * max(w, 32)) ps, w = columns per rank -- the fused kernel's
        measured cost (RMAT 10M/100M, round 2: 2.4 / 3.9 / 7.3 / 14.5 ms at w = 32 / 64 / 128 / 256; below 32 columns a
        gather still moves one 128-byte line, so narrower slices are not cheaper);
      * exchange (pv > 1 only): halo_frac * nodes_total rows of 4w bytes arrive per rank over its pv - 1 links
        (one xGMI link per peer, ``link_GBs`` per direction -- a MEASURED figure: bench.py times a pairwise
        exchange in-run and reports it; there is no built-in default).  halo_frac: cover rows per rank as a
        share of all vertices (graph dependent; ShardedGraph.halo_stats() reports the real one);
      * the two overlap (column chunks), so the iteration costs the larger of them."""
    w = max(feats // pf, 1)
    compute = entries_total / pv * (6.0 + 0.53 * max(w, 32)) * 1e-9
    comm = 0.0 if pv == 1 else halo_frac * nodes_total * 4.0 * w / ((pv - 1) * link_GBs * 1e9) * 1e3
    return max(compute, comm)


def choose_grid(world, feats, nodes_total, entries_total, link_GBs, **model):
    """The pv x pf factorisation of ``world`` (pf dividing ``feats``) with the lowest grid_cost_ms for a measured
    link rate.  bench.py does NOT use it for its headline number (that is always pv = world vertex blocks, the
    grid BASELINE.json names); it is a planning helper: feature slices replicate the whole graph on every rank
    (memory and prep grow with pf) and pay nothing per iteration, vertex blocks are the opposite."""
    best = None
    for pf in range(1, world + 1):
        if world % pf or feats % pf:
            continue
        pv = world // pf
        cost = grid_cost_ms(pv, pf, feats, nodes_total, entries_total, link_GBs, **model)
        if best is None or cost < best[0] - 1e-12:
            best = (cost, pv, pf)
    return best[1], best[2]
