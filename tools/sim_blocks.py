#!/usr/bin/env python3
"""ONE rank's share of the P-GPU strong-scaling run, rehearsed on one GPU: builds vertex block ``--rank`` of the
config-5 graph exactly as bench.py --gpus P would (same generator, same pull/push cover plan, same kernels) with a
loop-back communicator, and reports the plan sizes and the measured per-iteration kernel time of that block.

The loop-back mirrors what the peers would ask of this rank: the bench graph's pattern and normalised values are
symmetric, so the plan of the pair (q <- r) is the plan of (r <- q) with rows and columns swapped -- computed here from
this rank's own cross entries.  The exchange itself copies this rank's outgoing rows into its own halo regions (same
sizes by that symmetry), so only link time is missing; predicted iteration time = max(kernels, halo bytes / link rate).

    python tools/sim_blocks.py --world 8 --rank 0 > gpurun_out/sim_blocks_p8.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import bench
import gnntf
from gnntf import sharded


class LoopbackComm(sharded.Comm):
    def __init__(self, world, rank, degrees, transport="copy"):
        self.group, self.solo, self.rank, self.size = None, False, rank, world
        self.degrees, self.sg, self.calls, self.transport = degrees, None, 0, transport

    def _staged(self, t):
        return False

    def barrier(self):
        pass

    def broadcast(self, t, src=0):
        return t

    def all_reduce(self, t, op=None):
        if t.numel() == self.degrees.numel() and t.dtype == torch.float32:
            t.copy_(self.degrees)                       # the global column sums (unit weights: the degrees)
        return t

    def exchange(self, send_chunks, recv_chunks):
        self.exchange_pairs(list(enumerate(send_chunks)), list(enumerate(recv_chunks)))

    def exchange_pairs(self, sends, recvs):
        pairs = [(s, r) for (_, s), (_, r) in zip(sends, recvs) if s is not None and r is not None and min(s.shape[0], r.shape[0]) > 0]
        per_peer = {}
        for (q, s), (_, r) in zip(sends, recvs):
            if s is not None and r is not None:
                per_peer[q] = per_peer.get(q, 0) + min(s.shape[0], r.shape[0]) * s.shape[1] * 4
        if self.transport == "rccl":
            # the same rows through REAL RCCL point-to-point kernels: this one-rank group is its own peer, so every message is an
            # ncclSend / ncclRecv pair inside one group -- no link, but RCCL's kernels, their CUs and the host cost of the batch
            import torch.distributed as dist
            ops = []
            for s, r in pairs:
                m = min(s.shape[0], r.shape[0])
                ops.append(dist.P2POp(dist.irecv, r[:m], 0))
            for s, r in pairs:
                m = min(s.shape[0], r.shape[0])
                ops.append(dist.P2POp(dist.isend, s[:m], 0))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return
        if self.transport.startswith("sleep"):
            # a transfer that takes link time but neither HBM bandwidth nor CUs: one spinning thread for as long as the busiest of the
            # (parallel) links would need at the given rate -- what is left of the step beyond max(kernels, exchange) is the schedule's
            worst = max(per_peer.values(), default=0)
            if worst:
                torch.cuda._sleep(int(worst / (self.link_GBs * 1e9) * self.spin_per_s))
            return
        for s, r in pairs:                               # about the same sizes by symmetry: stand in for the peer's rows
            m = min(s.shape[0], r.shape[0])
            r[:m].copy_(s[:m])

    def all_gather_vec(self, t):                        # push_counts table: table[q][me] = rows q asks me to sum for it
        table = [torch.zeros_like(t) for _ in range(self.size)]
        for q in range(self.size):
            if q != self.rank:
                table[q][self.rank] = self.n_pushed_for[q]
        table[self.rank] = t
        return table

    def alltoallv(self, chunks):
        """Called three times by ShardedGraph._build_block: pulled ids, push edges (slot, col), push values."""
        sg, P, me = self.sg, self.size, self.rank
        if self.calls == 0:
            self._mirror()
        kind = self.calls % 3
        self.calls += 1
        return [self.mirrored[q][kind] if q != me else chunks[me] for q in range(P)]

    def _mirror(self):
        sg, P, me = self.sg, self.size, self.rank
        rows_g, cols_g, nvals, _ = sg.entries
        dev = rows_g.device
        bnd = torch.tensor(sg.bounds[1:], dtype=torch.int64, device=dev)
        owner = torch.bucketize(cols_g, bnd, right=True)
        self.mirrored, self.n_pushed_for = {}, {}
        for q in range(P):
            if q == me:
                continue
            sel = owner == q
            lo_q, n_q = sg.bounds[q], sg.bounds[q + 1] - sg.bounds[q]
            t_row = cols_g[sel] - lo_q                                  # q's local row (my column j)
            t_col = rows_g[sel]                                         # global id of my row i: q's remote column
            t_own = torch.full_like(t_col, me)
            push = sharded.cover_push_mask(t_row, t_col, t_own, q, n_q, bnd, sg.push_weight) if sg.cover == "cover" else torch.zeros_like(sel[sel])
            asked = torch.unique(t_col[~push])                          # q pulls these rows of mine
            keys = torch.unique(t_row[push])                            # q's pushed rows for peer me, ascending = slot order
            slot = torch.searchsorted(keys, t_row[push]) if keys.numel() else torch.zeros(0, dtype=torch.int64, device=dev)
            edges = torch.stack([slot, t_col[push]], 1).reshape(-1)
            self.mirrored[q] = (asked, edges, nvals[sel][push])
            self.n_pushed_for[q] = int(keys.numel())
        return


class SimGraph(sharded.ShardedGraph):
    def _build_block(self, row, col, nvals, split_rows):
        self.comm.sg = self
        return super()._build_block(row, col, nvals, split_rows)


def spin_rate():
    """Cycles of torch.cuda._sleep per second on this card (the emulated link's clock)."""
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    return 20_000_000 / (e0.elapsed_time(e1) * 1e-3)


def sweep(a, idx, vals, bounds, degrees, dev):
    """The table of DESIGN section 5: one plan per entry of --sweep (a push weight of the weighted cover, or 'pull') on the same
    graph: what the plan puts on the busiest link, what it leaves to the sender's push SpMM, the block's kernels alone, and the
    K = 10 step with the exchange replaced by emulated link time at every rate of --sweep-rates (2 chunks, early pull)."""
    P, r, C = a.world, a.rank, a.feats
    rates = [float(x) for x in a.sweep_rates.split(",") if x]
    spin = spin_rate() if rates else None
    _skipped = [torch.cuda.Stream(dev) for _ in range(max(a.lane_skip, 1))]
    for plan in a.sweep.split(","):
        cover, weight = ("pull", 0.0) if plan == "pull" else ("cover", float(plan))
        t0 = time.time()
        comm = LoopbackComm(P, r, degrees, "copy")
        sg = SimGraph(idx, vals, bounds, comm=comm, cover=cover, chunks=a.chunks, split_rows=not a.whole_rows, keep_entries=True, push_weight=weight)
        sg.entries = None
        comm.mirrored = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        t_plan = time.time() - t0
        H0 = torch.rand(sg.n_local, C, device=dev) * 2 - 1
        state = sg.make_state(H0)
        t_c = sg.time_compute(state, 0.1)
        st = sg.stats
        rec = {"world": P, "plan": plan, "cover": cover, "push_weight": weight, "chunks": a.chunks, "plan_s": round(t_plan, 1),
               "pull_rows": st["pull_rows"], "push_rows": st["push_rows"], "halo_rows": st["pull_rows"] + st["push_rows"],
               "busiest_link_rows": st["busiest_link_rows"], "busiest_link_MB_per_iteration": st["busiest_link_rows"] * C * 4 / 1e6,
               "halo_GB_per_iteration": (st["pull_rows"] + st["push_rows"]) * C * 4 / 1e9, "push_entries": st["push_entries"],
               "local_entries": sg.nnz_local, "kernels_ms_per_iteration": t_c * 1e3, "step_over_10_ms": {}}
        for rate in rates:
            comm.transport, comm.link_GBs, comm.spin_per_s = f"sleep:{rate}", rate, spin
            for early in (False, True):
                sg.propagate(state, 0.1, 10, early_pull=early)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(3):
                    sg.propagate(state, 0.1, 10, early_pull=early)
                torch.cuda.synchronize()
                rec["step_over_10_ms"][f"{rate:g}_GBs" + ("_early_pull" if early else "")] = round((time.time() - t0) / 3 / 10 * 1e3, 3)
            rec.setdefault("exchange_ms_per_iteration", {})[f"{rate:g}_GBs"] = round(st["busiest_link_rows"] * C * 4 / (rate * 1e9) * 1e3, 3)
        print(json.dumps(rec), flush=True)
        del sg, state, H0, comm
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--graph", choices=["rmat", "community"], default="rmat", help="rmat: the bench graph (config 5); community: planted partition x power-law degrees")
    ap.add_argument("--row-window", type=int, default=0, help="with --locality: the block's handles take their rows in windows of this many ids (ShardedGraph(row_window=))")
    ap.add_argument("--locality", action="store_true", help="community graph: renumber the vertices by gnntf.ordering.locality_order before cutting the blocks")
    ap.add_argument("--nodes", type=int, default=80_000_000)
    ap.add_argument("--entries", type=int, default=1_000_000_000)
    ap.add_argument("--feats", type=int, default=128)
    ap.add_argument("--cover", default="cover")
    ap.add_argument("--push-weight", type=float, default=0.0, help="cover plans: weight of a pushed row's sender-side entries (cover_push_mask)")
    ap.add_argument("--sweep", default="",
                    help="comma list of plans -- a push weight, or 'pull' -- rehearsed one after the other on the SAME generated graph; with "
                         "--sweep-rates (GB/s per link and direction) every plan is also stepped under emulated link time.  Prints one JSON "
                         "line per plan: rows on the busiest link, entries in the push SpMM, kernels per iteration, step / 10 per rate")
    ap.add_argument("--sweep-rates", default="35,42,50,64")
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--whole-rows", action="store_true")
    ap.add_argument("--early-pull", action="store_true")
    ap.add_argument("--dummy-streams", type=int, default=0, help="streams created before the process group (transport rccl)")
    ap.add_argument("--transport", default="copy",
                    help="loop-back exchange by device copies (copy), through RCCL send / recv pairs of a one-rank group (rccl), or emulated "
                         "link time without traffic (sleep:GBs -- per link and direction)")
    ap.add_argument("--lane-skip", type=int, default=0, help="streams created (and kept) before the exchange lane's: moves the lane to another hardware queue")
    ap.add_argument("--pmc-iterations", type=int, default=0,
                    help="for rocprofv3 --pmc passes: run ONLY this many kernels-alone iterations (what bench.py times as compute_ms_alone: "
                         "SpMM of the interior and boundary rows + pack, every column chunk) between two marker launches (k_stream), print "
                         "the plan sizes and stop -- profiles/summarize_blocks.py turns the passes into the block's pmc_traffic.json entry")
    a = ap.parse_args()
    if a.transport == "rccl":
        import torch.distributed as dist
        _keep = [torch.cuda.Stream(torch.device("cuda:0")) for _ in range(a.dummy_streams)]      # shifts which hardware queue RCCL's stream lands on
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    P, r, N = a.world, a.rank, a.nodes
    t0 = time.time()
    if a.graph == "community":
        # a graph WITH communities (gnntf.rmat.community_pairs, every pair once) -- and, with --locality, its vertices renumbered
        # community by community (gnntf.ordering.locality_order) BEFORE the cut into contiguous blocks: what the exchange of such a
        # graph would be under the order GNN(reorder="locality") computes
        from gnntf import ordering
        from gnntf.rmat import community_pairs
        u, w, _ = community_pairs(N, a.entries // 2, 1, dev)
        keys = torch.unique(torch.minimum(u, w) * N + torch.maximum(u, w))
        u, w = torch.div(keys, N, rounding_mode="floor"), keys % N
        del keys
        if a.locality:
            order = ordering.locality_order(torch.cat([torch.stack([u, w], 1), torch.stack([w, u], 1)]), N)
            newid = torch.empty_like(order)
            newid[order] = torch.arange(N, device=dev)
            u, w = newid[u], newid[w]
            del order, newid
    else:
        u, w = sharded.rmat_relabelled_pairs(N, a.entries // 2, seed=1, device=dev)
    degrees = (torch.bincount(u, minlength=N) + torch.bincount(w, minlength=N)).float()
    bounds = sharded.uniform_bounds(N, P)
    lo, hi = bounds[r], bounds[r + 1]
    mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
    idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
    del u, w, mu, mw
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    if a.sweep:
        return sweep(a, idx, vals, bounds, degrees, dev)
    t0 = time.time()
    comm = LoopbackComm(P, r, degrees, a.transport)
    if a.transport.startswith("sleep"):
        comm.link_GBs = float(a.transport.split(":")[1])
        comm.spin_per_s = spin_rate()
    _skipped = [torch.cuda.Stream(dev) for _ in range(a.lane_skip)]
    sg = SimGraph(idx, vals, bounds, comm=comm, cover=a.cover, chunks=a.chunks, split_rows=not a.whole_rows, keep_entries=True,
                  push_weight=a.push_weight, row_window=a.row_window)
    sg.entries = None
    del idx, vals, comm.mirrored
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t_plan = time.time() - t0
    C = a.feats
    H0 = torch.rand(sg.n_local, C, device=dev) * 2 - 1
    state = sg.make_state(H0)
    if a.pmc_iterations > 0:
        from gnntf import _native as nat
        src, sink = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        mark = lambda: nat.check(nat.lib().gnx_stream_read(nat.ptr(src), 64, nat.ptr(sink), nat.current_stream()))
        sg.time_compute(state, 0.1, repeats=0)                          # warm-up: slabs, lazy structures
        torch.cuda.synchronize()
        mark()
        t_c = sg.time_compute(state, 0.1, repeats=a.pmc_iterations - 1)
        mark()
        torch.cuda.synchronize()
        print(json.dumps({"world": P, "rank": r, "cover": a.cover, "chunks": a.chunks, "iterations_between_markers": a.pmc_iterations,
                          "kernels_ms_per_iteration": t_c * 1e3, "rows": sg.n_local, "entries": sg.nnz_local, "features": C, "stats": sg.stats}))
        return
    t_c = sg.time_compute(state, 0.1)
    t_x = sg.time_exchange(state)                                       # loop-back copies: local HBM traffic only

    def part_ms(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t0) / reps * 1e3
    every = range(len(state.cols))
    src = lambda c: state.bufs[c][0]
    dst = lambda c: sg.local_view(state.bufs[c][1])
    breakdown = {
        "interior_rows": part_ms(lambda: [sg._compute(state, c, src(c), dst(c), 0.1, interior=True, skip_empty=True) for c in every]),
        "boundary_rows": part_ms(lambda: [sg._compute(state, c, src(c), dst(c), 0.1, interior=False, skip_empty=True) for c in every]),
        "pack": part_ms(lambda: [sg._pack(state, c, state.bufs[c][1]) for c in every]),
        "pack_pulled_rows": part_ms(lambda: [sg._pack(state, c, state.bufs[c][1], "pull") for c in every]),
        "pack_pushed_sums": part_ms(lambda: [sg._pack(state, c, state.bufs[c][1], "push") for c in every]),
        "first_iterations_interior_rows": part_ms(lambda: [sg._compute(state, c, src(c), dst(c), 0.1, interior=True, skip_empty=False) for c in every]),
    }
    sg.propagate(state, 0.1, 10, early_pull=a.early_pull)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        sg.propagate(state, 0.1, 10, early_pull=a.early_pull)
    torch.cuda.synchronize()
    t_step = (time.time() - t0) / 3
    st = sg.stats
    halo_rows = st["pull_rows"] + st["push_rows"]
    out = {"world": P, "rank": r, "graph": {"kind": a.graph, "locality_order": bool(a.locality), "row_window": a.row_window, "nodes": N, "entries": a.entries, "features": C}, "options": {"cover": a.cover, "chunks": a.chunks,
           "split_rows": bool(sg.split_rows), "early_pull": a.early_pull, "transport": a.transport, "lane_skip": a.lane_skip}, "gen_s": round(t_gen, 2), "plan_s": round(t_plan, 2), "stats": st,
           "local_entries": sg.nnz_local, "push_graph_entries": (sg.push_graph.nnz if sg.push_graph is not None else 0), "halo_rows": halo_rows, "halo_bytes_per_iteration": halo_rows * C * 4,
           "pull_only_bytes_per_iteration": st["pull_only_rows"] * C * 4, "kernels_ms_per_iteration": t_c * 1e3,
           "kernels_breakdown_ms": breakdown, "loopback_copy_ms_per_iteration": t_x * 1e3, "step_ms_K10_loopback": t_step * 1e3,
           "predicted_iteration_ms": {f"{bw}_GBs_per_link": max(t_c * 1e3, halo_rows * C * 4 / ((P - 1) * bw * 1e9) * 1e3) for bw in (30, 45, 60, 75)},
           "kernel": sg.graph.last_kernel()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
