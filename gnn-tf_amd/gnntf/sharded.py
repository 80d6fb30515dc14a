"""Multi-GPU propagation (one process per GPU): 1-D vertex blocks with a pairwise halo exchange per
iteration, optionally crossed with feature slices (a pv x pf process grid).

The reference has no distributed code at all; this module is the multi-GPU form of
the same hot path -- PPRIteration.__forward__ (reference gnntf/core/gnn/architectures/filter.py:17-22)
over a normalised adjacency (gnntf/core/gnn/gnn.py:36-50) -- and must give the same logits as one GPU.

Rank r owns the global rows [lo, hi) of A_hat and the matching rows of H, H0 and the result.  For every
entry (i, j) whose column lives on another rank q there are two ways to get its contribution:

  pull   q sends the row H[j]; this rank multiplies it itself (the classic halo);
  push   q, which holds H[j] anyway, sums  A_hat[i, j] * H[j]  over ITS columns j of row i and sends that
         one partial-sum row; this rank only adds it (an entry of weight 1 on a "push slot" column).

Either way one feature row crosses the link, so what the exchange costs is the size of a VERTEX COVER of the
bipartite graph of cross entries between the two blocks: a pulled column covers all its entries, a pushed
row covers all of its.  On power-law graphs the hubs on either side cover most entries: the cover is about
half of the distinct remote columns a pull-only halo moves (``cover="pull"`` keeps the pull-only plan, whose
per-row summation order -- and therefore every bit of the result -- equals the one-GPU kernel's).

Layout per rank:
  * X = [ region(0) .. region(r-1) | local rows | region(r+1) .. region(P-1) ],  region(q) = [ rows pulled
    from q (ascending global id) | partial sums pushed by q ] -- one contiguous message per peer, and the
    column remap of pulled rows stays MONOTONIC in the global id;
  * the main CSR over X (interior rows -- no remote column -- and boundary rows as two handles when
    ``split_rows``), values already normalised with GLOBAL column sums (gnn.py:41-42);
  * a send buffer [rows the peers pull, peer by peer | partial sums pushed to the peers, peer by peer]: the first half is
    a gather of local rows (short), the second the product of the PUSH graph [pushed rows x local rows] with the local
    rows (most of the pack time); the layout and both packing launches are the library's (gnx_halo_plan_*).

Each iteration and column chunk: pack -> pairwise isend/irecv (RCCL group of point-to-point transfers:
every xGMI link carries its own peer's rows; no ring) -> fused SpMM+mix.  Overlap comes from three sources:
the feature columns are cut into ``chunks`` that propagate independently (filter.py:19-21 acts on every
column alike), so the exchange of one chunk runs on its own stream under the SpMM of the other; the
interior rows of a chunk are computed before its halo is waited for; and with ``early_pull`` the pulled rows
leave as soon as they are gathered, while the pushed partial sums are still being summed.

The heavy lifting goes through a small backend object; the product backend is libgnx.so (NativeBackend).
Tests on CPU ranks (gloo) supply their own checker backend: this module never imports a CPU implementation.
"""
from __future__ import annotations

import contextlib
import os
import time

import torch
import torch.distributed as dist

from . import _native as nat
from . import sparse
from .comm import Comm, choose_grid, grid_cost_ms, make_grid                  # noqa: F401  (historical home of these names)
from .shard_backend import NativeBackend, NativeHaloPlan                     # noqa: F401


def uniform_bounds(n_global, world):
    return [r * n_global // world for r in range(world + 1)]


def split_columns(C, chunks):
    """Column ranges of the ``chunks`` independent feature chunks (multiples of 32 columns = whole 128-byte
    lines wherever the width allows, so every chunk keeps aligned float4 rows)."""
    chunks = max(1, min(int(chunks), C))
    unit = 32 if C % 32 == 0 and C // 32 >= chunks else (4 if C % 4 == 0 and C // 4 >= chunks else 1)
    units = C // unit
    cuts = [(units * k // chunks) * unit for k in range(chunks + 1)]
    return [(cuts[k], cuts[k + 1]) for k in range(chunks) if cuts[k + 1] > cuts[k]]


def cover_push_mask(row, col, owner, rank, n_local, bnd, push_weight=0.0):
    """Which cross entries are PUSHED (summed by the column's owner) instead of pulled: a greedy WEIGHTED vertex cover of
    the cross entries between this block and each peer.  Choosing a column j (pull) costs one row on the link and one gather
    on the sender: weight 1 + w; choosing a row i (push) costs one row on the link and its d_i entries on that peer in the
    SENDER's push SpMM -- work that must finish before the row can leave: weight 1 + w d_i (w = ``push_weight``).  An entry
    (i, j) is first given to the endpoint with the lower weight per entry it covers: the column when
    d_i (1 + w) <= c_j (1 + w d_i), c_j = rows of this block referencing j.  w = 0 is the plain minimum-rows cover (the column
    if at least as many rows reference it as row i has columns on q); as w grows the plan approaches the pull-only halo (in the
    limit only rows whose columns nobody else references stay pushed: fewer rows on the link for the same work); in between it
    trades rows on the link for partial sums that delay them (the table over w).  Then every entry of a row
    that is pushed anyway joins the push (no further row on the link), which also frees the columns only such rows
    referenced.  A peer whose cover does not come out smaller than its plain halo keeps the plain halo.  ANY mask is a valid
    plan (every cross entry is multiplied exactly once, by one side): the weight changes cost, never results beyond float32
    summation order.
    ``row`` local ids, ``col`` global ids, ``owner`` rank owning each column (all int64 [m]); ``bnd`` the upper
    bounds of the P blocks.  Returns bool [m]."""
    world, n_global = int(bnd.numel()), int(bnd[-1])
    remote = owner != rank
    push = torch.zeros_like(remote)
    if world == 1 or not bool(remote.any()):
        return push
    dev = row.device
    er, ec, eq = row[remote], col[remote], owner[remote]
    kr = eq * n_local + er                                           # (peer, row) key
    dr = torch.bincount(kr, minlength=world * n_local)               # entries of row i on peer q
    dc = torch.bincount(ec, minlength=n_global)                      # rows of this block referencing column j
    if push_weight and push_weight > 0:
        w, d_row, c_col = float(push_weight), dr[kr].to(torch.float64), dc[ec].to(torch.float64)
        col_wins = d_row * (1.0 + w) <= c_col * (1.0 + w * d_row)
        del d_row, c_col
    else:
        col_wins = dc[ec] >= dr[kr]
    halo_per_peer = torch.bincount(torch.bucketize(torch.nonzero(dc).reshape(-1), bnd, right=True), minlength=world)
    del dr, dc
    pulled_col = torch.zeros(n_global, dtype=torch.bool, device=dev)
    pulled_col[ec[col_wins]] = True
    pushed_row = torch.zeros(world * n_local, dtype=torch.bool, device=dev)
    pushed_row[kr[~pulled_col[ec]]] = True
    pushed = pushed_row[kr]
    # per peer: rows the cover moves (columns still pulled + rows pushed) against the plain halo
    still = torch.zeros(n_global, dtype=torch.bool, device=dev)
    still[ec[~pushed]] = True
    cover_per_peer = torch.bincount(torch.bucketize(torch.nonzero(still).reshape(-1), bnd, right=True), minlength=world) \
        + torch.bincount(torch.div(torch.nonzero(pushed_row).reshape(-1), max(n_local, 1), rounding_mode="floor"), minlength=world)
    pushed &= (cover_per_peer < halo_per_peer)[eq]
    push[remote] = pushed
    return push


def max_relative_deviation(got, want, rows_per_pass=1 << 22):
    """max |got - want| / max(|want|, 1), in row slabs so that the temporaries stay small next to 40 GB operands."""
    worst = 0.0
    for r0 in range(0, got.shape[0], rows_per_pass):
        g, w = got[r0:r0 + rows_per_pass], want[r0:r0 + rows_per_pass]
        d = float(((g - w).abs() / w.abs().clamp_min(1.0)).max())
        if d != d:                                   # a NaN anywhere is a failure, not a value max() may skip
            return float("inf")
        worst = max(worst, d)
    return worst


class _Lanes:
    """Two in-order lanes -- compute (the caller's stream) and exchange (a stream of its own) -- joined by
    events.  On CPU ranks there are no streams and everything runs in program order."""

    def __init__(self, device, exchange_stream=None):
        self.gpu = device.type == "cuda"
        self.device = device
        self._comm = (exchange_stream if exchange_stream is not None else torch.cuda.Stream(device)) if self.gpu else None

    def exchange_lane(self):
        return torch.cuda.stream(self._comm) if self.gpu else contextlib.nullcontext()

    def mark(self, on_exchange_lane=False):
        if not self.gpu:
            return None
        ev = torch.cuda.Event()
        ev.record(self._comm if on_exchange_lane else torch.cuda.current_stream(self.device))
        return ev

    def wait(self, ev, on_exchange_lane=False):
        if ev is not None:
            (self._comm if on_exchange_lane else torch.cuda.current_stream(self.device)).wait_event(ev)


class ShardState:
    """Buffers of one propagation: per column chunk two ping-pong [regions | local | regions] buffers and a
    send buffer; H0 and the result hold this rank's rows at full width."""

    def __init__(self, H0):
        self.H0 = H0
        self.cur = 0


class ShardedGraph:
    """This rank's block of a symmetrically normalised, vertex-partitioned square graph."""

    MIN_INTERIOR_SHARE = 0.05      # share of a block's entries that must sit in interior rows for the interior / boundary split to pay

    def __init__(self, idx_global, vals, bounds, backend=None, group=None, normalized="symmetric", comm=None,
                 relabel=False, cover="cover", split_rows=True, chunks=2, keep_entries=False, edge_dropout=False, early_pull=False,
                 tune_overlap=True, push_weight=0.0, row_window=0):
        """``idx_global``: int64 [nnz, 2] (global row, global col) of the entries whose row this rank owns
        (unsorted, duplicates allowed); ``bounds``: the P+1 partition boundaries.  Collective: every rank of the
        vertex partition (``comm`` / ``group``) must call it with the same options.
        ``cover``: "cover" (pull/push vertex cover, default) or "pull" (classic halo; bitwise the one-GPU sums).
        ``row_window``: the GLOBAL numbering is a locality order (communities contiguous: gnntf.ordering.locality_order applied before
        the cut into blocks) -- the block's handles then take their rows in windows of this many consecutive local ids
        (gnx_graph_set_row_window: one window per XCD at a time).  Same sums; only for graphs that have communities.
        ``push_weight`` (cover plans): what a pushed row's entries on the sender weigh against a row on the link
        (cover_push_mask; 0 = fewest rows on the link, larger = fewer and shorter partial sums, more pulled rows).
        ``split_rows``: interior rows (no remote column) as a handle of their own, computed before the halo is
        waited for -- when they hold at least MIN_INTERIOR_SHARE of the block's entries ("always": whatever they hold).
        ``chunks``: independent column chunks whose exchange and SpMM overlap (default 2).
        ``keep_entries``: keep (global row, global col, normalised value, pushed?) of this rank's entries in
        ``self.entries`` (tests).  ``relabel`` (single vertex block only): store the shard with its vertices relabelled in stable order of
        descending entry count -- a legal preprocessing step that makes the sub-wave kernels
        5-20 % faster; propagate() permutes H0 on the way in and the result on the way out.
        ``early_pull``: propagate() sends the pulled rows ahead of the pushed partial sums (see propagate).
        ``tune_overlap``: probe once per communicator where the exchange runs beside the compute stream (Comm.tune_overlap;
        collective, capped in time, GNX_TUNE_OVERLAP=0 disables it as well).
        ``edge_dropout``: build the block for TRAINING with per-iteration edge dropout (layered.py:47-50 + gnn.py:41-42):
        raw values (``normalized`` is ignored -- every iteration re-normalises its own dropped entries), classic halo, whole
        rows; use dropped_scales / propagate_dropped / propagate_dropped_backward instead of propagate()."""
        if cover not in ("cover", "pull"):
            raise Exception("ShardedGraph: cover must be 'cover' or 'pull'")
        self.edge_dropout = bool(edge_dropout)
        if self.edge_dropout:
            normalized, cover, split_rows, relabel = "none", "pull", False, False
            if int(bounds[-1]) >= 2 ** 31:
                raise Exception("ShardedGraph: edge dropout across blocks keys its draws by int32 vertex ids")
        self.backend = backend if backend is not None else NativeBackend()
        self.comm = comm if comm is not None else Comm(group=group)
        self.group = self.comm.group
        self.rank, self.world = self.comm.rank, self.comm.size
        be = self.backend
        dev = idx_global.device
        self.device = dev
        self.bounds = [int(b) for b in bounds]
        if len(self.bounds) != self.world + 1:
            raise Exception("ShardedGraph: bounds must list world + 1 boundaries")
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        N = self.bounds[-1]
        self.lo, self.hi, self.n_global, self.n_local = lo, hi, N, hi - lo
        self.cover, self.chunks, self.early_pull = cover, max(1, int(chunks)), bool(early_pull)
        self.row_window = int(row_window)
        self.push_weight = float(push_weight) if cover == "cover" else 0.0
        if self.push_weight < 0:
            raise Exception("ShardedGraph: push_weight must not be negative")
        self.n_send_pull_max = self.n_send_push_max = 0
        if idx_global.numel() and (int(idx_global[:, 0].min()) < lo or int(idx_global[:, 0].max()) >= hi):
            raise Exception("ShardedGraph: an entry's row is outside this rank's range [%d, %d)" % (lo, hi))

        # local rows x global columns: coalesce, then normalise with GLOBAL column sums (gnn.py:41-42)
        local_idx = idx_global.clone()
        local_idx[:, 0] -= lo
        g0 = be.graph_from_coo(local_idx, vals, (self.n_local, N))
        del local_idx
        rowptr, colidx, raw, rowidx = be.csr_arrays(g0, with_rows=True)
        if normalized == "symmetric":
            deg = be.colsum(g0)
            self.comm.all_reduce(deg)
            self.sqrt_degree = deg[lo:hi].sqrt()        # this rank's part of the eigenvector D^1/2 1 (eigenvalue 1 when A is symmetric)
            D = be.degree_scale(deg, "symmetric")
            nvals = be.scale_values(g0, D[lo:hi], D)
            del deg, D
        elif normalized == "none":
            nvals, self.sqrt_degree = raw, None
        else:
            raise Exception("Invalid matrix normalization")
        del g0, raw
        self.row_order = None
        self.nnz_local = int(colidx.numel())
        if self.edge_dropout and self.nnz_local != int(idx_global.shape[0]):
            raise Exception("ShardedGraph: edge dropout across blocks needs a COO without duplicate entries")
        t = torch.tensor([self.nnz_local], dtype=torch.int64, device=dev)
        self.comm.all_reduce(t)
        self.nnz_global = int(t.item())
        self._lanes = _Lanes(dev)
        self.entries = None
        if keep_entries:
            self.entries = [rowidx.to(torch.int64) + lo, colidx.to(torch.int64), nvals.clone(), None]

        if self.world == 1:
            self._build_single_block(rowptr, colidx, nvals, relabel)
        else:
            if tune_overlap and hasattr(self.comm, "tune_overlap"):
                # (once per Comm) an exchange-lane stream whose transfers run beside the compute stream.  Every failure inside is agreed
                # on by all ranks before anyone continues (see Comm.tune_overlap), so nothing is caught here: an exception that does
                # escape is a bug worth seeing, not a state to continue from with ranks that may have diverged
                self.comm.tune_overlap(dev)
                self.group = self.comm.group
                if getattr(self.comm, "lane_stream", None) is not None:
                    self._lanes = _Lanes(dev, self.comm.lane_stream)
            self._build_block(rowidx.to(torch.int64), colidx.to(torch.int64), nvals, split_rows)

    # ---- one vertex block: no exchange -----------------------------------------------------------------
    def _build_single_block(self, rowptr, colidx, nvals, relabel):
        be, dev = self.backend, self.device
        if relabel and colidx.numel() > 0:
            deg = rowptr[1:] - rowptr[:-1]
            order = torch.argsort(deg, descending=True, stable=True)                 # new id -> old id
            newid = torch.empty_like(order)
            newid[order] = torch.arange(order.numel(), device=dev)
            rows = torch.repeat_interleave(torch.arange(self.n_local, device=dev), deg)
            g = be.graph_from_coo(torch.stack([newid[rows], newid[colidx.to(torch.int64)]], dim=1), nvals,
                                  (self.n_local, self.n_global))
            self.row_order, self.row_newid, self.row_order32 = order, newid, order.to(torch.int32)
        else:
            g = be.graph_from_csr(rowptr, colidx, nvals, (self.n_local, self.n_local))
        self.graph = g
        self.n_buf, self.n_before = self.n_local, 0
        self.recv_counts, self.send_counts = [0], [0]
        self.pull_counts, self.push_counts = [0], [0]
        self.n_send = self.n_send_pull = 0
        self.send_graph = self.push_graph = self.halo = None
        self.empty_rows_unreferenced = False
        self.stats = dict(pull_rows=0, push_rows=0, pull_only_rows=0, send_rows=0, interior_rows=self.n_local,
                          boundary_rows=0, local_rows=self.n_local, busiest_link_rows=0, push_entries=0)

    # ---- a block among several: halo plan --------------------------------------------------------------
    def _build_block(self, row, col, nvals, split_rows):
        be, dev, comm = self.backend, self.device, self.comm
        P, me, n_local, lo = self.world, self.rank, self.n_local, self.lo
        bnd = torch.tensor(self.bounds[1:], dtype=torch.int64, device=dev)
        owner = torch.bucketize(col, bnd, right=True)
        remote = owner != me
        push = cover_push_mask(row, col, owner, me, n_local, bnd, self.push_weight) if self.cover == "cover" else torch.zeros_like(remote)
        pull = remote & ~push
        if self.entries is not None:
            self.entries[3] = push.clone()
        n_pull_only = int(torch.unique(col[remote]).numel())                     # what a pull-only halo would move

        halo = torch.unique(col[pull])                                           # pulled global ids, ascending (= grouped by owner)
        h_owner = torch.bucketize(halo, bnd, right=True)
        pull_counts = torch.bincount(h_owner, minlength=P)
        kpush = torch.unique(owner[push] * n_local + row[push])                  # pushed (peer, row) keys, ascending
        p_owner = torch.div(kpush, max(n_local, 1), rounding_mode="floor")
        push_counts = torch.bincount(p_owner, minlength=P)
        recv_counts = pull_counts + push_counts
        excl = lambda c: torch.cumsum(c, 0) - c
        region_start = excl(recv_counts) + (torch.arange(P, device=dev) > me).to(torch.int64) * n_local
        n_before = int(excl(recv_counts)[me])
        self.n_before = n_before
        self.n_buf = n_local + int(recv_counts.sum())
        halo_col = region_start[h_owner] + torch.arange(halo.numel(), device=dev) - excl(pull_counts)[h_owner]
        slot_col = region_start[p_owner] + pull_counts[p_owner] + torch.arange(kpush.numel(), device=dev) - excl(push_counts)[p_owner]
        self.halo_ids = halo

        # entries this rank multiplies itself: local + pulled columns, plus weight-1 entries on the push slots
        keep = ~push
        kcol = col[keep]
        krow = row[keep]
        klocal = ~remote[keep]
        pos = torch.searchsorted(halo, kcol) if halo.numel() else torch.zeros_like(kcol)
        pulled_col = halo_col[pos.clamp(max=max(halo.numel() - 1, 0))] if halo.numel() else torch.zeros_like(kcol)
        new_col = torch.where(klocal, kcol - lo + n_before, pulled_col)
        m_rows = torch.cat([krow, kpush % max(n_local, 1)])
        m_cols = torch.cat([new_col, slot_col])
        m_vals = torch.cat([nvals[keep], torch.ones(kpush.numel(), dtype=torch.float32, device=dev)])
        is_bnd = torch.zeros(n_local, dtype=torch.bool, device=dev)
        is_bnd[row[remote]] = True
        del new_col, pulled_col, pos, kcol, krow, klocal, keep

        # ---- tell every peer what to send: pulled ids + the entries of the rows it sums for me ---------------
        push_slot = torch.searchsorted(kpush, owner[push] * n_local + row[push]) if kpush.numel() else torch.zeros(0, dtype=torch.int64, device=dev)
        p_own_e = owner[push]
        order = torch.argsort(p_own_e, stable=True)
        e_slot = (push_slot - excl(push_counts)[p_own_e])[order]                 # slot inside the peer's push list
        e_col = col[push][order]
        e_val = nvals[push][order]
        e_counts = torch.bincount(p_own_e, minlength=P)
        e_off = [0] + torch.cumsum(e_counts, 0).tolist()
        p_off = [0] + torch.cumsum(pull_counts, 0).tolist()
        asked_ids = comm.alltoallv([halo[p_off[q]:p_off[q + 1]] for q in range(P)])
        edges = comm.alltoallv([torch.stack([e_slot[e_off[q]:e_off[q + 1]], e_col[e_off[q]:e_off[q + 1]]], 1).reshape(-1)
                                for q in range(P)])
        evals = comm.alltoallv([e_val[e_off[q]:e_off[q + 1]] for q in range(P)])
        table = comm.all_gather_vec(push_counts)                                 # table[g][q]: rows q pushes to g
        del push_slot, p_own_e, order, e_slot, e_col, e_val, row, col, owner, remote, push, pull, nvals

        self.pull_counts = [int(c) for c in pull_counts.tolist()]
        self.push_counts = [int(c) for c in push_counts.tolist()]
        self.recv_counts = [a + b for a, b in zip(self.pull_counts, self.push_counts)]
        rs = [int(x) for x in region_start.tolist()]
        self.recv_slices = [(rs[q], rs[q] + self.recv_counts[q]) for q in range(P)]
        self.recv_pull_slices = [(rs[q], rs[q] + self.pull_counts[q]) for q in range(P)]
        self.recv_push_slices = [(rs[q] + self.pull_counts[q], rs[q] + self.recv_counts[q]) for q in range(P)]
        # the send buffer: [rows the peers pull, peer by peer | partial sums pushed to the peers, peer by peer].  The first half is
        # a gather of local rows, the second the product of the push graph with the local rows (gnx_halo_pack).
        self.send_pull_counts = [0 if g == me else int(asked_ids[g].numel()) for g in range(P)]
        self.send_push_counts = [0 if g == me else int(table[g][me]) for g in range(P)]
        self.send_counts = [a + b for a, b in zip(self.send_pull_counts, self.send_push_counts)]
        self.n_send_pull, n_push = sum(self.send_pull_counts), sum(self.send_push_counts)
        self.n_send = self.n_send_pull + n_push
        self.send_pull_slices, self.send_push_slices = [], []
        at_pull, at_push = 0, self.n_send_pull
        s_rows, s_cols, s_vals = [], [], []
        for g in range(P):
            a_g, b_g = self.send_pull_counts[g], self.send_push_counts[g]
            if g != me:
                e = edges[g].reshape(-1, 2)
                if b_g == 0 and e.shape[0] != 0 or (e.shape[0] and int(e[:, 0].max()) >= b_g):
                    raise Exception("ShardedGraph: a peer's push plan is inconsistent")
                s_rows.append(at_push - self.n_send_pull + e[:, 0])
                s_cols.append(e[:, 1] - lo)
                s_vals.append(evals[g])
            self.send_pull_slices.append((at_pull, at_pull + a_g))
            self.send_push_slices.append((at_push, at_push + b_g))
            at_pull, at_push = at_pull + a_g, at_push + b_g
        pull_src = torch.cat([asked_ids[g] - lo for g in range(P) if g != me]) if P > 1 else torch.zeros(0, dtype=torch.int64, device=dev)
        if pull_src.numel() and (int(pull_src.min()) < 0 or int(pull_src.max()) >= n_local):
            raise Exception("ShardedGraph: a peer asked for a row this rank does not own")
        self.send_pull_src = pull_src.to(torch.int32)
        s_rows = torch.cat(s_rows) if s_rows else torch.zeros(0, dtype=torch.int64, device=dev)
        s_cols = torch.cat(s_cols) if s_cols else torch.zeros(0, dtype=torch.int64, device=dev)
        s_vals = torch.cat(s_vals) if s_vals else torch.zeros(0, dtype=torch.float32, device=dev)
        if s_cols.numel() and (int(s_cols.min()) < 0 or int(s_cols.max()) >= n_local):
            raise Exception("ShardedGraph: a peer asked for a partial sum over rows this rank does not own")
        # Is any local row WITHOUT entries used by somebody -- as a local column of this block, as a row a peer pulls, or inside a
        # partial sum pushed to a peer?  If none is (always so for a symmetric pattern: such a vertex is isolated), nobody ever
        # gathers those rows: propagate() writes them into the result only, never into the ping-pong buffers
        used = torch.zeros(max(n_local, 1), dtype=torch.bool, device=dev)
        local_cols = m_cols[(m_cols >= n_before) & (m_cols < n_before + n_local)] - n_before
        used[local_cols] = True
        used[pull_src] = True
        used[s_cols] = True
        has_entries = torch.zeros(max(n_local, 1), dtype=torch.bool, device=dev)
        has_entries[m_rows] = True
        self.empty_rows_unreferenced = not bool((used & ~has_entries)[:n_local].any())
        del used, has_entries, local_cols
        self.push_graph = be.graph_from_coo(torch.stack([s_rows, s_cols], 1), s_vals, (n_push, n_local)) if n_push else None
        self.halo = be.halo_plan(me, n_local, self.pull_counts, self.push_counts, self.send_pull_counts, self.send_push_counts,
                                 self.send_pull_src, self.push_graph)
        # the library's layout of the two buffers is THE layout: what this module computed above must agree with it
        if (self.halo.n_buf, self.halo.local_row0, self.halo.n_send, self.halo.n_send_pull) != (self.n_buf, n_before, self.n_send, self.n_send_pull) \
                or any(self.halo.recv_row0[q] != rs[q] for q in range(P) if self.recv_counts[q]) \
                or any(self.halo.send_pull_row0[q] != self.send_pull_slices[q][0] for q in range(P) if self.send_pull_counts[q]) \
                or any(self.halo.send_push_row0[q] != self.send_push_slices[q][0] for q in range(P) if self.send_push_counts[q]):
            raise Exception("ShardedGraph: the halo plan's layout disagrees with the block's")
        # training with edge dropout sends what this rank computed for its halo columns BACK to their owners, who add it up
        # through the transposed send graph (a pulled row = one entry of weight 1); only that mode needs it as a graph
        self.send_graph = None
        if self.edge_dropout and self.n_send_pull:
            self.send_graph = be.graph_from_coo(torch.stack([torch.arange(self.n_send_pull, device=dev), pull_src], 1),
                                                torch.ones(self.n_send_pull, dtype=torch.float32, device=dev), (self.n_send_pull, n_local))
        del s_rows, s_cols, s_vals, asked_ids, edges, evals, pull_src

        # ---- the main CSR over X; interior rows (no remote column: their sums need no halo) apart -------------
        n_bnd = int(is_bnd.sum())
        sel = is_bnd[m_rows]
        # interior rows in a handle of their own only when they hold real work to put under the exchange: on a randomly
        # partitioned power-law graph they are almost all isolated vertices, and a launch that finds nothing to do is pure cost
        interior_share = 1.0 - float(sel.sum()) / max(int(sel.numel()), 1)
        self.split_rows = 0 < n_bnd < n_local and (split_rows == "always" or (bool(split_rows) and interior_share >= self.MIN_INTERIOR_SHARE))
        if self.split_rows:
            rank_b = torch.cumsum(is_bnd.to(torch.int64), 0) - 1
            rank_i = torch.cumsum((~is_bnd).to(torch.int64), 0) - 1
            self.rows_bnd = torch.nonzero(is_bnd).reshape(-1).to(torch.int32)
            self.rows_int = torch.nonzero(~is_bnd).reshape(-1).to(torch.int32)
            self.graph = be.graph_from_coo(torch.stack([rank_b[m_rows[sel]], m_cols[sel]], 1), m_vals[sel], (n_bnd, self.n_buf))
            self.graph_int = be.graph_from_coo(torch.stack([rank_i[m_rows[~sel]], m_cols[~sel]], 1), m_vals[~sel],
                                               (n_local - n_bnd, self.n_buf))
        else:
            self.graph = be.graph_from_coo(torch.stack([m_rows, m_cols], 1), m_vals, (n_local, self.n_buf))
            self.graph_int, self.rows_bnd, self.rows_int = None, None, None
        if self.row_window > 0:                                # the caller's numbering carries locality: row windows on the block's handles
            for handle in (self.graph, self.graph_int):
                if handle is not None and hasattr(handle, "set_row_window"):
                    handle.set_row_window(self.row_window)
        if self.edge_dropout:                                  # dropout draws keyed by the GLOBAL (row, col) of every entry
            gid = torch.empty(self.n_buf, dtype=torch.int32, device=dev)
            gid[halo_col] = halo.to(torch.int32)
            gid[n_before:n_before + n_local] = torch.arange(lo, lo + n_local, dtype=torch.int32, device=dev)
            self.col_gid = gid
            be.set_block(self.graph, lo, n_before, gid)
        self.stats = dict(pull_rows=sum(self.pull_counts), push_rows=sum(self.push_counts), pull_only_rows=n_pull_only,
                          send_rows=self.n_send, interior_rows=n_local - n_bnd, boundary_rows=n_bnd, local_rows=n_local,
                          # every pair of ranks has its own link: what an exchange lasts is the BUSIEST one, either direction
                          busiest_link_rows=max(self.recv_counts + self.send_counts),
                          push_entries=(self.push_graph.nnz if self.push_graph is not None else 0))
        t = torch.tensor([self.n_send_pull, self.n_send - self.n_send_pull], dtype=torch.int64, device=dev)
        comm.all_reduce(t, dist.ReduceOp.MAX)
        self.n_send_pull_max, self.n_send_push_max = int(t[0]), int(t[1])

    # ---- propagation -----------------------------------------------------------------------------
    def local_view(self, buf):
        return buf[self.n_before:self.n_before + self.n_local]

    def make_state(self, H0, chunks=None):
        """Allocates the buffers for features of H0's width (this rank's rows)."""
        H0 = H0.to(torch.float32).contiguous()
        if H0.shape[0] != self.n_local:
            raise Exception("make_state: H0 must hold this rank's %d rows" % self.n_local)
        state = ShardState(H0)
        C, dev = H0.shape[1], H0.device
        if self.world == 1:
            state.bufs = [torch.zeros((self.n_local, C), dtype=torch.float32, device=dev) for _ in range(2)]
            if self.row_order is not None:                             # relabelled shard: H0 in the new order, result in the old
                state.H0_user, state.H0 = H0, H0.index_select(0, self.row_order)
                state.result = torch.empty_like(H0)
            return state
        state.cols = split_columns(C, self.chunks if chunks is None else chunks)
        state.bufs = [[torch.zeros((self.n_buf, c1 - c0), dtype=torch.float32, device=dev) for _ in range(2)] for c0, c1 in state.cols]
        state.send = [torch.zeros((max(self.n_send, 1), c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in state.cols]
        state.result = torch.empty_like(H0)
        return state

    def _pack(self, state, c, buf, part="all"):
        """Outgoing rows of chunk c from the local part of ``buf``: the rows the peers pull ("pull": a gather), the partial sums
        pushed to them ("push": an SpMM over the entries this rank sums for its peers), or both."""
        if self.n_send:
            self.backend.halo_pack(self.halo, part, buf, state.send[c])

    def _exchange(self, state, c, buf, part="all"):
        """One group of point-to-point transfers: the chosen half (or both halves) of every peer's message."""
        sends, recvs = [], []
        for q in range(self.world):
            if part != "push":
                a, b = self.send_pull_slices[q]
                sends.append((q, state.send[c][a:b]))
                a, b = self.recv_pull_slices[q]
                recvs.append((q, buf[a:b]))
            if part != "pull":
                a, b = self.send_push_slices[q]
                sends.append((q, state.send[c][a:b]))
                a, b = self.recv_push_slices[q]
                recvs.append((q, buf[a:b]))
        self.comm.exchange_pairs(sends, recvs)

    def _compute(self, state, c, src, out, a, interior, skip_empty=False):
        c0, c1 = state.cols[c]
        H0 = state.H0[:, c0:c1]
        if not self.split_rows:
            if not interior:
                self.backend.spmm_mix(self.graph, None, src, H0, 1.0 - a, a, out, skip_empty=skip_empty)
        elif interior:
            self.backend.spmm_mix(self.graph_int, None, src, H0, 1.0 - a, a, out, rows=self.rows_int, skip_empty=skip_empty)
        else:
            self.backend.spmm_mix(self.graph, None, src, H0, 1.0 - a, a, out, rows=self.rows_bnd, skip_empty=skip_empty)

    def propagate(self, state: ShardState, a: float = 0.1, iterations: int = 10, start=None, early_pull=None):
        """H <- H0 (or ``start``: this rank's rows of another initial H, e.g. the input of a GCNII layer whose mix term is
        H0), then K iterations of H <- (1-a) A_hat H + a H0; returns this rank's rows of the result (in the caller's vertex
        order).  ``early_pull`` (default: the graph's setting): the rows the peers pull go on the links as soon as they are
        gathered, while the pushed partial sums are still being summed -- two messages per peer and iteration instead of one."""
        if start is not None and tuple(start.shape) != tuple(state.H0_user.shape if self.row_order is not None else state.H0.shape):
            raise Exception("propagate: start must have the shape of H0")
        if self.world == 1:
            return self._propagate_single_block(state, a, iterations, start)
        first = state.H0 if start is None else start
        if iterations == 0:
            state.result.copy_(first)
            return state.result
        early = self.early_pull if early_pull is None else bool(early_pull)
        early = early and self.n_send_pull_max > 0 and self.n_send_push_max > 0       # (collective decision: every rank takes the same branch)
        lanes = self._lanes
        packed = []

        def pack(c, buf):
            self._pack(state, c, buf, "pull")
            half = lanes.mark() if early else None
            self._pack(state, c, buf, "push")
            return half, lanes.mark()

        for c, (c0, c1) in enumerate(state.cols):
            self.local_view(state.bufs[c][0]).copy_(first[:, c0:c1])
            packed.append(pack(c, state.bufs[c][0]))
        for k in range(iterations):
            last = k == iterations - 1
            # a row without entries is a * H0 after every iteration: it is written the first time each ping-pong buffer is a
            # destination (k = 0, 1) and into the result (last iteration), and left alone in between
            # ... and when nobody references such rows at all (self.empty_rows_unreferenced) only the result ever needs them
            settled = (k >= 2 or self.empty_rows_unreferenced) and k < iterations - 1
            for c, (c0, c1) in enumerate(state.cols):
                src, dst = state.bufs[c][k % 2], state.bufs[c][1 - k % 2]
                with lanes.exchange_lane():                            # runs under the other chunk's SpMM
                    half, whole = packed[c]
                    if early:
                        lanes.wait(half, on_exchange_lane=True)
                        self._exchange(state, c, src, "pull")
                        lanes.wait(whole, on_exchange_lane=True)
                        self._exchange(state, c, src, "push")
                    else:
                        lanes.wait(whole, on_exchange_lane=True)
                        self._exchange(state, c, src, "all")
                    arrived = lanes.mark(on_exchange_lane=True)
                out = state.result[:, c0:c1] if last else self.local_view(dst)
                self._compute(state, c, src, out, a, interior=True, skip_empty=settled)    # needs no halo
                lanes.wait(arrived)
                self._compute(state, c, src, out, a, interior=False, skip_empty=settled)
                if not last:
                    packed[c] = pack(c, dst)
        return state.result

    def _propagate_single_block(self, state, a, iterations, start=None):
        # no exchange; the first iteration reads H0 in place, and on a relabelled shard the last one scatters its
        # rows straight back into the caller's order
        if iterations == 0:
            return start if start is not None else (state.H0_user if self.row_order is not None else state.H0)
        state.cur = 0
        src = state.H0
        if start is not None:
            src = start.to(torch.float32).contiguous() if self.row_order is None else start.to(torch.float32).index_select(0, self.row_order)
        for k in range(iterations):
            last = k == iterations - 1
            if last and self.row_order is not None:
                self.backend.spmm_mix(self.graph, None, src, state.H0, 1.0 - a, a, state.result, out_rows=self.row_order32)
                return state.result
            dst = state.bufs[1 - state.cur]
            self.backend.spmm_mix(self.graph, None, src, state.H0, 1.0 - a, a, dst)
            state.cur = 1 - state.cur
            src = dst
        return src

    # ---- training with edge dropout (layered.py:47-50 + gnn.py:41-42 in every iteration) --------------------------
    # Iteration k of a training forward uses A_k = D_k^-1/2 drop_k(A) D_k^-1/2 with D_k the COLUMN sums of the dropped values.
    # The draw of entry (i, j) is keyed by (seed, stream, global i, global j): the masks -- and with them the logits -- do not
    # depend on the partition.  A block holds the raw values of its rows; the weights are made inside the
    # SpMM (gnx_spmm_dropped).  Column sums and the backward need the halo exchange run BACKWARDS: what a rank computed for
    # the columns of its halo goes back to their owners, who add it up (the transposed send graph does the adding).
    def _need_dropout_block(self):
        if not self.edge_dropout:
            raise Exception("ShardedGraph: build the block with edge_dropout=True for training with edge dropout")

    def _exchange_back(self, full, back):
        """The regions of ``full`` go back to their owners; ``back`` [n_send, C] receives what the peers hold for the rows this
        rank sends them (an edge-dropout block pulls only: every message is one slice)."""
        self.comm.exchange([full[a:b] if b > a else None for a, b in self.recv_slices],
                           [back[a:b] if b > a else None for a, b in self.send_pull_slices])

    def _pull(self, buf, send):
        """Fills the regions of ``buf`` from the peers' local rows (one gather launch + one pairwise exchange)."""
        if self.n_send:
            self.backend.halo_pack(self.halo, "pull", buf, send)
        self.comm.exchange([send[a:b] if b > a else None for a, b in self.send_pull_slices],
                           [buf[a:b] if b > a else None for a, b in self.recv_slices])

    def _add_back(self, full, back, out):
        """out = the local rows of ``full`` + what the peers computed for them (``back``, summed by the transposed send graph:
        a fixed order, whatever the number of peers)."""
        if self.send_graph is not None:
            self.backend.spmm_t_mix(self.send_graph, back[:self.n_send], self.local_view(full), 1.0, 1.0, out)
        else:
            out.copy_(self.local_view(full))

    def dropped_scales(self, p, seed, first_stream, iterations):
        """[iterations, n_buf]: divide_no_nan(1, sqrt(GLOBAL column sums of the dropped values)) (gnn.py:41) of the dropout
        streams first_stream .. first_stream + iterations - 1, for every column of this block's buffer.  Collective."""
        self._need_dropout_block()
        be, K = self.backend, int(iterations)
        part = be.colsum_streams(self.graph, p, seed, first_stream, K)
        if self.world == 1:
            return be.degree_scale(part)
        dev = self.device
        full = part.t().contiguous()                                            # [n_buf, K]: one "feature row" per column
        back = torch.zeros((max(self.n_send, 1), K), dtype=torch.float32, device=dev)
        self._exchange_back(full, back)
        total = torch.empty((self.n_local, K), dtype=torch.float32, device=dev)
        self._add_back(full, back, total)
        be.degree_scale(total)
        self.local_view(full).copy_(total)
        self._pull(full, back)                                                   # the owners' scales into the halo columns
        return full.t().contiguous()

    def _dropout_chunks(self, C, chunks):
        """Column chunks of the edge-dropout path: every column of H propagates alike (the masks and scales belong to the entries),
        so the exchange of one chunk runs on the exchange lane under the SpMM of the next.  Default: two chunks from 64 columns on
        (narrower chunks would pay more for the per-entry weights, which every launch recomputes, than the overlap returns)."""
        if self.world == 1:
            return [(0, C)]
        return split_columns(C, (2 if C >= 64 else 1) if chunks is None else chunks)

    def propagate_dropped(self, H0, a, iterations, p, seed, first_stream, scales, chunks=None):
        """The K training-mode iterations over this rank's rows: H <- (1-a) A_k H + a H0, k = 0 .. K-1."""
        self._need_dropout_block()
        be = self.backend
        H0 = H0.to(torch.float32).contiguous()
        C, dev = H0.shape[1], H0.device
        # every epilogue hands the NEXT iteration its column scale with the row (gnx_spmm_dropped_chained): the scale is a global
        # per-vertex quantity, so the rows a peer pulls arrive pre-scaled as well
        nxt = lambda k: scales[k + 1] if k + 1 < iterations else None
        if self.world == 1:
            H = H0
            for k in range(iterations):
                out = torch.empty_like(H0)
                be.spmm_dropped_chained(self.graph, scales[k], p, seed, first_stream + k, k > 0, nxt(k), H, H0, 1.0 - a, a, out)
                H = out
            return H
        if iterations == 0:
            return H0.clone()
        cols = self._dropout_chunks(C, chunks)
        lanes = self._lanes
        bufs = [[torch.zeros((self.n_buf, c1 - c0), dtype=torch.float32, device=dev) for _ in range(2)] for c0, c1 in cols]
        send = [torch.zeros((max(self.n_send, 1), c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in cols]
        result = torch.empty_like(H0)
        packed = []
        for c, (c0, c1) in enumerate(cols):
            self.local_view(bufs[c][0]).copy_(H0[:, c0:c1])
            if self.n_send:
                be.halo_pack(self.halo, "pull", bufs[c][0], send[c])
            packed.append(lanes.mark())
        for k in range(iterations):
            last = k == iterations - 1
            for c, (c0, c1) in enumerate(cols):
                src, dst = bufs[c][k % 2], bufs[c][1 - k % 2]
                with lanes.exchange_lane():                            # under the other chunk's SpMM
                    lanes.wait(packed[c], on_exchange_lane=True)
                    self.comm.exchange([send[c][a0:a1] if a1 > a0 else None for a0, a1 in self.send_pull_slices],
                                       [src[a0:a1] if a1 > a0 else None for a0, a1 in self.recv_slices])
                    arrived = lanes.mark(on_exchange_lane=True)
                lanes.wait(arrived)
                out = result[:, c0:c1] if last else self.local_view(dst)
                be.spmm_dropped_chained(self.graph, scales[k], p, seed, first_stream + k, k > 0, nxt(k), src, H0[:, c0:c1], 1.0 - a, a, out)
                if not last:
                    if self.n_send:
                        be.halo_pack(self.halo, "pull", dst, send[c])
                    packed[c] = lanes.mark()
        return result

    def propagate_dropped_backward(self, g, a, iterations, p, seed, first_stream, scales, chunks=None):
        """dH0 of propagate_dropped for the output gradient g (this rank's rows): g_k = (1-a) A_k^T g_{k+1},
        dH0 = g_0 + a sum_k g_{k+1}.  A_k^T reaches the halo columns too: those rows go back to their owners -- chunk by chunk,
        the return trip of one column chunk under the transposed SpMM of the next."""
        self._need_dropout_block()
        be = self.backend
        G = g.to(torch.float32).contiguous()
        C, dev = G.shape[1], G.device
        gH0 = torch.zeros_like(G)
        if self.world == 1:
            for k in range(iterations - 1, -1, -1):
                gH0.add_(G, alpha=a)
                nxt = torch.empty_like(G)
                be.spmm_dropped(self.graph, scales[k], p, seed, first_stream + k, True, G, None, 1.0 - a, 0.0, nxt)
                G = nxt
            gH0.add_(G)
            return gH0
        cols = self._dropout_chunks(C, chunks)
        lanes = self._lanes
        full = [torch.empty((self.n_buf, c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in cols]
        back = [torch.zeros((max(self.n_send, 1), c1 - c0), dtype=torch.float32, device=dev) for c0, c1 in cols]
        Gc = [G[:, c0:c1].contiguous() for c0, c1 in cols]
        for k in range(iterations - 1, -1, -1):
            returned = []
            for c, (c0, c1) in enumerate(cols):
                gH0[:, c0:c1].add_(Gc[c], alpha=a)
                be.spmm_dropped(self.graph, scales[k], p, seed, first_stream + k, True, Gc[c], None, 1.0 - a, 0.0, full[c])
                computed = lanes.mark()
                with lanes.exchange_lane():
                    lanes.wait(computed, on_exchange_lane=True)
                    self._exchange_back(full[c], back[c])
                    returned.append(lanes.mark(on_exchange_lane=True))
            for c in range(len(cols)):
                lanes.wait(returned[c])
                nxt = torch.empty_like(Gc[c])
                self._add_back(full[c], back[c], nxt)
                Gc[c] = nxt
        for c, (c0, c1) in enumerate(cols):
            gH0[:, c0:c1].add_(Gc[c])
        return gH0

    # ---- measurements for bench.py ------------------------------------------------------------------------
    def time_exchange(self, state, repeats=3):
        """Seconds of one bare exchange of every chunk's outgoing rows (no compute beside it), max over ranks."""
        if self.world == 1:
            return 0.0
        best = None
        for _ in range(repeats + 1):
            self._sync()
            t0 = time.perf_counter()
            for c in range(len(state.cols)):
                self._exchange(state, c, state.bufs[c][0])
            self._sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)          # the first round opens the connections
        t = torch.tensor([best], dtype=torch.float64, device=self.device)
        self.comm.all_reduce(t, dist.ReduceOp.MAX)
        return float(t.item())

    def time_compute(self, state, a=0.1, repeats=3):
        """Seconds of one iteration's kernels alone (pack + SpMM of every chunk, no exchange), max over ranks."""
        best = None
        for _ in range(repeats + 1):
            self._sync()
            t0 = time.perf_counter()
            if self.world == 1:
                self.backend.spmm_mix(self.graph, None, state.H0, state.H0, 1.0 - a, a, state.bufs[1])
            else:
                for c in range(len(state.cols)):
                    src, dst = state.bufs[c][0], state.bufs[c][1]
                    self._compute(state, c, src, self.local_view(dst), a, interior=True, skip_empty=True)    # a steady-state iteration
                    self._compute(state, c, src, self.local_view(dst), a, interior=False, skip_empty=True)
                    self._pack(state, c, dst)
            self._sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        t = torch.tensor([best], dtype=torch.float64, device=self.device)
        self.comm.all_reduce(t, dist.ReduceOp.MAX)
        return float(t.item())

    def fixed_point_error(self, state, a=0.1, iterations=10):
        """In-run check of the whole vertex-block path (plan, kernels, exchange): for a SYMMETRIC graph H0 = sqrt(column sums) x s
        (a different factor s_c per column) is a fixed point of H <- (1-a) A_hat H + a H0, so after any number of iterations every
        element must still equal H0.  Returns the largest relative deviation over all ranks (rows of isolated vertices must stay 0).
        Overwrites state.H0.  Collective."""
        if self.sqrt_degree is None:
            raise Exception("fixed_point_error: needs a symmetrically normalised graph")
        C = state.H0.shape[1]
        s = 1.0 + torch.arange(C, dtype=torch.float32, device=self.device) / C
        order = self.row_order if self.row_order is not None else None
        E0 = self.sqrt_degree[:, None] * s[None, :]
        if self.world == 1 and order is not None:              # relabelled single block: propagate() expects the caller's order in H0_user
            state.H0_user.copy_(E0)
            state.H0.copy_(E0.index_select(0, order))
        else:
            state.H0.copy_(E0)
        out = self.propagate(state, a, iterations)
        err = torch.tensor([max_relative_deviation(out, E0)], dtype=torch.float64, device=self.device)
        self.comm.all_reduce(err, dist.ReduceOp.MAX)
        return float(err.item())

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.comm.barrier()

    def halo_stats(self):
        """Max over ranks of the plan sizes (rows): what crosses the links per iteration and how the rows split."""
        keys = ["pull_rows", "push_rows", "pull_only_rows", "send_rows", "interior_rows", "boundary_rows", "local_rows", "busiest_link_rows",
                "push_entries"]
        t = torch.tensor([self.stats[k] for k in keys], dtype=torch.int64, device=self.device)
        self.comm.all_reduce(t, dist.ReduceOp.MAX)
        out = {"max_" + k: int(v) for k, v in zip(keys, t.tolist())}
        t = torch.tensor([self.stats["pull_rows"] + self.stats["push_rows"]], dtype=torch.int64, device=self.device)
        self.comm.all_reduce(t, dist.ReduceOp.MAX)
        out["max_halo_rows"] = int(t.item())
        out["cover"], out["split_rows"], out["chunks"] = self.cover, bool(getattr(self, "split_rows", False)), self.chunks
        out["push_weight"] = self.push_weight
        out["early_pull"] = self.early_pull
        return out


# ---- the other parts of the vertex-block path, under their historical names ------------------------------------------------
# Resolved on first use (PEP 562): rmat.py and sharded_layers.py import THIS module, so importing them here at load time would
# make `import gnntf.rmat` fail on a half-initialised gnntf.sharded.
_MOVED = {name: "sharded_layers" for name in ("BlockNodeClassification", "ShardedGCNIILayer", "ShardedGCNLayer", "ShardedPPRLoop",
                                              "SummedGradients")}
_MOVED.update({name: "rmat" for name in ("build_rmat_blocks", "build_rmat_shard", "rmat_block_entries", "rmat_relabelled_pairs",
                                         "rmat_undirected_keys")})


def __getattr__(name):
    if name in _MOVED:
        import importlib
        return getattr(importlib.import_module("." + _MOVED[name], __package__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
