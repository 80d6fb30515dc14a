"""Multi-GPU propagation (one process per GPU): a pv x pf grid of vertex blocks x feature slices.

Feature columns propagate independently (the step acts on every column of H alike), so slicing the
columns over ranks needs no exchange at all; vertex blocks need the halo exchange described below.
choose_grid() picks the grid with a small measured cost model (grid_cost_ms).

The reference has no distributed code at all; this module is the
multi-GPU form of the same hot path -- PPRIteration.__forward__ (reference
gnntf/core/gnn/architectures/filter.py:17-22) over a normalised adjacency
(gnntf/core/gnn/gnn.py:36-50) -- and must give the same logits as one GPU.

Layout per rank r owning the global rows [lo, hi):
  * the rows of A_hat with all their entries (a local CSR);
  * a feature buffer  X = [ halo_low | local rows | halo_high ]  where halo_low / halo_high
    are the remote rows this rank's entries reference with global id < lo / >= hi, each sorted
    by global id.  The column remap global -> buffer position is therefore MONOTONIC, so the
    order of a row's entries -- and with it the floating-point summation order of the kernel --
    is the same as on one GPU;
  * per peer q a send list (which of my rows q needs).
Each iteration: pack the send rows (gnx_gather_rows) -> pairwise isend/irecv of the halo rows
(RCCL group of point-to-point transfers: every xGMI link carries its own peer's rows; no ring)
-> fused SpMM+mix over [X] writing the local part of the other ping-pong buffer.

Column sums for the normalisation need one all-reduce of an N-vector at build time.
The heavy lifting goes through a small backend object; the product backend is libgnx.so
(NativeBackend).  Tests on CPU ranks (gloo) supply their own checker backend: this module
never imports a CPU implementation.
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist

from . import _native as nat
from . import sparse


class NativeBackend:
    """libgnx.so through ctypes; every tensor lives on this rank's GPU."""

    def graph_from_coo(self, idx, vals, shape):
        return sparse.DeviceGraph(sparse.SparseCOO(idx, vals, shape), device=idx.device)

    def graph_from_csr(self, rowptr, colidx, vals, shape):
        return sparse.DeviceGraph(csr=(rowptr, colidx, vals, shape))

    def csr_arrays(self, graph):
        return graph.csr_arrays()

    def colsum(self, graph):
        out = torch.empty(graph.n_cols, dtype=torch.float32, device=graph.device)
        with torch.cuda.device(graph.device):
            nat.check(nat.lib().gnx_graph_colsum(graph.handle, 0.0, 0, 0, nat.ptr(out), nat.current_stream()))
        return out

    def degree_scale(self, deg, normalized="symmetric"):
        with torch.cuda.device(deg.device):
            nat.check(nat.lib().gnx_degree_scale(nat.ptr(deg), deg.numel(), nat.NORM[normalized], 0, nat.current_stream()))
        return deg

    def scale_values(self, graph, row_scale, col_scale):
        out = torch.empty(graph.nnz, dtype=torch.float32, device=graph.device)
        row_scale = row_scale.contiguous() if row_scale is not None else None
        with torch.cuda.device(graph.device):
            nat.check(nat.lib().gnx_graph_scale_values(graph.handle, 0.0, 0, 0, nat.ptr(row_scale), nat.ptr(col_scale),
                                                       nat.ptr(out), nat.current_stream()))
        return out

    def spmm_mix(self, graph, vals, X, H0, beta, alpha, out, out_rows=None):
        sparse._launch(sparse.Adjacency(graph, vals), X, H0, beta, alpha, nat.ACT_NONE, out=out, out_rows=out_rows)

    def gather_rows(self, X, idx):
        return sparse.gather_rows(X, idx)


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

    def all_gather_vec(self, t):
        if self.size == 1:
            return [t]
        src = t.cpu() if self._staged(t) else t
        table = [torch.zeros_like(src) for _ in range(self.size)]
        dist.all_gather(table, src, group=self.group)
        return [x.to(t.device) for x in table]

    def exchange(self, send_chunks, recv_chunks):
        """Pairwise exchange: send_chunks[q] goes to group rank q, recv_chunks[q] is filled from it.
        One batch of point-to-point operations (NCCL/RCCL: a single group call, every peer pair on
        its own xGMI link)."""
        if self.size == 1:
            return
        if any(self._staged(t) for t in list(send_chunks) + list(recv_chunks)):
            host_recv = [None if t is None else torch.empty(t.shape, dtype=t.dtype) for t in recv_chunks]
            self.exchange([None if t is None else t.cpu() for t in send_chunks], host_recv)
            for q, (d, h) in enumerate(zip(recv_chunks, host_recv)):
                if q != self.rank and d is not None and d.numel() > 0:
                    d.copy_(h)
            return
        peer = (lambda q: q) if self.group is None else (lambda q: dist.get_global_rank(self.group, q))
        ops = []
        for q, t in enumerate(recv_chunks):
            if q != self.rank and t is not None and t.numel() > 0:
                ops.append(dist.P2POp(dist.irecv, t, peer(q), self.group))
        for q, t in enumerate(send_chunks):
            if q != self.rank and t is not None and t.numel() > 0:
                ops.append(dist.P2POp(dist.isend, t, peer(q), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


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


def grid_cost_ms(pv, pf, feats, nodes_total, entries_total, link_GBs=60.0, halo_frac=0.16):
    """Estimated time of ONE propagation iteration on a pv x pf grid, from round-1 measurements on MI355X:
      * compute: entries per rank x (5.5 + 0.61 * max(w, 32)) ps, w = columns per rank -- the fused kernel's
        measured cost (RMAT 10M/100M: 2.5 / 4.5 / 8.6 / 16.3 ms at w = 32 / 64 / 128 / 256; below 32 columns a
        gather still moves one 128-byte line, so narrower slices are not cheaper);
      * exchange (pv > 1 only): halo_frac * nodes_total rows of 4w bytes arrive per rank over its pv - 1 links
        (one xGMI link per peer, link_GBs per direction).  halo_frac = 0.16 is what a random 1-D partition of
        the RMAT workload needs (tools/sim_grid.py: 12.4-14.5M rows of 80M); it is graph dependent."""
    w = max(feats // pf, 1)
    compute = entries_total / pv * (5.5 + 0.61 * max(w, 32)) * 1e-9
    comm = 0.0 if pv == 1 else halo_frac * nodes_total * 4.0 * w / ((pv - 1) * link_GBs * 1e9) * 1e3
    return compute + comm


def choose_grid(world, feats, nodes_total=80_000_000, entries_total=800_000_000, **model):
    """Feature columns propagate independently (filter.py:19-21 acts on every column of H alike), so slicing
    them over ranks needs NO exchange, while vertex blocks pay a halo exchange per iteration but keep rows
    wide.  Picks the pv x pf factorisation of ``world`` (pf dividing ``feats``) with the lowest grid_cost_ms."""
    best = None
    for pf in range(1, world + 1):
        if world % pf or feats % pf:
            continue
        pv = world // pf
        cost = grid_cost_ms(pv, pf, feats, nodes_total, entries_total, **model)
        if best is None or cost < best[0] - 1e-12:
            best = (cost, pv, pf)
    return best[1], best[2]


def uniform_bounds(n_global, world):
    return [r * n_global // world for r in range(world + 1)]


class ShardState:
    """Ping-pong feature buffers of one propagation."""

    def __init__(self, bufs, H0):
        self.bufs, self.H0, self.cur = bufs, H0, 0


class ShardedGraph:
    """This rank's shard of a symmetrically normalised, vertex-partitioned square graph."""

    def __init__(self, idx_global, vals, bounds, backend=None, group=None, normalized="symmetric", comm=None,
                 relabel=False):
        """``idx_global``: int64 [nnz, 2] (global row, global col) of the entries whose row this
        rank owns (unsorted, duplicates allowed); ``bounds``: the P+1 partition boundaries.
        Collective: every rank of the vertex partition (``comm`` / ``group``) must call it.
        ``relabel`` (single vertex block only): store the shard with its vertices relabelled in stable order of
        descending entry count -- a legal preprocessing step that makes the sub-wave kernels
        5-20 % faster (the rows a wave shares, their H0/out rows and the hub rows become neighbours in memory);
        propagate() permutes H0 on the way in and the result on the way out, so callers never see the new ids."""
        self.backend = backend if backend is not None else NativeBackend()
        self.comm = comm if comm is not None else Comm(group=group)
        self.group = self.comm.group
        self.rank, self.world = self.comm.rank, self.comm.size
        be = self.backend
        dev = idx_global.device
        self.device = dev
        self.bounds = [int(b) for b in bounds]
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        N = self.bounds[-1]
        self.lo, self.hi, self.n_global, self.n_local = lo, hi, N, hi - lo
        if idx_global.numel() and (int(idx_global[:, 0].min()) < lo or int(idx_global[:, 0].max()) >= hi):
            raise Exception("ShardedGraph: an entry's row is outside this rank's range [%d, %d)" % (lo, hi))

        # local rows x global columns: coalesce, then normalise with GLOBAL column sums (gnn.py:41-42)
        local_idx = idx_global.clone()
        local_idx[:, 0] -= lo
        g0 = be.graph_from_coo(local_idx, vals, (self.n_local, N))
        rowptr, colidx, raw = be.csr_arrays(g0)
        if normalized == "symmetric":
            deg = be.colsum(g0)
            self.comm.all_reduce(deg)
            D = be.degree_scale(deg, "symmetric")
            nvals = be.scale_values(g0, D[lo:hi], D)
        elif normalized == "none":
            nvals = raw
        else:
            raise Exception("Invalid matrix normalization")
        self.row_order = None
        if relabel and self.world == 1 and colidx.numel() > 0:
            deg = rowptr[1:] - rowptr[:-1]
            order = torch.argsort(deg, descending=True, stable=True)                 # new id -> old id
            newid = torch.empty_like(order)
            newid[order] = torch.arange(order.numel(), device=dev)
            rows = torch.repeat_interleave(torch.arange(self.n_local, device=dev), deg)
            g0 = be.graph_from_coo(torch.stack([newid[rows], newid[colidx.to(torch.int64)]], dim=1), nvals, (self.n_local, N))
            rowptr, colidx, nvals = be.csr_arrays(g0)
            self.row_order, self.row_newid, self.row_order32 = order, newid, order.to(torch.int32)
            del rows, deg
        self.nnz_local = int(colidx.numel())
        t = torch.tensor([self.nnz_local], dtype=torch.int64, device=dev)
        self.comm.all_reduce(t)
        self.nnz_global = int(t.item())

        # halo plan: distinct remote columns, sorted by global id (=> grouped by owner)
        col = colidx.to(torch.int64)
        remote = (col < lo) | (col >= hi)
        halo = torch.unique(col[remote])
        self.halo_ids = halo                     # global ids of the halo rows: [low part | high part], ascending
        n_low = int((halo < lo).sum())
        self.n_low, self.n_high = n_low, int(halo.numel()) - n_low
        pos = torch.searchsorted(halo, col)
        new_col = torch.where(col < lo, pos, torch.where(col >= hi, pos + self.n_local, col - lo + n_low))
        self.n_buf = self.n_low + self.n_local + self.n_high
        self.graph = be.graph_from_csr(rowptr, new_col.to(torch.int32), nvals, (self.n_local, self.n_buf))
        self.vals = None   # values live in the handle (raw values of the remapped CSR are already normalised)
        del g0

        # who owns which halo row; what every peer needs from me
        bnd = torch.tensor(self.bounds[1:], dtype=torch.int64, device=dev)
        owner = torch.bucketize(halo, bnd, right=True)
        recv_counts = torch.bincount(owner, minlength=self.world).to(torch.int64)
        table = self.comm.all_gather_vec(recv_counts)
        self.recv_counts = [int(c) for c in recv_counts.tolist()]
        self.send_counts = [int(table[q][self.rank]) for q in range(self.world)]
        roff = [0]
        for c in self.recv_counts:
            roff.append(roff[-1] + c)
        want = [halo[roff[q]:roff[q + 1]].contiguous() for q in range(self.world)]      # ids I ask of q
        asked = [torch.empty(self.send_counts[q], dtype=torch.int64, device=dev) for q in range(self.world)]
        self.comm.exchange(want, asked)
        self.send_idx = (torch.cat(asked) - lo) if sum(self.send_counts) else torch.empty(0, dtype=torch.int64, device=dev)
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_local):
            raise Exception("ShardedGraph: a peer asked for a row this rank does not own")
        # where each peer's rows land in the buffer: low part for q < rank, high part for q > rank
        self.recv_slices = []
        for q in range(self.world):
            start = roff[q] if q < self.rank else roff[q] + self.n_local
            self.recv_slices.append((start, start + self.recv_counts[q]))
        soff = [0]
        for c in self.send_counts:
            soff.append(soff[-1] + c)
        self.send_slices = [(soff[q], soff[q + 1]) for q in range(self.world)]

    # ---- propagation -----------------------------------------------------------------------------
    def local_view(self, buf):
        return buf[self.n_low:self.n_low + self.n_local]

    def make_state(self, H0):
        """Allocates the two [halo_low | local | halo_high] buffers for features of H0's width."""
        H0 = H0.to(torch.float32).contiguous()
        if H0.shape[0] != self.n_local:
            raise Exception("make_state: H0 must hold this rank's %d rows" % self.n_local)
        bufs = [torch.zeros((self.n_buf, H0.shape[1]), dtype=torch.float32, device=H0.device) for _ in range(2)]
        state = ShardState(bufs, H0)
        if self.row_order is not None:                                 # relabelled shard: H0 in the new order, result buffer in the old
            state.H0_user, state.H0 = H0, H0.index_select(0, self.row_order)
            state.result = torch.empty_like(H0)
        return state

    def exchange_halo(self, buf):
        """Fills the halo rows of ``buf`` with the owners' current local rows."""
        local = self.local_view(buf)
        packed = self.backend.gather_rows(local, self.send_idx) if self.send_idx.numel() else None
        sends = [packed[a:b] if packed is not None and b > a else None for a, b in self.send_slices]
        recvs = [buf[a:b] if b > a else None for a, b in self.recv_slices]
        self.comm.exchange(sends, recvs)

    def step(self, state: ShardState, a: float):
        """One PPRIteration over the shard: halo exchange + fused SpMM/mix."""
        cur, nxt = state.bufs[state.cur], state.bufs[1 - state.cur]
        self.exchange_halo(cur)
        self.backend.spmm_mix(self.graph, None, cur, state.H0, 1.0 - a, a, self.local_view(nxt))
        state.cur = 1 - state.cur

    def propagate(self, state: ShardState, a: float = 0.1, iterations: int = 10):
        """H <- H0, then K iterations; returns this rank's rows of the result (in the caller's vertex order)."""
        state.cur = 0
        if self.n_buf != self.n_local or iterations == 0:              # halo present: the buffers carry [halo | local | halo]
            self.local_view(state.bufs[0]).copy_(state.H0)
            for _ in range(iterations):
                self.step(state, a)
            return self.local_view(state.bufs[state.cur])
        # a single vertex block: no exchange; the first iteration reads H0 in place, and on a relabelled shard
        # the last one scatters its rows straight back into the caller's order
        src = state.H0
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

    def halo_stats(self):
        t = torch.tensor([self.n_low + self.n_high, int(self.send_idx.numel()), self.n_local], dtype=torch.int64, device=self.device)
        self.comm.all_reduce(t, dist.ReduceOp.MAX)
        return {"max_halo_rows": int(t[0]), "max_send_rows": int(t[1]), "max_local_rows": int(t[2])}


# ---- synthetic sharded R-MAT (bench.py, N > 1) ------------------------------------------------------------
def _rmat_pairs(scale, m, gen, device, a=0.57, b=0.19, c=0.19):
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, device=device, generator=gen)
        src = src * 2 + (r >= a + b).long()
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()
    return src, dst


def build_rmat_shard(nodes_per_rank, entries_per_rank, seed, device, backend=None, group=None, grid=None, relabel=True):
    """Weak-scaling workload: a global R-MAT graph with nodes_per_rank * P vertices and about
    entries_per_rank * P stored (symmetrised, de-duplicated) entries, on a pv x pf process grid
    (default: pv = P vertex blocks, one feature slice).  Every rank draws its share of undirected edges;
    the shares are all-gathered so that every rank sees the same global edge set, the same global vertex
    permutation is applied, and each rank keeps the rows of its vertex block.
    Returns (ShardedGraph, info, (v, f, pv, pf))."""
    world_comm = Comm(group=group) if dist.is_initialized() else Comm(solo=True)
    rank, world = world_comm.rank, world_comm.size
    pv, pf = grid if grid is not None else (world, 1)
    v, f, comm = make_grid(world, rank, pv, pf)
    N = nodes_per_rank * world
    t0 = time.time()
    gen = torch.Generator(device=device).manual_seed(seed * 1000003 + rank)
    scale = max(1, (N - 1).bit_length())
    m = entries_per_rank // 2
    s, d = _rmat_pairs(scale, int(m * 1.012), gen, device)     # ~1 % are lost to self loops / duplicates
    s, d = s % N, d % N
    keep = s != d
    s, d = s[keep], d[keep]
    keys = torch.unique(torch.minimum(s, d) * N + torch.maximum(s, d))
    del s, d, keep
    # every rank receives every share (identical global edge set on all ranks)
    counts = world_comm.all_gather_vec(torch.tensor([keys.numel()], dtype=torch.int64, device=device))
    recvs = [keys if q == rank else torch.empty(int(counts[q]), dtype=torch.int64, device=device) for q in range(world)]
    world_comm.exchange([keys] * world, recvs)
    keys = torch.unique(torch.cat(recvs))                      # de-duplicate across shares
    del recvs
    pgen = torch.Generator(device=device).manual_seed(3)
    perm = torch.randperm(N, device=device, generator=pgen)
    if world > 1:                                              # the SAME permutation everywhere: rank 0's
        if world_comm._staged(perm):
            h = perm.cpu(); dist.broadcast(h, 0, group=group); perm.copy_(h)
        else:
            dist.broadcast(perm, 0, group=group)
    bounds = uniform_bounds(N, pv)
    lo, hi = bounds[v], bounds[v + 1]
    u, w = perm[keys // N], perm[keys % N]
    del keys, perm
    mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
    idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
    del u, w, mu, mw
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t_gen = time.time() - t0
    t0 = time.time()
    sg = ShardedGraph(idx, vals, bounds, backend=backend, comm=comm, relabel=relabel)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return sg, dict(gen_s=round(t_gen, 2), prep_s=round(time.time() - t0, 2)), (v, f, pv, pf)
