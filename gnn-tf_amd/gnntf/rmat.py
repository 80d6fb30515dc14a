"""Synthetic workloads of bench.py.  community_pairs: a planted-partition x power-law graph (the structure citation graphs have and
R-MAT lacks), for the locality order.  R-MAT: the generator of SURVEY.md 8(d) ((a, b, c, d) = (0.57, 0.19, 0.19, 0.05), ids folded onto
n vertices, self loops dropped, exactly the requested number of undirected edges, random relabelling) on the device, and the two
ways of turning it into vertex blocks -- ONE global graph cut into pv blocks (strong scaling, the bench's N > 1 workload) and a
graph grown with the world size (weak scaling, kept for tests and rehearsals)."""
from __future__ import annotations

import time

import torch
import torch.distributed as dist

from .sharded import Comm, ShardedGraph, make_grid, uniform_bounds

def _rmat_pairs(scale, m, gen, device, a=0.57, b=0.19, c=0.19):
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, device=device, generator=gen)
        src = src * 2 + (r >= a + b).long()
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()
    return src, dst


def rmat_undirected_keys(n, m_undirected, seed, device):
    """Exactly m_undirected distinct undirected edges {u < v} of an R-MAT graph (a, b, c, d) = (0.57, 0.19, 0.19,
    0.05) folded onto n vertices (ids modulo n, self loops dropped), as int64 keys u * n + v."""
    gen = torch.Generator(device=device).manual_seed(seed)
    scale = max(1, (n - 1).bit_length())
    keys = torch.empty(0, dtype=torch.int64, device=device)
    while keys.numel() < m_undirected:
        need = m_undirected - keys.numel()
        s, d = _rmat_pairs(scale, int(need * 1.25) + 1024, gen, device)
        s, d = s % n, d % n
        keep = s != d
        s, d = s[keep], d[keep]
        lo, hi = torch.minimum(s, d), torch.maximum(s, d)
        del s, d, keep
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del lo, hi
    if keys.numel() > m_undirected:
        pick = torch.randperm(keys.numel(), device=device, generator=gen)[:m_undirected]
        keys = keys[pick]
    return keys


def rmat_relabelled_pairs(n, m_undirected, seed, device, perm_seed=3):
    """The bench graph: rmat_undirected_keys + a random vertex relabelling (so locality is not an artefact of the
    generator).  Returns (u, v) int64 [m_undirected] each; the stored matrix is the symmetrised pattern."""
    keys = rmat_undirected_keys(n, m_undirected, seed, device)
    gen = torch.Generator(device=device).manual_seed(perm_seed)
    perm = torch.randperm(n, device=device, generator=gen)
    u, v = perm[torch.div(keys, n, rounding_mode="floor")], perm[keys % n]
    return u, v


def community_pairs(n, m, seed, device, mix=0.2, gamma=2.5, tau=2.0, size_lo=64, size_hi=65536, max_weight=30000.0):
    """A community-structured power-law graph (BTER / LFR-like, what citation and co-purchase graphs such as Cora or ogbn-arxiv look
    like and R-MAT does not): planted communities with power-law SIZES (exponent ``tau``, ``size_lo`` ... ``size_hi`` vertices) x
    power-law expected DEGREES (Pareto weights, exponent ``gamma``).  Each of the ``m`` undirected pairs picks its source in
    proportion to the weights and its target, with probability 1 - ``mix``, inside the source's community (again in proportion
    to the weights), else anywhere.  Vertices are then relabelled by a random permutation, like the bench graph.
    Returns (u, v, community of every vertex under the FINAL labels)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    draws = max(1024, int(4 * n / size_lo))
    sizes = (size_lo * (1.0 - torch.rand(draws, device=device, generator=gen, dtype=torch.float64)).pow(-1.0 / (tau - 1.0))).clamp(max=size_hi).long()
    ends = torch.cumsum(sizes, 0)
    ends = ends[: int(torch.searchsorted(ends, torch.tensor([n], device=device))[0]) + 1].clamp(max=n)          # community c = [ends[c-1], ends[c])
    comm = torch.bucketize(torch.arange(n, device=device), ends, right=True)
    weight = (1.0 - torch.rand(n, device=device, generator=gen, dtype=torch.float64)).pow(-1.0 / (gamma - 1.0)).clamp(max=max_weight)
    cdf = torch.cumsum(weight, 0)
    total = float(cdf[-1])
    u = torch.searchsorted(cdf, torch.rand(m, device=device, generator=gen, dtype=torch.float64) * total).clamp(max=n - 1)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), ends[:-1]])
    c_lo = torch.where(starts > 0, cdf[(starts - 1).clamp(min=0)], torch.zeros_like(cdf[:1]))                    # weight mass before each community
    c_hi = cdf[ends - 1]
    cu = comm[u]
    inside = torch.rand(m, device=device, generator=gen) >= mix
    lo = torch.where(inside, c_lo[cu], torch.zeros_like(c_lo[:1]))
    hi = torch.where(inside, c_hi[cu], torch.full_like(c_hi[:1], total))
    v = torch.searchsorted(cdf, lo + torch.rand(m, device=device, generator=gen, dtype=torch.float64) * (hi - lo)).clamp(max=n - 1)
    del cdf, weight, lo, hi, inside, cu
    keep = u != v
    u, v = u[keep], v[keep]
    perm = torch.randperm(n, device=device, generator=gen)
    comm_final = torch.empty_like(comm)
    comm_final[perm] = comm
    return perm[u], perm[v], comm_final


def rmat_block_entries(n_global, entries_global, seed, device, group=None, grid=None, replicate=False):
    """This rank's entries of the STRONG-scaling workload: ONE global R-MAT graph (n_global vertices, entries_global stored
    entries -- the same graph for every world size, and the graph bench.py's one-GPU run builds) cut into pv contiguous vertex
    blocks.  Rank 0 generates the edge list and broadcasts it; every rank keeps the entries of its rows.
    ``replicate``: every rank generates the list itself instead (same seed, same device type: the same list) -- for rehearsals on
    transports where an 8 GB broadcast is the slow part (gloo, staged through the host); the ranks then compare a checksum of
    their lists and fail loudly if they differ.
    Returns (idx int64 [m, 2] global (row, col), vals, bounds, comm of the vertex partition, (v, f, pv, pf), seconds)."""
    world_comm = Comm(group=group) if dist.is_initialized() else Comm(solo=True)
    rank, world = world_comm.rank, world_comm.size
    pv, pf = grid if grid is not None else (world, 1)
    v, f, comm = make_grid(world, rank, pv, pf)
    t0 = time.time()
    m = entries_global // 2
    if replicate and world > 1:
        for turn in range(world):                          # one rank at a time: the ranks of a rehearsal may share one card
            if turn == rank:
                u, w = rmat_relabelled_pairs(n_global, m, seed, device)
                pairs = torch.stack([u, w])
                del u, w
            world_comm.barrier()
        mark = (pairs[0] * 31 + pairs[1]).sum().reshape(1).to(torch.float64)       # (wraps; equal lists give equal sums)
        lo_mark, hi_mark = mark.clone(), mark.clone()
        world_comm.all_reduce(lo_mark, dist.ReduceOp.MIN)
        world_comm.all_reduce(hi_mark, dist.ReduceOp.MAX)
        if float(lo_mark) != float(hi_mark):
            raise Exception("rmat_block_entries: the ranks generated different edge lists")
    else:
        if rank == 0:
            u, w = rmat_relabelled_pairs(n_global, m, seed, device)
            pairs = torch.stack([u, w])
            del u, w
        else:
            pairs = torch.empty((2, m), dtype=torch.int64, device=device)
        world_comm.broadcast(pairs, 0)
    u, w = pairs[0], pairs[1]
    bounds = uniform_bounds(n_global, pv)
    lo, hi = bounds[v], bounds[v + 1]
    mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
    idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
    del u, w, mu, mw, pairs
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return idx, vals, bounds, comm, (v, f, pv, pf), time.time() - t0


def build_rmat_blocks(n_global, entries_global, seed, device, backend=None, group=None, grid=None, replicate=False, **graph_options):
    """rmat_block_entries + the ShardedGraph over them.  Returns (ShardedGraph, info, (v, f, pv, pf))."""
    idx, vals, bounds, comm, where, t_gen = rmat_block_entries(n_global, entries_global, seed, device, group=group, grid=grid,
                                                               replicate=replicate)
    t0 = time.time()
    sg = ShardedGraph(idx, vals, bounds, backend=backend, comm=comm, **graph_options)
    del idx, vals
    if device.type == "cuda":
        torch.cuda.synchronize(device)
        torch.cuda.empty_cache()
    return sg, dict(gen_s=round(t_gen, 2), prep_s=round(time.time() - t0, 2)), where


def build_rmat_shard(nodes_per_rank, entries_per_rank, seed, device, backend=None, group=None, grid=None, relabel=True,
                     **graph_options):
    """WEAK-scaling variant (kept for rehearsals and tests): a global R-MAT graph with nodes_per_rank * P vertices
    and about entries_per_rank * P stored entries on a pv x pf grid.  Every rank draws its share of undirected
    edges; the shares are exchanged so that every rank sees the same global edge set, the same global vertex
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
    keys = torch.unique(torch.cat(world_comm.alltoallv([keys] * world)))     # de-duplicate across shares
    pgen = torch.Generator(device=device).manual_seed(3)
    perm = torch.randperm(N, device=device, generator=pgen)
    world_comm.broadcast(perm, 0)                              # the SAME permutation everywhere: rank 0's
    bounds = uniform_bounds(N, pv)
    lo, hi = bounds[v], bounds[v + 1]
    u, w = perm[torch.div(keys, N, rounding_mode="floor")], perm[keys % N]
    del keys, perm
    mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
    idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
    del u, w, mu, mw
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t_gen = time.time() - t0
    t0 = time.time()
    sg = ShardedGraph(idx, vals, bounds, backend=backend, comm=comm, relabel=relabel, **graph_options)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return sg, dict(gen_s=round(t_gen, 2), prep_s=round(time.time() - t0, 2)), (v, f, pv, pf)
