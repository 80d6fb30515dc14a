#!/usr/bin/env python3
"""Does a LOCALITY order of the tail help the narrow-width K loop?  

At C <= 32 a gather moves one 128-byte line for a 16 ... 128-byte row; the K loop already runs C <= 16 on a degree-relabelled
copy (hubs first, 4 - 8 hub rows per line).  That relabelling is a STABLE sort by (clamped) degree, so the order of the vertices
INSIDE a degree bin is whatever the input labelling was -- here a random permutation.  This tool relabels the config-4 graph
before handing it to the library, so that inside every degree bin vertices follow

  workload      the bench's random labelling (baseline)
  degree        the degree bins alone, applied up front (what the library's own relabelling does; the baseline for C > 16)
  hub_grouped   the rank of their most popular neighbour: the leaves of one hub become neighbours in memory, so the hub's row
                gathers them from consecutive lines
  bfs           a breadth-first (Cuthill-McKee style) order from the heaviest vertex, parents in order, children by parent

and times gnx_appnp_propagate (K = 10) at the given widths.  For C > 16 (no internal relabelling) the pre-labelling itself is the
degree order + the tail order.  Run under rocprofv3 --pmc FETCH_SIZE with --only NAME --feats 8 for the fabric bytes.

    python3 tools/narrow_order_experiment.py --feats 8,16,32 [--only hub_grouped]"""
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
from gnntf import _native as nat
from gnntf import ordering, sharded
from gnntf.rmat import community_pairs

INF = 1 << 40


def bfs_positions(src, dst, n, start):
    """Position of every vertex in a breadth-first order from ``start`` (children ordered by their first parent's position);
    vertices the search does not reach keep INF."""
    dev = src.device
    pos = torch.full((n,), INF, dtype=torch.int64, device=dev)
    pos[start] = 0
    lo, hi = 0, 1
    levels = 0
    while True:
        ps = pos[src]
        m = (ps >= lo) & (ps < hi) & (pos[dst] == INF)
        if not bool(m.any()):
            break
        parent = torch.full((n,), INF, dtype=torch.int64, device=dev)
        parent.scatter_reduce_(0, dst[m], ps[m], reduce="amin")
        new = torch.nonzero(parent < INF).reshape(-1)
        new = new[torch.argsort(parent[new], stable=True)]
        pos[new] = hi + torch.arange(new.numel(), device=dev)
        lo, hi = hi, hi + int(new.numel())
        levels += 1
        del ps, m, parent, new
    return pos, levels


def tail_key(name, u, v, n, deg, comm=None):
    """int64 [n]: the secondary sort key inside a degree bin."""
    dev = u.device
    if name in ("workload", "degree"):
        return torch.arange(n, device=dev), {}
    order = torch.argsort(deg, descending=True, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(n, device=dev)
    src, dst = torch.cat([u, v]), torch.cat([v, u])
    if name == "hub_grouped":
        key = torch.full((n,), INF, dtype=torch.int64, device=dev)
        key.scatter_reduce_(0, dst, rank[src], reduce="amin")            # rank of the most popular neighbour
        return key, {}
    if name == "planted":                                                # the generator's own communities: the best any locality order can do
        return comm.clone(), {}
    if name in ("lpa2", "lpa3", "lpa4"):                               # multi-level: the order itself is the key
        order = ordering.locality_order(torch.stack([dst, src], 1), n, levels=int(name[3:]))
        key = torch.empty_like(order)
        key[order] = torch.arange(n, device=dev)
        return key, {}
    if name.startswith("lpa_r"):                                        # label propagation with another number of rounds
        order = ordering.locality_order(torch.stack([dst, src], 1), n, rounds=int(name[5:]))
        key = torch.empty_like(order)
        key[order] = torch.arange(n, device=dev)
        return key, {}
    if name == "lpa":
        label = ordering.propagate_labels(dst, src, n)
        return label, {"lpa_labels": int(torch.unique(label).numel())}
    if name == "bfs":
        pos, levels = bfs_positions(src, dst, n, int(order[0]))
        return pos, {"bfs_levels": levels, "bfs_reached": int((pos < INF).sum())}
    raise SystemExit("unknown order " + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=str, default="8,16,32")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--graph", choices=["rmat", "community"], default="rmat",
                    help="rmat: the bench graph (config 4); community: planted partition x power-law degrees (community_pairs), same N / entries")
    ap.add_argument("--mix", type=float, default=0.2, help="community graph: share of the pairs that leave their community")
    ap.add_argument("--window", type=int, default=0,
                    help="> 0: hand the library the order as a LOCALITY order -- gnx_graph_set_row_window(window): rows taken in windows of this many "
                         "consecutive ids, degree-binned inside a window, no degree-relabelled copy (implies --pure for every order but workload)")
    ap.add_argument("--pure", action="store_true", help="order by the tail key ALONE (no degree bins in front of it) -- for C > 16, where the library does not relabel")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    n, K = a.nodes, 10
    comm = None
    if a.graph == "community":
        u, v, comm = community_pairs(n, a.entries // 2, 1, dev, mix=a.mix)
    else:
        u, v = sharded.rmat_relabelled_pairs(n, a.entries // 2, seed=1, device=dev)
    deg = torch.bincount(u, minlength=n) + torch.bincount(v, minlength=n)
    lib = nat.lib()
    names = [x for x in (a.only.split(",") if a.only else ["workload", "degree", "hub_grouped", "bfs"] + (["planted"] if comm is not None else []))]
    for name in names:
        torch.cuda.synchronize(); t0 = time.time()
        key, info = tail_key(name, u, v, n, deg, comm)
        if name == "workload":
            newid = None
        else:
            bin_ = torch.zeros_like(deg) if (a.pure or a.window > 0) else 512 - deg.clamp(max=512)      # the library's degree bins (heaviest first)
            k1 = torch.argsort(key, stable=True)                          # lexicographic (bin, key, old id) by two stable sorts
            order = k1[torch.argsort(bin_[k1], stable=True)]
            newid = torch.empty_like(order)
            newid[order] = torch.arange(n, device=dev)
            del k1, order, bin_
        torch.cuda.synchronize(); t_order = time.time() - t0
        uu, vv = (u, v) if newid is None else (newid[u], newid[v])
        info["share_of_pairs_within_a_window"] = float(((uu - vv).abs() < max(a.window, 1)).float().mean()) if a.window > 0 else None
        idx = torch.cat([torch.stack([uu, vv], 1), torch.stack([vv, uu], 1)])
        g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=dev), (n, n)), device=dev)
        del idx, uu, vv, key
        if a.window > 0 and name != "workload":
            g.set_row_window(a.window)
        adj = gnntf.normalize(g, "symmetric")
        for C in [int(c) for c in a.feats.split(",")]:
            gen = torch.Generator(device=dev).manual_seed(2)
            H0 = torch.rand(n, C, device=dev, generator=gen) * 2 - 1
            out, work = torch.empty_like(H0), torch.empty_like(H0)
            run = lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), 0.1, K, C, nat.ptr(out), nat.ptr(work),
                                                            nat.current_stream()))
            times = []
            for r in range(a.rounds + 1):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); run(); e.record()
                torch.cuda.synchronize()
                if r:
                    times.append(s.elapsed_time(e))
            ms = sorted(times)[len(times) // 2]
            b = bench.alg_bytes_per_iteration(n, g.nnz, C)
            print(json.dumps(dict(graph=a.graph, entries=g.nnz, empty_rows=int((deg == 0).sum()), max_degree=int(deg.max()), pure=a.pure, window=(a.window if name != "workload" else 0),
                                  order=name, C=C, ms_per_K10=ms, kernel=g.last_kernel(), alg_frac_of_8TBs=b * K / ms / 1e6 / 8000, alg_GB_per_launch=b / 1e9,
                                  order_seconds=round(t_order, 2), checksum=float(out.double().sum()), **info)), flush=True)
            del H0, out, work
        del g, adj
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
