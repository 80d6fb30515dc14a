#!/usr/bin/env python3
"""Would a HOT / COLD split of the gathers pay?  (round 4, after the counters showed the row kernels waiting on memory 91 % of
their wave time at the chip's gather rate.)

The gathers of a power-law graph go mostly to a few hub rows of H; the 256 MiB Infinity Cache could hold the hottest 128 MB of
them, but the cold gathers interleaved with them flush it (a line survives only while less than about 256 MB of other traffic
passes between two uses).  Idea: relabel the vertices by degree, so that the hot rows of H are one contiguous prefix and the hot
entries of every CSR row are a prefix of that row, and run one iteration as TWO launches of the existing kernel --

    tmp = (1 - a) A_hot H + a H0          (only columns < T: every gather goes to the 128 MB hot prefix)
    out = (1 - a) A_cold H + tmp          (the rest)

-- paying a write + read of tmp (2 N C 4 bytes) for gathers that are served on-die.  This tool times exactly that with the
library as it is (two DeviceGraphs from the filtered COO) against the single launch on the same relabelled graph, and a graph
whose H fits the cache entirely (what an on-die gather costs at best).

    python3 tools/hot_cold_experiment.py --nodes 10000000 --entries 100000000 --feats 128,256 --hot-mb 64,128"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import gnntf
from gnntf import sharded
from gnntf.sparse import _launch


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e))
    return sorted(ms)[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=str, default="128,256")
    ap.add_argument("--hot-mb", type=str, default="64,128")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    n = a.nodes
    u, v = sharded.rmat_relabelled_pairs(n, a.entries // 2, seed=1, device=dev)
    deg = torch.bincount(u, minlength=n) + torch.bincount(v, minlength=n)
    order = torch.argsort(deg, descending=True, stable=True)
    newid = torch.empty_like(order)
    newid[order] = torch.arange(n, device=dev)
    u, v = newid[u], newid[v]                                           # degree-relabelled: vertex 0 is the heaviest
    del order, newid
    rows, cols = torch.cat([u, v]), torch.cat([v, u])
    del u, v
    D = deg.float().sort(descending=True).values.clamp_min(1.0).rsqrt()       # symmetric normalisation of the unit-weight graph
    vals = D[rows] * D[cols]
    full = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows, cols], 1), vals, (n, n)), device=dev)
    adj_full = gnntf.Adjacency(full, None)
    for C in [int(c) for c in a.feats.split(",")]:
        H = torch.rand(n, C, device=dev) * 2 - 1
        H0 = torch.rand(n, C, device=dev) * 2 - 1
        out = torch.empty_like(H)
        t_single = timed(lambda: _launch(adj_full, H, H0, 0.9, 0.1, 0, out=out))
        want = out.clone()
        line = {"nodes": n, "entries": int(rows.numel()), "C": C, "single_launch_ms": t_single, "kernel": full.last_kernel()}
        for mb in [int(x) for x in a.hot_mb.split(",")]:
            T = min(n, (mb << 20) // (C * 4))
            hot = cols < T
            g_hot = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows[hot], cols[hot]], 1), vals[hot], (n, n)), device=dev)
            g_cold = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows[~hot], cols[~hot]], 1), vals[~hot], (n, n)), device=dev)
            a_hot, a_cold = gnntf.Adjacency(g_hot, None), gnntf.Adjacency(g_cold, None)
            tmp = torch.empty_like(H)

            def two():
                _launch(a_hot, H, H0, 0.9, 0.1, 0, out=tmp)
                _launch(a_cold, H, tmp, 0.9, 1.0, 0, out=out)
            t_two = timed(two)
            t_hot = timed(lambda: _launch(a_hot, H, H0, 0.9, 0.1, 0, out=tmp))
            t_cold = timed(lambda: _launch(a_cold, H, tmp, 0.9, 1.0, 0, out=out))
            two()
            err = float(((out - want).abs() / want.abs().clamp_min(1.0)).max())
            share = float(hot.float().mean())
            line[f"hot_{mb}MB"] = {"hot_columns": T, "hot_entry_share": share, "two_launches_ms": t_two, "hot_ms": t_hot, "cold_ms": t_cold,
                                   "hot_G_entries_per_s": share * rows.numel() / t_hot / 1e6, "cold_G_entries_per_s": (1 - share) * rows.numel() / t_cold / 1e6,
                                   "single_G_entries_per_s": rows.numel() / t_single / 1e6, "max_rel_diff": err}
            del g_hot, g_cold, a_hot, a_cold, tmp, hot
            torch.cuda.empty_cache()
        print(json.dumps(line), flush=True)
        del H, H0, out, want


if __name__ == "__main__":
    main()
