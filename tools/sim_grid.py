#!/usr/bin/env python3
"""One-GPU simulation of the 8-GPU weak-scaling workload (80M vertices / ~800M entries) to size the
process grid Pv (vertex blocks) x Pf (feature slices): per-GPU compute time of each candidate and the halo a
vertex block would need.  python tools/sim_grid.py [--world 8] > gpurun_out/sim_grid.json"""
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
from gnntf.sparse import _launch
from tools.bench_widths import timed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    P, C = a.world, a.feats
    N = a.nodes * P
    t0 = time.time()
    g, adj, prep = bench.build_single(argparse.Namespace(nodes=N, entries=a.entries * P), dev)
    out = {"world": P, "N": N, "entries": g.nnz, "prep": prep, "grids": []}
    rowptr, colidx, _ = g.csr_arrays()
    deg = (rowptr[1:] - rowptr[:-1])
    out["degree"] = {"zero_frac": float((deg == 0).float().mean()), "le2_frac": float((deg <= 2).float().mean()),
                     "max": int(deg.max()), "long_rows": int((deg > 512).sum()), "entries_in_long_rows": int(deg[deg > 512].sum())}
    rows = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32), deg)
    pf = 1
    while pf <= P:
        pv = P // pf
        width = C // pf
        rec = {"pv": pv, "pf": pf, "width": width}
        # halo of vertex block 0 under a pv-way contiguous partition
        hi = N // pv
        mine = rows < hi
        cols0 = colidx[mine].long()
        remote = cols0[cols0 >= hi]
        halo = int(torch.unique(remote).numel()) if remote.numel() else 0
        rec.update(local_entries=int(mine.sum()), halo_rows=halo, halo_bytes=halo * width * 4)
        del mine, cols0, remote
        # compute time: the block's rows at this width (block 0's sub-graph, columns left global)
        sub_nnz = int(rowptr[hi])
        sg = gnntf.DeviceGraph(csr=(rowptr[:hi + 1].clone(), colidx[:sub_nnz].clone(), adj.vals[:sub_nnz].clone(), (hi, N)))
        X = torch.rand(N, width, device=dev)
        H0 = torch.rand(hi, width, device=dev)
        buf = torch.empty(hi, width, device=dev)
        sadj = gnntf.Adjacency(sg)
        ms = timed(lambda: _launch(sadj, X, H0, 0.9, 0.1, 0, out=buf), reps=3, warm=1)
        rec.update(compute_ms=ms, kernel=sg.last_kernel())
        for link_GBs in (50.0, 75.0):          # per-direction per-link estimates
            comm_ms = (halo * width * 4) / max(pv - 1, 1) / (link_GBs * 1e9) * 1e3 if pv > 1 else 0.0
            rec[f"comm_ms_at_{int(link_GBs)}GBs_per_link"] = comm_ms
        out["grids"].append(rec)
        del sg, X, H0, buf, sadj
        torch.cuda.empty_cache()
        pf *= 2
    out["total_s"] = time.time() - t0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
