#!/usr/bin/env python3
"""Share of rows / entries that the long-row kernels handle (rows with more than LONG_ROW = 512 entries), config 4 and 5."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench
dev = torch.device("cuda:0")
out = {}
for name, n, e in (("config4", 10_000_000, 100_000_000), ("config5", 80_000_000, 1_000_000_000)):
    g, adj, _ = bench.build_single(argparse.Namespace(nodes=n, entries=e), dev)
    rowptr = g.csr_arrays()[0]
    deg = rowptr[1:] - rowptr[:-1]
    rec = {"rows": n, "entries": int(deg.sum()), "empty_rows": int((deg == 0).sum()), "max_row": int(deg.max())}
    for thr in (64, 128, 256, 512, 2048, 8192):
        m = deg > thr
        rec[f"rows_gt_{thr}"] = int(m.sum()); rec[f"entries_in_rows_gt_{thr}"] = int(deg[m].sum())
    out[name] = rec
    del g, adj, rowptr, deg
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
