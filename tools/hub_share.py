#!/usr/bin/env python3
"""Share of gathered rows that fall on the K highest-degree vertices of the bench graph."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
dev = torch.device("cuda:0")
a = argparse.Namespace(nodes=10_000_000, entries=100_000_000)
g, adj, _ = bench.build_single(a, dev)
rowptr, colidx, _ = g.csr_arrays()
deg = (rowptr[1:] - rowptr[:-1])
short = deg <= 512
cnt = torch.bincount(colidx.long(), minlength=g.n_rows)                     # references per column, all rows
rows = torch.repeat_interleave(torch.arange(g.n_rows, device=dev), deg)
cnt_short = torch.bincount(colidx.long()[short[rows]], minlength=g.n_rows)  # references from the row-per-wave kernel only
out = {}
for name, c in (("all_rows", cnt), ("short_rows_only", cnt_short)):
    s, _ = torch.sort(c, descending=True)
    cs = torch.cumsum(s, 0).double() / s.sum().double()
    out[name] = {str(k): round(float(cs[k - 1]), 4) for k in (128, 1024, 4096, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152)}
    out[name]["max_refs"] = int(s[0])
print(json.dumps(out, indent=1))
