#!/usr/bin/env python3
"""How the row-per-wave kernel behaves on short rows: random d-regular-ish graphs (every row exactly d
random neighbours), N=10M, C=256.  Prints algorithmic TB/s per degree."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
from gnntf.sparse import _launch
from tools.bench_widths import timed

n, C = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
H = torch.rand(n, C, device=dev); H0 = torch.rand(n, C, device=dev); out = torch.empty_like(H)
for d in (1, 2, 3, 4, 6, 8, 12, 16, 32):
    rows = torch.arange(n, device=dev).repeat_interleave(d)
    cols = torch.randint(0, n, (n * d,), device=dev)
    g = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows, cols], 1), torch.ones(n * d, device=dev), (n, n)), device=dev)
    adj = gnntf.Adjacency(g)
    ms = timed(lambda: _launch(adj, H, H0, 0.9, 0.1, 0, out=out), reps=5, warm=2)
    b = bench.alg_bytes_per_iteration(n, g.nnz, C)
    print(json.dumps({"d": d, "nnz": g.nnz, "ms": round(ms, 3), "alg_TBs": round(b / ms / 1e9, 3), "kernel": g.last_kernel()}))
    del g, adj, rows, cols
