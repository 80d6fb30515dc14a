#!/usr/bin/env python3
"""Does vertex ORDER matter for the row-gather kernel?  Same RMAT graph, three labelings:
random (the workload's), degree-descending, and degree-descending for columns only via the same
relabel.  Prints ms per fused launch at C=256 and C=32."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
from gnntf.sparse import _launch
from tools.bench_widths import timed

dev = torch.device("cuda:0")
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
widths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 32]
a = argparse.Namespace(nodes=10_000_000 * mult, entries=100_000_000 * mult)
g, adj, _ = bench.build_single(a, dev)
n = g.n_rows
rowptr, colidx, raw, rows = g.csr_arrays(with_rows=True)
deg = rowptr[1:] - rowptr[:-1]
order = torch.argsort(deg, descending=True, stable=True)          # new id -> old id
newid = torch.empty_like(order); newid[order] = torch.arange(n, device=dev)
labelings = {"random": None, "degree_desc": newid}
for name, relabel in labelings.items():
    if relabel is None:
        gg, aa = g, adj
    else:
        idx = torch.stack([relabel[rows.long()], relabel[colidx.long()]], 1)
        gg = gnntf.DeviceGraph(gnntf.SparseCOO(idx, raw, (n, n)), device=dev)
        aa = gnntf.normalize(gg, "symmetric")
    for C in widths:
        H = torch.rand(n, C, device=dev); H0 = torch.rand(n, C, device=dev); out = torch.empty_like(H)
        ms = timed(lambda: _launch(aa, H, H0, 0.9, 0.1, 0, out=out), reps=5, warm=2)
        print(json.dumps({"labeling": name, "C": C, "ms": round(ms, 3), "kernel": gg.last_kernel()}))
        del H, H0, out
