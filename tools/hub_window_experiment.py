#!/usr/bin/env python3
"""Experiment: narrow feature widths, hub columns processed in a pass of their own (their rows compacted into a table small
enough to stay in L2), the rest in a second pass.  Emulated with two DeviceGraphs through the public API.
    python tools/hub_window_experiment.py"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
from gnntf.sparse import _launch

dev = torch.device("cuda:0")
gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
n = g.n_rows
rowptr, colidx, raw, rows = g.csr_arrays(with_rows=True)
vals = adj.vals
indeg = torch.bincount(colidx.long(), minlength=n)
order = torch.argsort(indeg, descending=True, stable=True)
out = {}
for C in (8, 16, 32):
    H = torch.rand(n, C, device=dev) * 2 - 1
    H0 = torch.rand(n, C, device=dev) * 2 - 1
    buf = torch.empty_like(H)
    t_plain = bench.median_ms(lambda: _launch(adj, H, H0, 0.9, 0.1, 0, out=buf), reps=5, warm=2)
    ref = buf.clone()
    res = {"plain_ms": t_plain}
    for K in (16384, 65536, 262144, 1048576):
        hub_ids = order[:K].contiguous()
        slot = torch.full((n,), -1, dtype=torch.int64, device=dev)
        slot[hub_ids] = torch.arange(K, device=dev)
        s = slot[colidx.long()]
        is_hub = s >= 0
        share = float(is_hub.float().mean())
        gh = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows.long()[is_hub], s[is_hub]], 1), vals[is_hub], (n, K)), device=dev)
        gr = gnntf.DeviceGraph(gnntf.SparseCOO(torch.stack([rows.long()[~is_hub], colidx.long()[~is_hub]], 1), vals[~is_hub], (n, n)), device=dev)
        ah, ar = gnntf.Adjacency(gh, None), gnntf.Adjacency(gr, None)
        S = torch.empty_like(H)

        def step():
            Hh = gnntf.gather_rows(H, hub_ids)
            _launch(ah, Hh, H0, 0.9, 0.1, 0, out=S)
            _launch(ar, H, S, 0.9, 1.0, 0, out=buf)
        t = bench.median_ms(step, reps=5, warm=2)
        err = float((buf - ref).abs().max())
        t_h = bench.median_ms(lambda: _launch(ah, gnntf.gather_rows(H, hub_ids), H0, 0.9, 0.1, 0, out=S), reps=5, warm=2)
        res[f"K{K}"] = {"hub_entry_share": round(share, 3), "two_pass_ms": t, "hub_pass_ms": t_h, "max_abs_diff": err}
        del gh, gr, ah, ar, S, slot, s, is_hub
    out[f"C{C}"] = res
    print(C, json.dumps(res), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "hub_window.json"), "w"), indent=1)
