#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench): the fused SpMM+mix at several feature widths on
the config-4 graph, the arxiv-shaped GCN forward of config 3, and the training-mode step.
    python tools/bench_widths.py [--nodes N --entries E] > gpurun_out/widths.json
"""
import argparse
import json
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import bench
import gnntf


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--widths", type=str, default="8,16,32,64,128,256,512")
    ap.add_argument("--skip-arxiv", action="store_true")
    ap.add_argument("--skip-train", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    out = {"graph": {"nodes": args.nodes, "entries": args.entries}, "widths": [], "peak_GBs": bench.HBM_PEAK_GBS}
    g, adj, prep = bench.build_single(args, dev)
    out["prep"] = prep
    n, nnz = g.n_rows, g.nnz
    for C in [int(c) for c in args.widths.split(",")]:
        H = torch.rand(n, C, device=dev) * 2 - 1
        H0 = torch.rand(n, C, device=dev) * 2 - 1
        buf = torch.empty_like(H)
        from gnntf.sparse import _launch
        ms = timed(lambda: _launch(adj, H, H0, 0.9, 0.1, 0, out=buf))
        b = bench.alg_bytes_per_iteration(n, nnz, C)
        kernel = g.last_kernel()
        with torch.no_grad():
            loop_ms = timed(lambda: gnntf.appnp_propagate(adj, H0, 0.1, 10), reps=3, warm=1)
            gnntf.sparse.PAD_WIDTHS = False
            unpadded_ms = timed(lambda: gnntf.appnp_propagate(adj, H0, 0.1, 10), reps=3, warm=1)
            gnntf.sparse.PAD_WIDTHS = True
        out["widths"].append({"C": C, "kernel": kernel, "ms": ms, "k10_loop_ms": loop_ms, "k10_loop_unpadded_ms": unpadded_ms, "loop_kernel": g.last_kernel(), "edges_per_s": nnz / ms * 1e3,
                              "alg_GBs": b / ms / 1e6, "frac": b / ms / 1e6 / bench.HBM_PEAK_GBS})
        del H, H0, buf
    if not args.skip_train:
        # training-mode cost of one PPRIteration: dropout + renormalise (3 passes) + fused step + backward
        C = 64
        H = (torch.rand(n, C, device=dev) * 2 - 1).requires_grad_()
        H0 = (torch.rand(n, C, device=dev) * 2 - 1).requires_grad_()
        gout = torch.rand(n, C, device=dev)
        t_norm = timed(lambda: gnntf.normalize(g, "symmetric", "none", dropout=0.5, seed=1, stream_id=3))
        dadj = gnntf.normalize(g, "symmetric", "none", dropout=0.5, seed=1, stream_id=3)
        t_fwd = timed(lambda: gnntf.ppr_step(dadj, H, H0, 0.1))

        def fb():
            H.grad = None; H0.grad = None
            gnntf.ppr_step(dadj, H, H0, 0.1).backward(gout)
        t_fb = timed(fb)
        out["train_C64"] = {"dropout_normalize_ms": t_norm, "forward_ms": t_fwd, "forward_backward_ms": t_fb}
        del H, H0, gout, dadj
    del g, adj
    torch.cuda.empty_cache()
    if not args.skip_arxiv:
        # config 3: arxiv-shaped 2-layer GCN forward (N=169,343; ~2.3M stored entries; 128 -> 64 -> 40)
        a2 = argparse.Namespace(nodes=169_343, entries=2_332_486)
        g3, adj3, _ = bench.build_single(a2, dev)
        X = torch.randn(g3.n_rows, 128, device=dev)
        model = gnntf.GCN(g3, X, num_classes=40)
        model.training_mode(False)
        with torch.no_grad():
            t_fwd = timed(lambda: model(model.features), reps=20, warm=5)
            X64 = torch.randn(g3.n_rows, 64, device=dev)
            t128 = timed(lambda: gnntf.spmm(adj3, X), reps=20, warm=5)
            t64 = timed(lambda: gnntf.spmm(adj3, X64), reps=20, warm=5)
        out["arxiv_gcn"] = {"nodes": g3.n_rows, "entries": g3.nnz, "forward_ms": t_fwd, "spmm128_ms": t128, "spmm64_ms": t64,
                            "spmm128_edges_per_s": g3.nnz / t128 * 1e3, "spmm64_edges_per_s": g3.nnz / t64 * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
