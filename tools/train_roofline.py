#!/usr/bin/env python3
"""The training-mode step bench.py times as ``secondary.training_step_C64`` -- K = 10 PPR iterations forward + backward, every
iteration with its own dropped + re-normalised adjacency (layered.py:47-50 + gnn.py:37-42), weights produced inside the SpMM --
alone in a process, for rocprofv3 (kernel trace, then FETCH_SIZE / WRITE_SIZE passes of their own):

    python3 tools/train_roofline.py [--feats 64] [--steps 2] [--launches 10]

The run is cut into SEGMENTS by a marker kernel (k_stream, a 64-float gnx_stream_read that nothing else here launches), so that
profiles/summarize_train.py can tell the forward launches from the backward ones (same kernel names on a symmetric graph):

    warm-up | MARK | ``launches`` forward iterations (gnx_spmm_dropped_chained, k >= 1) | MARK | ``launches`` backward iterations
    (gnx_spmm_dropped_back, a middle iteration) | MARK | 3 x the degree scales of all K streams (gnx_graph_colsum_streams) | MARK | ``steps``
    whole steps | MARK

Prints one JSON line: ms per launch / step by events, entries kept per dropout stream, the byte model (bench.alg_bytes_dropped_*)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import bench
import gnntf
from gnntf import _native as nat
from gnntf import sparse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--iterations", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    g, adj, _ = bench.build_single(a, dev)
    K, C, n = a.iterations, a.feats, g.n_rows
    H0 = (torch.rand(n, C, device=dev) * 2 - 1).requires_grad_()
    gout = torch.rand(n, C, device=dev)
    mark_src, mark_sink = torch.zeros(64, device=dev), torch.zeros(64, device=dev)

    def mark():
        nat.check(nat.lib().gnx_stream_read(nat.ptr(mark_src), 64, nat.ptr(mark_sink), nat.current_stream()))

    def step():
        H0.grad = None
        scales = sparse.dropped_degree_scales(g, 0.5, 1, 0, K)
        make = lambda k, bwd=False: sparse.dropped_adjacency(g, 0.5, 1, k, D=scales[k])
        gnntf.ppr_loop(make, H0, 0.1, K).backward(gout)

    def timed(fn, reps):
        evs = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in evs)
        return ms[len(ms) // 2]

    step()                                                              # warm-up: transposed structure, long-row slabs, allocator
    torch.cuda.synchronize()
    res = {"n": n, "nnz": g.nnz, "C": C, "K": K}
    scales = sparse.dropped_degree_scales(g, 0.5, 1, 0, K)
    X = H0.detach()
    adj1 = sparse.dropped_adjacency(g, 0.5, 1, 1, D=scales[1])
    with torch.no_grad():
        sparse._launch_chained(adj1, X, X, 0.9, 0.1, True, scales[2], skip_empty=True)
        S_run, Y_run = torch.zeros_like(gout), torch.empty_like(gout)
        back = lambda: sparse._launch_back(adj1, gout, True, scales[0], S_run, 1.0, 0.09, S_run, 0.9, Y_run, skip_empty=True)
        back()
        torch.cuda.synchronize()
        mark()
        res["forward_launch_ms"] = timed(lambda: sparse._launch_chained(adj1, X, X, 0.9, 0.1, True, scales[2], skip_empty=True), a.launches)
        res["kernel"] = g.last_kernel()
        mark()
        res["backward_launch_ms"] = timed(back, a.launches)
        mark()
        res["degree_scales_all_streams_ms"] = timed(lambda: sparse.dropped_degree_scales(g, 0.5, 1, 0, K), 3)
        mark()
    res["ms_per_step"] = timed(step, a.steps)
    mark()
    torch.cuda.synchronize()
    kept = bench.kept_entries(g, 0.5, 1, 1, 1)[0]
    res["kept_entries_stream_1"] = kept
    res["alg_bytes_forward_launch"] = bench.alg_bytes_dropped_iteration(n, g.nnz, kept, C)
    res["alg_bytes_backward_launch"] = bench.alg_bytes_dropped_iteration(n, g.nnz, kept, C, backward=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
