#!/usr/bin/env python3
"""A/B of the training-mode row kernels (k_spmm_group_drop / k_spmm_long_partial_group_drop) on the config-4 graph: gathers in
flight per lane (U = 4 / 8) and the index prefetch across rounds (PIPE), forward (gnx_spmm_dropped_chained) and backward
(gnx_spmm_dropped_back) launches.  Needs the tuning build:
    make -C gnn-tf_amd/csrc TUNING=1
    GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so python3 tools/drop_ab.py --feats 64,128"""
import argparse
import ctypes
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

VARIANTS = (("u4", 0), ("u8", 1 << 17), ("u4_pipe", 1 << 19), ("u8_pipe", (1 << 17) | (1 << 19)), ("u4_long_u8", 1 << 18),
            ("u8_pipe_long_u8", (1 << 17) | (1 << 18) | (1 << 19)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=str, default="64,128")
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    g, adj, _ = bench.build_single(a, dev)
    lib = nat.lib()
    lib.gnx_debug_set_tune.argtypes = [ctypes.c_int]
    n = g.n_rows
    scales = sparse.dropped_degree_scales(g, 0.5, 1, 0, 3)
    adj1 = sparse.dropped_adjacency(g, 0.5, 1, 1, D=scales[1])
    for C in [int(c) for c in a.feats.split(",")]:
        X = torch.rand(n, C, device=dev) * 2 - 1
        G = torch.rand(n, C, device=dev)
        S, Y = torch.zeros_like(G), torch.empty_like(G)
        fwd = lambda: sparse._launch_chained(adj1, X, X, 0.9, 0.1, True, scales[2])
        bwd = lambda: sparse._launch_back(adj1, G, True, scales[0], S, 0.0, 0.09, S, 0.9, Y)
        line, ref = {"C": C}, {}
        for name, tune in VARIANTS:
            res = {}
            for what, fn in (("forward", fwd), ("backward", bwd)):
                times = []
                for r in range(a.rounds + 1):
                    lib.gnx_debug_set_tune(tune)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); out = fn(); e.record()
                    torch.cuda.synchronize()
                    if r:
                        times.append(s.elapsed_time(e))
                res[what + "_ms"] = sorted(times)[len(times) // 2]
                got = (out if what == "forward" else Y).clone()
                if name == "u4":
                    ref[what] = got
                res[what + "_same_bits"] = bool(torch.equal(got, ref[what]))
            res["kernel"] = g.last_kernel()
            line[name] = res
        lib.gnx_debug_set_tune(-1)
        print(json.dumps(line), flush=True)
        del X, G, S, Y, ref


if __name__ == "__main__":
    main()
