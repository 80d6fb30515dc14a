#!/usr/bin/env python3
"""Interleaved A/B of SpMM kernel variants in ONE process (cdna guide rule 24).
Needs a TUNING build of the library (the product build has no variant switches):
    make -C gnn-tf_amd/csrc clean && make -C gnn-tf_amd/csrc TUNING=1
    python tools/tune_spmm.py --feats 256 --variants 0,1,2,3,6,7,16,32,48,64"""
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
from gnntf.sparse import _launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=int, default=256)
    ap.add_argument("--variants", type=str, default="0,1,2,4,6,7,16,32,48,64")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    g, adj, _ = bench.build_single(a, dev)
    lib = nat.lib()
    if not hasattr(lib, "gnx_debug_set_tune"):
        raise SystemExit("tune_spmm.py needs a tuning build: make -C gnn-tf_amd/csrc clean && make -C gnn-tf_amd/csrc TUNING=1")
    lib.gnx_debug_set_tune.argtypes = [ctypes.c_int]
    n, C = g.n_rows, a.feats
    H = torch.rand(n, C, device=dev) * 2 - 1
    H0 = torch.rand(n, C, device=dev) * 2 - 1
    buf = torch.empty_like(H)
    variants = [int(v) for v in a.variants.split(",")]
    ref = None
    times = {v: [] for v in variants}
    for r in range(a.rounds + 1):
        for v in variants:
            lib.gnx_debug_set_tune(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                _launch(adj, H, H0, 0.9, 0.1, 0, out=buf)
            e.record()
            torch.cuda.synchronize()
            if r == 0:
                if ref is None:
                    ref = buf[:1 << 20].clone()
                else:
                    assert torch.allclose(ref, buf[:1 << 20], rtol=1e-5, atol=1e-6), f"variant {v} changes the result"
            else:
                times[v].append(s.elapsed_time(e) / 3)
    b = bench.alg_bytes_per_iteration(n, g.nnz, C)
    for v in variants:
        t = sorted(times[v])
        print(json.dumps({"variant": v, "median_ms": t[len(t) // 2], "min_ms": t[0], "frac_median": b / t[len(t) // 2] / 1e6 / 8000}))


if __name__ == "__main__":
    main()
