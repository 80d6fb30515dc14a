#!/usr/bin/env python3
"""gnx_dense (float32 MFMA, csrc/gnx_dense.hip) timings -- the one script that replaces the former one-offs
dense_one / dense_only / dense_ab / dense_time / dense_wreg_ab.

    python3 tools/dense_bench.py                                  # the standard shapes, with torch beside them
    python3 tools/dense_bench.py --shapes 10000000x256x64 --launches 6 --quiet       # a bare loop for rocprofv3 (was dense_one / dense_only)
    python3 tools/dense_bench.py --shapes 10000000x256x64 --check                    # + max error against a float64 product on sampled rows
    GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so GNX_DENSE_WREG=0 python3 tools/dense_bench.py ...   # kernel A/B of the tuning build

Shapes are ROWSxFxO; every line: ms per call, TFLOP/s, GB/s of X read + output written."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import gnntf

STANDARD = "10000000x256x64,10000000x128x128,10000000x64x256,10000000x128x64,10000000x256x7,2000000x100x40"


def timed(fn, launches, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(launches):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / launches * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=STANDARD)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--torch", action="store_true", help="also time torch.relu(torch.addmm(b, X, W)) (hipBLASLt / rocBLAS + separate bias / relu passes)")
    ap.add_argument("--check", action="store_true", help="max error against a float64 product on sampled rows")
    ap.add_argument("--quiet", action="store_true", help="launch only (profiler runs): no timing lines")
    a = ap.parse_args()
    env = {k: os.environ[k] for k in ("GNX_DENSE_WREG", "GNX_DENSE_RING", "GNX_LIBRARY") if k in os.environ}
    for shape in a.shapes.split(","):
        n, F, O = (int(float(x)) for x in shape.split("x"))
        X, W, b = torch.randn(n, F, device="cuda"), torch.randn(F, O, device="cuda"), torch.randn(1, O, device="cuda")
        if a.quiet:
            for _ in range(a.launches):
                gnntf.dense(X, W, b, relu=True)
            torch.cuda.synchronize()
            continue
        line = dict(n=n, F=F, O=O, **env)
        if a.check:
            out = gnntf.dense(X, W, b, relu=True)
            rows = torch.cat([torch.arange(0, min(4096, n), device="cuda"), torch.randint(0, n, (8192,), device="cuda"), torch.arange(max(n - 4096, 0), n, device="cuda")])
            line["max_err_vs_f64"] = float((out[rows].double() - torch.relu(X[rows].double() @ W.double() + b.double())).abs().max())
        ms = timed(lambda: gnntf.dense(X, W, b, relu=True), a.launches)
        line.update(ms=round(ms, 3), TF=round(2 * n * F * O / ms / 1e9, 1), GBs=round(4 * n * (F + O) / ms / 1e6, 1))
        if a.torch:
            line["torch_addmm_relu_ms"] = round(timed(lambda: torch.relu(torch.addmm(b, X, W)), a.launches), 3)
        print(line, flush=True)
        del X, W, b


if __name__ == "__main__":
    main()
