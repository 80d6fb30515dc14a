#!/usr/bin/env python3
"""A/B of the K loop's row kernel on the config-4 graph: cooperative vs per-lane index fetch, and the floor of a launch whose
gathers all hit (every gather reads row 0).  Needs the tuning build (side by side with the product library):
    make -C gnn-tf_amd/csrc TUNING=1
    GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so python tools/narrow_ab.py --feats 8,16,32,64"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--entries", type=int, default=100_000_000)
    ap.add_argument("--feats", type=str, default="8,16,32")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--only", type=str, default="", help="comma-separated variant names to run (default: all)")
    ap.add_argument("--variants", type=str, default="", help="name:tune,... replaces the built-in variant list (tuning build's gnx_debug_set_tune bits)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gnntf.set_default_device(dev)
    g, adj, _ = bench.build_single(a, dev)
    lib = nat.lib()
    tunable = hasattr(lib, "gnx_debug_set_tune")
    if tunable:
        lib.gnx_debug_set_tune.argtypes = [ctypes.c_int]
    n, K = g.n_rows, 10
    for C in [int(c) for c in a.feats.split(",")]:
        H0 = torch.rand(n, C, device=dev) * 2 - 1
        out, work = torch.empty_like(H0), torch.empty_like(H0)
        run = lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), 0.1, K, C, nat.ptr(out), nat.ptr(work),
                                                        nat.current_stream()))
        res = {}
        variants = (("coop_index_fetch", 0), ("per_lane_index_fetch", 32768), ("coop_all_gathers_hit", 16384), ("per_lane_all_gathers_hit", 16384 | 32768))
        if a.variants:
            variants = tuple((v.split(":")[0], int(v.split(":")[1])) for v in a.variants.split(","))
        if a.only:
            variants = tuple(v for v in variants if v[0] in a.only.split(","))
        for name, tune in variants if tunable else (("default", None),):
            times = []
            for r in range(a.rounds + 1):
                if tune is not None:
                    lib.gnx_debug_set_tune(tune)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); run(); e.record()
                torch.cuda.synchronize()
                if r:
                    times.append(s.elapsed_time(e))
            res[name] = {"ms_per_K10": sorted(times)[len(times) // 2], "kernel": g.last_kernel()}
            res[name + "_result"] = out.clone()
        if tunable:
            lib.gnx_debug_set_tune(-1)
            diff = float((res["coop_index_fetch_result"] - res["per_lane_index_fetch_result"]).abs().max()) if "per_lane_index_fetch_result" in res and "coop_index_fetch_result" in res else None
        else:
            diff = None
        b = bench.alg_bytes_per_iteration(n, g.nnz, C)
        line = {"C": C, "max_abs_diff": diff}
        for k, v in res.items():
            if not k.endswith("_result"):
                line[k] = dict(v, frac_of_8TBs=b * K / v["ms_per_K10"] / 1e6 / 8000)
        print(json.dumps(line), flush=True)
        del H0, out, work, res


if __name__ == "__main__":
    main()
