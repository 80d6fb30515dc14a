#!/usr/bin/env python3
"""bench.py -- propagated edges/s of the APPNP K=10 propagation on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one K-iteration propagation H <- (1-a) A_hat H + a H0 (K = 10) over the resident
synthetic graph, through the C ABI of libgnx.so.  N = 1: BASELINE.json configs[3], the roofline
run (RMAT 10M nodes / 100M stored entries, 256 float32 features).  N > 1: the same generator with
10M nodes / 100M entries PER GPU (weak scaling), 1-D vertex shards, halo rows exchanged over
RCCL every iteration.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=10_000_000, help="vertices per GPU")
    ap.add_argument("--entries", type=int, default=100_000_000, help="stored directed entries per GPU")
    ap.add_argument("--feats", type=int, default=256)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--grid", type=str, default="", help="PVxPF process grid (vertex blocks x feature slices); default: chosen by gnntf.sharded.choose_grid")
    ap.add_argument("--force-sharded", action="store_true", help="run the vertex-partitioned path even with one rank (rehearsal)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    return ap.parse_args()


# ---- synthetic graph --------------------------------------------
def rmat_pairs(scale, m, gen, device, a=0.57, b=0.19, c=0.19):
    src = torch.zeros(m, dtype=torch.int64, device=device)
    dst = torch.zeros(m, dtype=torch.int64, device=device)
    for _ in range(scale):
        r = torch.rand(m, device=device, generator=gen)
        src = src * 2 + (r >= a + b).long()
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()
    return src, dst


def rmat_undirected_keys(n, m_undirected, seed, device):
    """Exactly m_undirected distinct undirected edges {u < v} of an R-MAT graph relabelled onto n
    vertices (ids folded by modulo, self loops dropped), as int64 keys u * n + v."""
    gen = torch.Generator(device=device).manual_seed(seed)
    scale = max(1, int(np.ceil(np.log2(n))))
    keys = torch.empty(0, dtype=torch.int64, device=device)
    while keys.numel() < m_undirected:
        need = m_undirected - keys.numel()
        s, d = rmat_pairs(scale, int(need * 1.25) + 1024, gen, device)
        s, d = s % n, d % n
        keep = s != d
        s, d = s[keep], d[keep]
        lo, hi = torch.minimum(s, d), torch.maximum(s, d)
        keys = torch.unique(torch.cat([keys, lo * n + hi]))
        del s, d, lo, hi, keep
    if keys.numel() > m_undirected:
        pick = torch.randperm(keys.numel(), device=device, generator=gen)[:m_undirected]
        keys = keys[pick]
    return keys


def build_single(args, device):
    import gnntf
    n, m = args.nodes, args.entries // 2
    t0 = time.time()
    keys = rmat_undirected_keys(n, m, seed=1, device=device)
    gen = torch.Generator(device=device).manual_seed(3)
    perm = torch.randperm(n, device=device, generator=gen)          # random vertex relabelling (seed 3)
    u, v = perm[keys // n], perm[keys % n]
    del keys
    idx = torch.cat([torch.stack([u, v], 1), torch.stack([v, u], 1)])      # symmetrised COO, unsorted
    del u, v, perm
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, n)), device=device)      # A0: COO -> CSR on the device
    del idx, vals
    adj = gnntf.normalize(g, "symmetric")                                          # A2, once (eval mode)
    torch.cuda.synchronize()
    t_prep = time.time() - t0
    torch.cuda.empty_cache()
    return g, adj, dict(gen_s=round(t_gen, 2), prep_s=round(t_prep, 2))


def alg_bytes_per_iteration(n, nnz, C):
    """nnz*(4 col + 4 val + 4C gathered row) + N*(4 rowptr + 4C H0 + 4C out)."""
    return nnz * (8 + 4 * C) + n * (4 + 8 * C)


# ---- CPU baseline: the oracle's C port on a bounded sample -------------------------------------------
def cpu_baseline(g, H0, args):
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_oracle())
    lib.oracle_sample_iteration.restype = ctypes.c_int
    lib.oracle_sample_iteration.argtypes = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64,
                                                                                                      ctypes.c_void_p]
    lib.oracle_num_threads.restype = ctypes.c_int
    rowptr, colidx, vals = (t.cpu().numpy() for t in g.csr_arrays())
    H = H0.cpu().numpy()
    n, C = H.shape
    nnz = int(rowptr[-1])

    def run(rows):
        out = np.empty((rows, C), dtype=np.float32)
        t0 = time.time()
        rc = lib.oracle_sample_iteration(n, rows, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, H.ctypes.data,
                                         H.ctypes.data, args.alpha, C, out.ctypes.data)
        assert rc == 0
        return time.time() - t0, int(rowptr[rows])

    probe_rows = min(n, 100_000)
    t_probe, e_probe = run(probe_rows)
    t_norm, _ = run(0)                                            # the whole-graph renormalisation alone
    rate = max(e_probe, 1) / max(t_probe - t_norm, 1e-3)           # entries/s of the SpMM+mix part
    rows = int(min(n, max(probe_rows, (args.cpu_seconds - t_norm) * rate / max(nnz / n, 1e-9))))
    t, e = run(rows)
    return {"value": e / t, "unit": "edges/s", "cores": int(lib.oracle_num_threads()), "kind": "port",
            "sample": f"1 of {args.iterations} iterations over the first {rows} of {n} rows ({e} entries, C={C}) incl. the "
                      f"per-iteration whole-graph renormalisation the reference does (gnn.py:36-50); {t:.1f} s; "
                      f"CPU restatement of gnntf's TF-CPU path (TensorFlow unavailable)"}


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: anything native libraries print there (RCCL's version banner)
    # is sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the propagation path has no CPU fallback")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import gnntf
    gnntf.set_default_device(device)
    sharded_path = world > 1 or args.force_sharded
    if sharded_path:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("GNX_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of several ranks on ONE card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    K, C, a = args.iterations, args.feats, args.alpha

    if not sharded_path:
        g, adj, prep = build_single(args, device)
        n_local, nnz_local, nnz_global = g.n_rows, g.nnz, g.nnz
        gen = torch.Generator(device=device).manual_seed(2)
        H0 = torch.rand(n_local, C, device=device, generator=gen) * 2 - 1       # U(-1, 1), seed 2
        out = torch.empty_like(H0)
        work = torch.empty_like(H0)
        from gnntf import _native as nat
        lib = nat.lib()

        def step():
            nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(out), nat.ptr(work),
                                              nat.current_stream()))
        halo = None
    else:
        from gnntf import sharded
        if args.grid:
            grid = tuple(int(x) for x in args.grid.split("x"))
        else:
            grid = sharded.choose_grid(world, C, args.nodes * world, args.entries * world)
        sg, prep, (gv, gf, pv, pf) = sharded.build_rmat_shard(args.nodes, args.entries, seed=1, device=device, grid=grid)
        n_local, nnz_local, nnz_global = sg.n_local, sg.nnz_local, sg.nnz_global
        C_local = C // pf                                                       # this rank's feature slice
        gen = torch.Generator(device=device).manual_seed(2 + rank)
        H0 = torch.rand(n_local, C_local, device=device, generator=gen) * 2 - 1
        state = sg.make_state(H0)

        def step():
            sg.propagate(state, a, K)
        halo = sg.halo_stats()

    def barrier():
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s, e in ev:                     # events sit on the stream the kernels are launched on
        s.record()
        step()
        e.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if sharded_path:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms = [s.elapsed_time(e) for s, e in ev]

    if rank == 0:
        edges = nnz_global * K * args.steps
        launch_s = (sum(step_ms) / len(step_ms)) / 1e3 / K          # one fused SpMM+mix launch (+ its long-row tail)
        b_alg = alg_bytes_per_iteration(n_local, nnz_local, C_local if sharded_path else C)
        achieved = b_alg / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            rec = json.load(open(tpath))
            if rec.get("workload") == f"rmat_n{args.nodes}_nnz{args.entries}_C{C}" and not sharded_path:
                traffic = rec.get("hbm_bytes_per_launch")
        result = {
            "metric": "propagated edges/sec (APPNP K=10)", "value": edges / elapsed, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"rmat_n{args.nodes}_nnz{args.entries}_C{C}_appnp_K{K}" + ("" if world == 1 else f"_per_gpu_x{world}"),
                       "rows_per_rank": n_local, "stored_entries_per_rank": nnz_local, "stored_entries_total": nnz_global,
                       "features": C, "iterations": K, "alpha": a, "partition": (f"{pv}_vertex_blocks_x_{pf}_feature_slices" if sharded_path else "none"),
                       "halo": halo, "prep": prep, "kernel": (sg.graph.last_kernel() if sharded_path else g.last_kernel())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "alg_bytes_per_launch": b_alg, "launch_ms": launch_s * 1e3,
                         "note": "one launch = one fused SpMM+mix iteration incl. its long-row kernels"},
        }
        if not sharded_path and args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(g, H0, args)
        else:
            result["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
