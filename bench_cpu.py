"""bench_cpu.py -- the cpu_baseline leg of bench.py: the oracle's CPU restatements of gnntf's TF-CPU path timed
on the host cores of the same box, in the same run, on a bounded sample.  The only place of the bench that touches oracle/, and
only after the timed region: a reported baseline, never the thing measured."""
import ctypes
import os
import time


def cpu_baseline(g, adj, H0, args):
    """One of the K iterations over a row prefix of the SAME workload, three ways (every cost is linear in the entries walked, the
    gathers span all of H): (iii) the C / OpenMP port, (i) scipy CSR on one thread, (ii) torch.sparse.mm on all threads.
    ``value`` is the strongest of the three that does what the reference does -- the C / OpenMP port with the per-iteration
    renormalisation (gnn.py:36-50 called from filter.py:18): whole-graph normalisation timed once over all entries, SpMM + mix on
    the sample and scaled to all entries, value = nnz / (t_norm + t_spmm * nnz / e)."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    from oracle import cpu_baselines as cb
    lib = ctypes.CDLL(ge.build_oracle())
    sig = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    lib.oracle_sample_iteration_par.restype = ctypes.c_int
    lib.oracle_sample_iteration_par.argtypes = sig
    lib.oracle_num_threads.restype = ctypes.c_int
    rowptr, colidx, vals = (t.cpu().numpy() for t in g.csr_arrays())
    nvals = adj.vals.cpu().numpy()                                                # A_hat's values (device normalisation)
    H = H0.cpu().numpy()
    n, C = H.shape
    nnz = int(rowptr[-1])
    budget = max(args.cpu_seconds, 3.0)

    def run(rows, renorm=1):
        out = np.empty((max(rows, 1), C), dtype=np.float32)
        given = vals if renorm else nvals                                         # renorm = 0: the values are A_hat's already
        t0 = time.time()
        rc = lib.oracle_sample_iteration_par(n, rows, rowptr.ctypes.data, colidx.ctypes.data, given.ctypes.data, H.ctypes.data,
                                             H.ctypes.data, args.alpha, C, out.ctypes.data, renorm)
        assert rc == 0
        return time.time() - t0, int(rowptr[rows]), out

    def rows_for(rate, seconds, floor):
        """Row prefix whose entries take about ``seconds`` at ``rate`` entries/s."""
        want = int(min(nnz, max(floor, seconds * rate)))
        return int(min(n, max(1, np.searchsorted(rowptr, want, side="left"))))

    # (iii) the oracle's C / OpenMP port, all host threads
    probe_rows = rows_for(1.0, 0.0, min(nnz, 2_000_000))
    t_probe, e_probe, _ = run(probe_rows, renorm=0)
    t_norm, _, _ = run(0)                                                         # the whole-graph renormalisation alone
    rows = rows_for(e_probe / max(t_probe, 1e-3), 0.4 * budget, e_probe)
    t_only, e, ref_out = run(rows, renorm=0)
    cores = int(lib.oracle_num_threads())
    port = {"value": nnz / (t_norm + t_only * nnz / e), "unit": "edges/s", "cores": cores, "kind": "port",
            "spmm_only_value": e / t_only,
            "sample_short": f"C/OpenMP oracle port, {cores} thr: 1 of {args.iterations} iter.; renorm all {nnz} entries {t_norm:.1f}s + "
                            f"SpMM first {rows} rows {t_only:.1f}s, scaled",
            "sample": f"C / OpenMP port of the oracle (oracle/propagate_ref.c), all host threads: 1 of {args.iterations} iterations; the whole-graph "
                      f"renormalisation the reference does in every iteration (gnn.py:36-50 called from filter.py:18) timed over all {nnz} "
                      f"entries ({t_norm:.2f} s), SpMM + mix over the first {rows} of {n} rows ({e} entries, C={C}: {t_only:.2f} s) and scaled "
                      f"to all entries: value = nnz / (t_norm + t_spmm * nnz / e); spmm_only_value = e / t_spmm with the adjacency "
                      f"normalised beforehand (what the GPU figure times); CPU restatement of gnntf's TF-CPU path (TensorFlow unavailable)"}
    # (i) scipy CSR on ONE thread, re-normalising inside the iteration (sample-sized: its cost is linear in the entries too)
    rows1 = rows_for(2e6, 0.0, min(nnz, 1_000_000))
    _, tn, ts = cb.scipy_iteration(rowptr, colidx, vals, H, H, args.alpha, rows1)
    e1 = int(rowptr[rows1])
    rows1 = rows_for(e1 / max(tn + ts, 1e-3), 0.35 * budget, e1)
    _, tn, ts = cb.scipy_iteration(rowptr, colidx, vals, H, H, args.alpha, rows1)
    e1 = int(rowptr[rows1])
    port["scipy_single_thread"] = {"value": e1 / (tn + ts), "unit": "edges/s", "cores": 1, "kind": "port", "spmm_only_value": e1 / ts,
                                   "sample": f"scipy.sparse CSR @ dense, float32, one thread (oracle/cpu_baselines.py): first {rows1} rows ({e1} entries): "
                                             f"column sums + divide_no_nan + two value scalings of those entries {tn:.2f} s, SpMM + mix {ts:.2f} s"}
    # (ii) torch.sparse.mm on all host threads, adjacency normalised beforehand
    rows2 = rows_for(e / t_only / 4, 0.25 * budget, min(nnz, 2_000_000))
    out2, t2, used = cb.torch_sparse_iteration(rowptr, colidx, nvals, H, H, args.alpha, rows2)
    e2 = int(rowptr[rows2])
    agree = None
    if rows2 <= rows:                                                            # same rows, same values: the two restatements must agree
        agree = float(np.abs(out2 - ref_out[:rows2]).max())
    port["torch_sparse_all_threads"] = {"value": e2 / t2, "unit": "edges/s", "cores": int(used), "kind": "port",
                                        "sample": f"torch.sparse.mm (CPU, COO, {used} threads) + mix over the first {rows2} rows ({e2} entries), adjacency "
                                                  f"normalised beforehand: {t2:.2f} s; max |difference| to the C port on those rows: {agree}"}
    port["host"] = {"os_cpu_count": os.cpu_count(), "torch_threads": int(torch.get_num_threads())}
    return port
