"""CPU baselines (i) and (ii) of SURVEY.md section 8(d), beside the C/OpenMP port (iii, oracle/propagate_ref.c).

TEST INFRASTRUCTURE ONLY -- loaded by tests/ and by bench.py's ``cpu_baseline`` leg, never by the product path.
PARITY UNPINNED like the rest of oracle/ (see gnntf_oracle.py's header): restatements, checked against the numpy oracle in
tests/test_oracle_c.py, not against reference outputs.

Both time ONE propagation iteration of gnntf's TF-CPU path (paths relative to /root/reference)

    gnntf/core/gnn/gnn.py:36-50                      get_adjacency: column sums, divide_no_nan(1, sqrt(d)), D[i] v D[j]
    gnntf/core/gnn/architectures/filter.py:17-22     P = A_hat . H ;  H' = P (1-a) + H0 a

over the first ``rows`` rows of a CSR (a bounded sample: every cost below is linear in the entries walked, the gathers
still span all of H):

  (i)  scipy_iteration          scipy.sparse CSR, ONE thread, re-normalising inside the iteration as the reference does on every
                                PPRIteration call (filter.py:18 calls get_adjacency each time);
  (ii) torch_sparse_iteration   torch.sparse.mm on all host threads over an adjacency normalised beforehand.
"""
import time

import numpy as np
import scipy.sparse as sp


def divide_no_nan(x, y):
    """tf.math.divide_no_nan: 0 where the denominator is 0."""
    out = np.zeros_like(y)
    np.divide(x, y, out=out, where=y != 0)
    return out


def scipy_iteration(rowptr, colidx, raw_vals, H, H0, a, rows=None, colsum=None):
    """One faithful iteration over rows [0, rows) on ONE thread.  Returns (H' [rows, C], seconds of the normalisation, seconds of
    the SpMM + mix).  ``colsum`` None: the column sums are taken over the sampled rows only (exact when rows covers the matrix;
    on a sample the VALUES are then those of the sampled sub-matrix -- the cost per entry is what is measured)."""
    n = len(rowptr) - 1
    rows = n if rows is None else int(rows)
    e = int(rowptr[rows])
    ptr, col, val = rowptr[:rows + 1], colidx[:e], raw_vals[:e]
    t0 = time.perf_counter()
    if colsum is None:
        colsum = np.bincount(col, weights=val, minlength=H.shape[0]).astype(np.float32)      # tf.sparse.reduce_sum(axis=0), gnn.py:41
    D = divide_no_nan(np.float32(1.0), np.sqrt(colsum))                                       # gnn.py:41
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(ptr))
    nvals = (D[row_of] * val) * D[col]                                                       # gnn.py:42 (two cwise multiplies)
    t_norm = time.perf_counter() - t0
    t0 = time.perf_counter()
    A = sp.csr_matrix((nvals, col, ptr), shape=(rows, H.shape[0]))
    out = (A @ H) * np.float32(1.0 - a) + H0[:rows] * np.float32(a)                          # filter.py:19-21
    t_spmm = time.perf_counter() - t0
    return out.astype(np.float32, copy=False), t_norm, t_spmm


def torch_sparse_iteration(rowptr, colidx, nvals, H, H0, a, rows=None, threads=None):
    """SpMM + mix over rows [0, rows) with torch.sparse.mm on ``threads`` host threads (default: all); ``nvals`` are A_hat's values.
    Returns (H' [rows, C], seconds, threads used).  (COO: torch's CPU CSR sparse.mm crashed at the bench's shapes.)"""
    import os

    import torch
    n = len(rowptr) - 1
    rows = n if rows is None else int(rows)
    e = int(rowptr[rows])
    threads = int(threads or os.cpu_count() or 1)
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rowptr[:rows + 1]))
        idx = torch.from_numpy(np.stack([row_of, colidx[:e].astype(np.int64)]))
        A = torch.sparse_coo_tensor(idx, torch.from_numpy(np.ascontiguousarray(nvals[:e])), size=(rows, H.shape[0])).coalesce()
        Ht, H0t = torch.from_numpy(H), torch.from_numpy(H0)
        t0 = time.perf_counter()
        out = torch.sparse.mm(A, Ht) * (1.0 - a) + H0t[:rows] * a
        dt = time.perf_counter() - t0
        used = torch.get_num_threads()
    finally:
        torch.set_num_threads(before)
    return out.numpy(), dt, used
