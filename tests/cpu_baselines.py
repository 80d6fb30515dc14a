#!/usr/bin/env python3
"""The three CPU baselines of BASELINE.md section 3 on a bounded sample of config 4 (host cores of the
box): B1 faithful scipy (single thread, re-normalising every iteration), B2 torch.sparse CSR (all
threads, normalised once), B3 the oracle's C/OpenMP port (all threads, faithful).  One iteration over a
row prefix of the graph; entries/s."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import argparse
import numpy as np
import scipy.sparse as sp
import torch
import bench, gnntf
import __graft_entry__ as ge

dev = torch.device("cuda:0")
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
rowptr, colidx, vals = (t.cpu().numpy() for t in g.csr_arrays())
n, C, a = g.n_rows, 256, 0.1
H = np.random.default_rng(2).uniform(-1, 1, size=(n, C)).astype(np.float32)
rows = 500_000                                                   # sample: the first 500k rows
e = int(rowptr[rows])
out = {"sample_rows": rows, "sample_entries": e, "C": C, "host_threads": os.cpu_count()}

# B1: scipy, single thread, faithful (column sums + scaling over the WHOLE graph, then the sampled rows)
t0 = time.time()
A = sp.csr_matrix((vals, colidx, rowptr), shape=(n, n))
d = np.asarray(A.sum(axis=0)).ravel()
D = np.where(d > 0, 1 / np.sqrt(np.where(d > 0, d, 1)), 0).astype(np.float32)
An = sp.diags(D) @ A @ sp.diags(D)
t_norm = time.time() - t0
t0 = time.time()
P = An[:rows] @ H
out_b1 = P * np.float32(1 - a) + H[:rows] * np.float32(a)
t_b1 = time.time() - t0
out["B1_scipy_1thread"] = {"normalise_whole_graph_s": t_norm, "step_s": t_b1, "entries_per_s": e / (t_b1 + t_norm * e / len(vals))}

# B2: torch CPU CSR, all threads, normalised once
torch.set_num_threads(os.cpu_count())
sub = An[:rows].tocoo()                                       # (torch's CPU CSR sparse.mm segfaults at this shape; COO works)
At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([sub.row, sub.col]).astype(np.int64)), torch.from_numpy(sub.data),
                             size=(rows, n)).coalesce()
Ht = torch.from_numpy(H)
t0 = time.time()
o2 = torch.sparse.mm(At, Ht) * (1 - a) + Ht[:rows] * a
t_b2 = time.time() - t0
out["B2_torch_cpu"] = {"step_s": t_b2, "entries_per_s": e / t_b2, "threads": torch.get_num_threads()}

# B3: C/OpenMP port (faithful)
lib = ctypes.CDLL(ge.build_oracle())
lib.oracle_sample_iteration.argtypes = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64, ctypes.c_void_p]
o3 = np.empty((rows, C), dtype=np.float32)
t0 = time.time()
lib.oracle_sample_iteration(n, rows, rowptr.ctypes.data, colidx.ctypes.data, vals.ctypes.data, H.ctypes.data, H.ctypes.data, a, C, o3.ctypes.data)
t_b3 = time.time() - t0
out["B3_c_openmp"] = {"step_plus_normalise_s": t_b3, "entries_per_s": e / t_b3, "threads": lib.oracle_num_threads()}
np.testing.assert_allclose(o3, out_b1, rtol=1e-4, atol=1e-5)
print(json.dumps(out, indent=1))
