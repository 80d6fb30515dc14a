"""Worker of tests/test_sharded_cpu.py: one gloo rank of the vertex-partitioned propagation.
The checker backend below stands in for libgnx.so on CPU ranks (tests may use the oracle; the
product backend is gnntf.sharded.NativeBackend)."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd"), os.path.join(ROOT, "tests")]

import graphs  # noqa: E402
from gnntf import sharded  # noqa: E402
from oracle import gnntf_oracle as orc  # noqa: E402


class _G:
    def __init__(self, rowptr, colidx, vals, shape):
        self.rowptr, self.colidx, self.vals, self.shape = rowptr, colidx, vals, shape
        self.nnz, self.n_rows, self.n_cols = len(colidx), shape[0], shape[1]

    def last_kernel(self):
        return "oracle"


class OracleBackend:
    def graph_from_coo(self, idx, vals, shape):
        return _G(*orc.coo_to_csr_coalesced(idx.numpy(), vals.numpy(), shape), shape)

    def graph_from_csr(self, rowptr, colidx, vals, shape):
        return _G(rowptr.numpy(), colidx.numpy(), vals.numpy(), shape)

    def csr_arrays(self, g):
        return torch.from_numpy(g.rowptr), torch.from_numpy(g.colidx), torch.from_numpy(g.vals)

    def colsum(self, g):
        out = np.zeros(g.n_cols, dtype=np.float32)
        np.add.at(out, g.colidx, g.vals)
        return torch.from_numpy(out)

    def degree_scale(self, deg, normalized="symmetric"):
        deg.copy_(torch.from_numpy(orc.divide_no_nan(np.float32(1), np.sqrt(deg.numpy()))))
        return deg

    def scale_values(self, g, row_scale, col_scale):
        rows = np.repeat(np.arange(g.n_rows), np.diff(g.rowptr))
        return torch.from_numpy(row_scale.numpy()[rows] * g.vals * col_scale.numpy()[g.colidx])

    def spmm_mix(self, g, vals, X, H0, beta, alpha, out, out_rows=None):
        m = sp.csr_matrix((g.vals if vals is None else vals.numpy(), g.colidx, g.rowptr), shape=g.shape)
        res = torch.from_numpy((m @ X.numpy()) * np.float32(beta) + H0.numpy() * np.float32(alpha))
        if out_rows is None:
            out.copy_(res)
        else:
            out[out_rows.long()] = res

    def gather_rows(self, X, idx):
        return X[idx].contiguous()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mode = sys.argv[1]
    on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"          # GPU box: every rank shares cuda:0, libgnx.so backend
    dev = torch.device("cuda:0" if on_gpu else "cpu")
    backend = None if on_gpu else OracleBackend()
    C, K, a = (64 if on_gpu else 12), 10, 0.1
    pv, pf = world, 1
    if mode == "slices":
        n = 1003                                                   # not divisible by the world size
        coo, vals, shape = graphs.rmat_symmetric_coo(n, 9000, seed=5)
        dup = coo[:200]                                            # duplicates to coalesce
        coo, vals = np.concatenate([coo, dup]), np.concatenate([vals, np.full(200, 0.5, dtype=np.float32)])
        bounds = sharded.uniform_bounds(n, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
        sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend)
        v, f = rank, 0
    else:                                                          # the bench's distributed generator on a pv x pf grid
        if mode.startswith("grid"):
            pv, pf = (int(x) for x in mode[4:].split("x"))
        sg, _, (v, f, pv, pf) = sharded.build_rmat_shard(500, 6000, seed=1, device=dev, backend=backend, grid=(pv, pf))
        n, lo, hi = sg.n_global, sg.lo, sg.hi
    cs = C // pf                                                   # this rank's feature slice
    H0_full = np.random.default_rng(1).uniform(-1, 1, size=(n, C)).astype(np.float32)
    H0 = torch.from_numpy(H0_full[lo:hi, f * cs:(f + 1) * cs].copy()).to(dev)
    state = sg.make_state(H0)
    out = sg.propagate(state, a, K).clone()
    again = sg.propagate(state, a, K).clone()
    assert torch.equal(out, again), "propagate is not repeatable"
    assert sg.n_buf == sg.n_low + sg.n_local + sg.n_high and sum(sg.recv_counts) == sg.n_low + sg.n_high
    assert sg.recv_counts[sg.rank] == 0 and sg.send_counts[sg.rank] == 0 and sg.world == pv
    if pv == 1:
        assert sg.n_low + sg.n_high == 0                           # feature slices alone: no halo, no exchange
        assert sg.row_order is not None                            # ... and the shard is stored degree-relabelled

    # every rank's shard, mapped back to global ids (undoing the monotonic column remap)
    rowptr, colidx, nvals = (t.cpu().numpy() for t in sg.backend.csr_arrays(sg.graph))
    halo = sg.halo_ids.cpu().numpy()
    pos = colidx.astype(np.int64)
    gcol = np.where(pos < sg.n_low, halo[np.minimum(pos, max(len(halo) - 1, 0))] if len(halo) else 0,
                    np.where(pos < sg.n_low + sg.n_local, pos - sg.n_low + lo,
                             halo[np.clip(pos - sg.n_local, 0, max(len(halo) - 1, 0))] if len(halo) else 0))
    grow = np.repeat(np.arange(sg.n_local), np.diff(rowptr)) + lo
    if sg.row_order is not None:                                   # relabelled single-block shard: back to the caller's ids
        order = sg.row_order.cpu().numpy()
        grow, gcol = order[grow], order[gcol]
        csr = np.lexsort((gcol, grow))
        grow, gcol, nvals = grow[csr], gcol[csr], nvals[csr]
    parts = [None] * world
    dist.all_gather_object(parts, (v, f, lo, hi, grow, gcol, nvals, out.cpu().numpy()))
    first = sorted([p for p in parts if p[1] == 0], key=lambda p: p[0])     # one feature slice holds the whole graph once
    g_rows, g_cols = np.concatenate([p[4] for p in first]), np.concatenate([p[5] for p in first])
    g_vals = np.concatenate([p[6] for p in first])
    got_all = np.zeros((n, C), dtype=np.float32)
    for pv_, pf_, lo_, hi_, _, _, _, o in parts:
        got_all[lo_:hi_, pf_ * cs:(pf_ + 1) * cs] = o
    assert len(g_rows) == sg.nnz_global
    for p in parts:                                                # every feature slice of a vertex block holds the same shard
        twin = [q for q in first if q[0] == p[0]][0]
        assert np.array_equal(p[4], twin[4]) and np.array_equal(p[5], twin[5]) and np.array_equal(p[6], twin[6])
    if mode == "slices":
        raw_coo, raw_vals = coo, vals
    else:                                                          # generator: unit weights on the union of the shards' patterns
        raw_coo, raw_vals = np.stack([g_rows, g_cols], 1), np.ones(len(g_rows), dtype=np.float32)
        key = g_rows * n + g_cols
        assert len(np.unique(key)) == len(key) and set(key.tolist()) == set((g_cols * n + g_rows).tolist())   # symmetric, no dups
        assert (g_rows != g_cols).all()
    ai, av = orc.get_adjacency(raw_coo, raw_vals, (n, n))
    _, _, want_vals = orc.coo_to_csr_coalesced(ai, av, (n, n))
    np.testing.assert_allclose(g_vals, want_vals, rtol=2e-6)       # normalised shard values == single-process normalisation
    want = orc.appnp_propagate(raw_coo, raw_vals, (n, n), H0_full, a=a, iterations=K)
    np.testing.assert_allclose(got_all, want, rtol=1e-4, atol=1e-5)
    assert (got_all.argmax(1) == want.argmax(1)).all()
    if on_gpu:                                                     # against ONE GPU holding the whole graph: same summation order
        import gnntf
        whole = gnntf.normalize(gnntf.DeviceGraph(gnntf.SparseCOO(raw_coo, raw_vals, (n, n)), device=dev), "symmetric")
        single = gnntf.appnp_propagate(whole, torch.from_numpy(H0_full).to(dev), a, K).cpu().numpy()
        tol = (1e-5, 1e-6) if sg.row_order is not None else (1e-6, 1e-7)    # relabelling changes the summation order
        np.testing.assert_allclose(got_all, single, rtol=tol[0], atol=tol[1])
    if rank == 0:
        print("OK", mode, "world", world, "grid", f"{pv}x{pf}", "nnz", sg.nnz_global, "halo", sg.n_low + sg.n_high, "kernel", sg.graph.last_kernel())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
