"""Worker of tests/test_sharded_cpu.py: one gloo rank of the vertex-partitioned propagation.
The checker backend below stands in for libgnx.so on CPU ranks (tests may use the oracle; the
product backend is gnntf.sharded.NativeBackend).

    dist_worker.py MODE [cpu|cuda] [cover|pull],[split|whole],CHUNKS
"""
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
        return _G(*orc.coo_to_csr_coalesced(idx.numpy().reshape(-1, 2), vals.numpy(), shape), shape)

    def graph_from_csr(self, rowptr, colidx, vals, shape):
        return _G(rowptr.numpy(), colidx.numpy(), vals.numpy(), shape)

    def csr_arrays(self, g, with_rows=False):
        out = (torch.from_numpy(g.rowptr), torch.from_numpy(g.colidx), torch.from_numpy(g.vals))
        if with_rows:
            out += (torch.from_numpy(np.repeat(np.arange(g.n_rows), np.diff(g.rowptr)).astype(np.int32)),)
        return out

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

    def _product(self, g, vals, X):
        m = sp.csr_matrix((g.vals if vals is None else vals.numpy(), g.colidx, g.rowptr), shape=g.shape)
        return m @ X.numpy()

    def spmm_mix(self, g, vals, X, H0, beta, alpha, out, out_rows=None, rows=None, skip_empty=False):
        prod = self._product(g, vals, X) * np.float32(beta)
        keep = torch.from_numpy(np.diff(g.rowptr) > 0) if skip_empty else torch.ones(g.n_rows, dtype=torch.bool)
        if rows is not None:
            r = rows.long()[keep]
            out[r] = torch.from_numpy(prod)[keep] + H0[r] * np.float32(alpha)
            return
        if skip_empty:
            assert out_rows is None
            out[keep] = torch.from_numpy(prod)[keep] + H0[keep] * np.float32(alpha)
            return
        res = torch.from_numpy(prod + H0.numpy() * np.float32(alpha))
        if out_rows is None:
            out.copy_(res)
        else:
            out[out_rows.long()] = res

    def spmm_plain(self, g, X, out):
        out.copy_(torch.from_numpy(self._product(g, None, X)))

    def gather_rows(self, X, idx):
        return X[idx].contiguous()

    # ---- the exchange plan: the layout rules of gnx_halo_plan_create restated, packing by numpy -------------------------
    def halo_plan(self, rank, n_local, recv_pull, recv_push, send_pull, send_push, pull_src, push_graph):
        P = len(recv_pull)
        plan = type("Plan", (), {})()
        at, plan.recv_row0 = 0, []
        for q in range(P):
            if q == rank:
                plan.local_row0 = at
                at += n_local
            plan.recv_row0.append(at)
            at += recv_pull[q] + recv_push[q]
        plan.n_buf, plan.n_send_pull = at, sum(send_pull)
        plan.n_send = plan.n_send_pull + sum(send_push)
        plan.send_pull_row0 = [sum(send_pull[:q]) for q in range(P)]
        plan.send_push_row0 = [plan.n_send_pull + sum(send_push[:q]) for q in range(P)]
        plan.n_local, plan.pull_src, plan.push_graph = n_local, pull_src.long(), push_graph
        return plan

    def halo_pack(self, plan, part, buf, send):
        local = buf[plan.local_row0:plan.local_row0 + plan.n_local]
        if part != "push" and plan.n_send_pull:
            send[:plan.n_send_pull] = local[plan.pull_src]
        if part != "pull" and plan.n_send > plan.n_send_pull:
            send[plan.n_send_pull:plan.n_send] = torch.from_numpy(self._product(plan.push_graph, None, local))

    # ---- edge dropout on a vertex block: the arithmetic of gnx_graph_colsum_streams / gnx_spmm_dropped, in numpy ----
    def set_block(self, g, row0_global, row0_buf, col_gid):
        g.row0_global, g.row0_buf, g.col_gid = int(row0_global), int(row0_buf), col_gid.numpy().astype(np.int64)

    def _dropped_raw(self, g, p, seed, stream):
        rows = np.repeat(np.arange(g.n_rows), np.diff(g.rowptr))
        gid = getattr(g, "col_gid", None)
        krow = rows + getattr(g, "row0_global", 0)
        kcol = g.colidx.astype(np.int64) if gid is None else gid[g.colidx]
        keep = orc.hash_u24(seed, stream, krow, kcol, np.zeros(len(rows), dtype=np.int64)) >= orc.dropout_threshold(p)
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return rows, np.where(keep, g.vals * scale, np.float32(0)).astype(np.float32)

    def colsum_streams(self, g, p, seed, first_stream, n_streams):
        out = np.zeros((n_streams, g.n_cols), dtype=np.float32)
        for k in range(n_streams):
            _, v = self._dropped_raw(g, p, seed, first_stream + k)
            np.add.at(out[k], g.colidx, v)
        return torch.from_numpy(out)

    def spmm_dropped(self, g, D, p, seed, stream_id, transposed, X, H0, beta, alpha, out):
        rows, v = self._dropped_raw(g, p, seed, stream_id)
        D = D.numpy()
        w = (D[rows + getattr(g, "row0_buf", 0)] * v) * D[g.colidx]
        m = sp.csr_matrix((w, g.colidx, g.rowptr), shape=g.shape)
        prod = ((m.T if transposed else m) @ X.numpy()) * np.float32(beta)
        if H0 is not None:
            prod = prod + H0.numpy() * np.float32(alpha)
        out.copy_(torch.from_numpy(np.ascontiguousarray(prod, dtype=np.float32)))

    def spmm_dropped_chained(self, g, D, p, seed, stream_id, prescaled, D_next, X, H0, beta, alpha, out):
        rows, v = self._dropped_raw(g, p, seed, stream_id)
        D = D.numpy()
        w = D[rows + getattr(g, "row0_buf", 0)] * v
        if not prescaled:
            w = w * D[g.colidx]
        m = sp.csr_matrix((w, g.colidx, g.rowptr), shape=g.shape)
        res = (m @ X.numpy()) * np.float32(beta) + H0.numpy() * np.float32(alpha)
        if D_next is not None:
            r0 = getattr(g, "row0_buf", 0)
            res = res * D_next.numpy()[r0:r0 + g.n_rows, None]
        out.copy_(torch.from_numpy(np.ascontiguousarray(res, dtype=np.float32)))

    def spmm_t_mix(self, g, X, H0, beta, alpha, out):
        m = sp.csr_matrix((g.vals, g.colidx, g.rowptr), shape=g.shape)
        out.copy_(torch.from_numpy(np.ascontiguousarray((m.T @ X.numpy()) * np.float32(beta) + H0.numpy() * np.float32(alpha), dtype=np.float32)))


def train_on_blocks(rank, world, dev, backend, graph_dropout=0.0):
    """architecture.train() with every rank holding ONE vertex block (ShardedPPRLoop + SummedGradients +
    BlockNodeClassification) against single-process dense float64 training of the same model: same parameters afterwards.
    graph_dropout > 0: every training forward drops + re-normalises the edges per iteration (masks by global ids, streams
    numbered as Layered._next_mask_stream does); the reference rebuilds the same dropped adjacencies from the oracle."""
    import gnntf
    gnntf.set_default_device(dev)
    gnntf.set_seed(33)
    n, F, hidden, classes, K, a, epochs = 600, 10, 8, 7, 6, 0.1, 8      # (an odd class count)
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 5000, seed=8)
    rng = np.random.default_rng(8)
    X = rng.standard_normal((n, F)).astype(np.float32)
    labels = rng.integers(0, classes, size=n)
    train_ids, valid_ids = np.arange(0, 240), np.arange(240, 420)
    bounds = sharded.uniform_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
    sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend)
    model = gnntf.Trainable(torch.from_numpy(X[lo:hi]).to(dev))
    model.add(gnntf.Dense(hidden, activation=gnntf.relu))
    head = model.add(gnntf.Dense(classes, regularize=False))
    if graph_dropout:
        dg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend,
                                  edge_dropout=True)
        model.add(sharded.ShardedPPRLoop(head, sg, a, K, graph_dropout=graph_dropout, dropout_graph=dg))
    else:
        model.add(sharded.ShardedPPRLoop(head, sg, a, K))
    local = lambda ids: ids[(ids >= lo) & (ids < hi)]
    tasks = [sharded.BlockNodeClassification(list(local(ids) - lo), labels[local(ids)], sg.comm) for ids in (train_ids, valid_ids)]
    torch.manual_seed(21)                                          # reset() draws the same initial weights on every rank
    # train()'s DEFAULT regularization (5e-4): SummedGradients.replicas keeps the summed weight decay at one process's
    model.train(train=tasks[0], valid=tasks[1], epochs=epochs, patience=50,
                optimizer=lambda params: sharded.SummedGradients(torch.optim.Adam(params, lr=0.01, eps=1e-7), sg.comm))
    got = [v.var.detach().cpu().numpy().astype(np.float64) for v in model.vars()]
    accuracy = model.evaluate(tasks[1])
    # ---- the same training in one process, dense float64 ----------------------------------------------------------------
    torch.manual_seed(21)
    ref = gnntf.Layered((n, F))
    ref.add(gnntf.Dense(hidden, activation=gnntf.relu)); ref.add(gnntf.Dense(classes, regularize=False))
    ref.reset()
    W1, b1, W2, b2 = [torch.tensor(v.var.detach().cpu().numpy().astype(np.float64), requires_grad=True) for v in ref.vars()]
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    A = torch.from_numpy(orc.to_dense(ai, av, shape, dtype=np.float64))
    Xt, yt = torch.from_numpy(X.astype(np.float64)), torch.from_numpy(labels)
    opt = torch.optim.Adam([W1, b1, W2, b2], lr=0.01, eps=1e-7)

    def dropped(stream):
        di, dv = orc.get_adjacency(coo, vals, shape, graph_dropout=graph_dropout, training=True, seed=33, stream=stream, dtype=np.float64)
        return torch.from_numpy(orc.to_dense(di, dv, shape, dtype=np.float64))

    def forward(first=None):
        H0 = torch.relu(Xt @ W1 + b1) @ W2 + b2
        H = H0
        for k in range(K):
            H = ((A if first is None else dropped(first + k)) @ H) * (1 - a) + H0 * a
        return H
    best, best_params = float("inf"), None
    for epoch in range(epochs):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(forward(epoch * K if graph_dropout else None)[train_ids], yt[train_ids]) + 5e-4 * ((W1 ** 2).sum() + (b1 ** 2).sum()) / 2
        loss.backward()
        opt.step()
        with torch.no_grad():
            out = forward()
            v = float(torch.nn.functional.cross_entropy(out[valid_ids], yt[valid_ids]))
        if v < best:
            best, best_params = v, [p.detach().clone().numpy() for p in (W1, b1, W2, b2)]
    for name, g_, w_ in zip(("W1", "b1", "W2", "b2"), got, best_params):
        np.testing.assert_allclose(g_, w_, rtol=2e-3, atol=2e-5, err_msg=name)
    with torch.no_grad():
        for p, q in zip((W1, b1, W2, b2), best_params):
            p.copy_(torch.from_numpy(q))
        want_acc = float((forward()[valid_ids].argmax(1) == yt[valid_ids]).double().mean())
    assert abs(accuracy - want_acc) < 0.02, (accuracy, want_acc)
    gnntf.set_default_device(None)
    if rank == 0:
        print("OK train_dropout" if graph_dropout else "OK train", "world", world, "valid accuracy", round(accuracy, 3))
    dist.barrier()
    dist.destroy_process_group()


def train_gcn_on_blocks(rank, world, dev, backend):
    """A 2-layer GCN (gcn.py:108-113: relu on both layers) whose aggregation runs over vertex blocks (ShardedGCNLayer), trained
    with SummedGradients + BlockNodeClassification, against single-process dense float64 training of the same model."""
    import gnntf
    gnntf.set_default_device(dev)
    n, F, hidden, classes, epochs = 500, 12, 16, 5, 6
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 4000, seed=9)
    rng = np.random.default_rng(9)
    X = rng.standard_normal((n, F)).astype(np.float32)
    labels = rng.integers(0, classes, size=n)
    train_ids, valid_ids = np.arange(0, 200), np.arange(200, 350)
    bounds = sharded.uniform_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
    sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend)
    model = gnntf.Trainable(torch.from_numpy(X[lo:hi]).to(dev))
    model.add(sharded.ShardedGCNLayer(sg, hidden))
    model.add(sharded.ShardedGCNLayer(sg, classes))
    local = lambda ids: ids[(ids >= lo) & (ids < hi)]
    tasks = [sharded.BlockNodeClassification(list(local(ids) - lo), labels[local(ids)], sg.comm) for ids in (train_ids, valid_ids)]
    torch.manual_seed(5)
    model.train(train=tasks[0], valid=tasks[1], epochs=epochs, patience=50, regularization=5e-4,
                optimizer=lambda params: sharded.SummedGradients(torch.optim.Adam(params, lr=0.01, eps=1e-7), sg.comm))
    got = [v.var.detach().cpu().numpy().astype(np.float64) for v in model.vars()]
    # ---- one process, dense float64 -------------------------------------------------------------------------------------
    torch.manual_seed(5)
    ref = gnntf.Layered((n, F))
    ref.add(gnntf.Dense(hidden)); ref.add(gnntf.Dense(classes))          # same shapes and initialisers as the GCN layers' variables
    ref.reset()
    W1, b1, W2, b2 = [torch.tensor(v.var.detach().cpu().numpy().astype(np.float64), requires_grad=True) for v in ref.vars()]
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    A = torch.from_numpy(orc.to_dense(ai, av, shape, dtype=np.float64))
    Xt, yt = torch.from_numpy(X.astype(np.float64)), torch.from_numpy(labels)
    opt = torch.optim.Adam([W1, b1, W2, b2], lr=0.01, eps=1e-7)
    forward = lambda: torch.relu((A @ torch.relu((A @ Xt) @ W1 + b1)) @ W2 + b2)
    best, best_params = float("inf"), None
    for _ in range(epochs):
        opt.zero_grad()
        l2 = sum((p ** 2).sum() for p in (W1, b1, W2, b2)) / 2
        (torch.nn.functional.cross_entropy(forward()[train_ids], yt[train_ids]) + 5e-4 * l2).backward()
        opt.step()
        with torch.no_grad():
            v = float(torch.nn.functional.cross_entropy(forward()[valid_ids], yt[valid_ids]))
        if v < best:
            best, best_params = v, [p.detach().clone().numpy() for p in (W1, b1, W2, b2)]
    for name, g_, w_ in zip(("W1", "b1", "W2", "b2"), got, best_params):
        np.testing.assert_allclose(g_, w_, rtol=2e-3, atol=2e-5, err_msg=name)
    gnntf.set_default_device(None)
    if rank == 0:
        print("OK gcn world", world)
    dist.barrier()
    dist.destroy_process_group()


def train_gcnii_on_blocks(rank, world, dev, backend):
    """Dense -> 3 GCNII layers (gcn.py:7-27; aggregation over vertex blocks, ShardedGCNIILayer) -> Dense, trained on blocks, against
    single-process dense float64 training of the same model."""
    import math
    import gnntf
    gnntf.set_default_device(dev)
    n, F, hidden, classes, layers, a, l, epochs = 480, 10, 16, 4, 3, 0.1, 0.5, 6
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 3600, seed=4)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, F)).astype(np.float32)
    labels = rng.integers(0, classes, size=n)
    train_ids, valid_ids = np.arange(0, 200), np.arange(200, 340)
    bounds = sharded.uniform_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
    sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend)
    model = gnntf.Trainable(torch.from_numpy(X[lo:hi]).to(dev))
    H0 = model.add(gnntf.Dense(hidden, activation=gnntf.relu))
    for k in range(layers):
        model.add(sharded.ShardedGCNIILayer(sg, H0, a, l, k, activation=gnntf.relu, dropout=0))
    model.add(gnntf.Dense(classes))
    local = lambda ids: ids[(ids >= lo) & (ids < hi)]
    tasks = [sharded.BlockNodeClassification(list(local(ids) - lo), labels[local(ids)], sg.comm) for ids in (train_ids, valid_ids)]
    torch.manual_seed(6)
    model.train(train=tasks[0], valid=tasks[1], epochs=epochs, patience=50, regularization=5e-4,
                optimizer=lambda params: sharded.SummedGradients(torch.optim.Adam(params, lr=0.01, eps=1e-7), sg.comm))
    got = [v.var.detach().cpu().numpy().astype(np.float64) for v in model.vars()]
    # ---- one process, dense float64 -------------------------------------------------------------------------------------
    torch.manual_seed(6)
    ref = gnntf.Layered((n, F))
    ref.add(gnntf.Dense(hidden)); ref.add(gnntf.Dense(classes))
    ref.reset()
    Wi, bi, Wo, bo = [torch.tensor(v.var.detach().cpu().numpy().astype(np.float64), requires_grad=True) for v in ref.vars()]
    Ws = [torch.zeros((hidden, hidden), dtype=torch.float64, requires_grad=True) for _ in range(layers)]
    params = [Wi, bi] + Ws + [Wo, bo]                       # the order model.vars() registers them in
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    A = torch.from_numpy(orc.to_dense(ai, av, shape, dtype=np.float64))
    Xt, yt = torch.from_numpy(X.astype(np.float64)), torch.from_numpy(labels)
    opt = torch.optim.Adam(params, lr=0.01, eps=1e-7)

    def forward():
        h0 = torch.relu(Xt @ Wi + bi)
        h = h0
        for k, W in enumerate(Ws):
            b = math.log1p(l / (k + 1))
            h = torch.relu(((1 - a) * (A @ h) + a * h0) @ ((1 - b) * torch.eye(hidden, dtype=torch.float64) + b * W))
        return h @ Wo + bo
    best, best_params = float("inf"), None
    for _ in range(epochs):
        opt.zero_grad()
        l2 = sum((p ** 2).sum() for p in params) / 2
        (torch.nn.functional.cross_entropy(forward()[train_ids], yt[train_ids]) + 5e-4 * l2).backward()
        opt.step()
        with torch.no_grad():
            v = float(torch.nn.functional.cross_entropy(forward()[valid_ids], yt[valid_ids]))
        if v < best:
            best, best_params = v, [p.detach().clone().numpy() for p in params]
    assert len(got) == len(best_params)
    for i, (g_, w_) in enumerate(zip(got, best_params)):
        assert g_.shape == w_.shape, (i, g_.shape, w_.shape)
        np.testing.assert_allclose(g_, w_, rtol=2e-3, atol=2e-5, err_msg="parameter %d" % i)
    assert max(float(np.abs(w).max()) for w in got[2:2 + layers]) > 1e-3      # the GCNII transforms did move away from zero
    gnntf.set_default_device(None)
    if rank == 0:
        print("OK gcnii world", world)
    dist.barrier()
    dist.destroy_process_group()


def dropout_on_blocks(rank, world, dev, backend, directed):
    """Training-mode propagation with per-iteration edge dropout on vertex blocks: the masks are keyed by global (row, col),
    so forward and backward must equal the single-process oracle with the same seed, whatever the partition."""
    n, C, K, a, p, seed, first = 700, 9, 5, 0.1, 0.5, 12345, 7
    if directed:
        rng = np.random.default_rng(11)
        key = np.unique(rng.integers(n, size=6000) * n + (rng.random(6000) ** 3 * n).astype(np.int64))
        coo = np.stack([key // n, key % n], 1)
        vals = rng.random(len(key)).astype(np.float32) + 0.5
    else:
        coo, vals, _ = graphs.rmat_symmetric_coo(n, 7000, seed=5)
    rng = np.random.default_rng(2)
    H0_full = rng.uniform(-1, 1, size=(n, C)).astype(np.float32)
    G_full = rng.uniform(-1, 1, size=(n, C)).astype(np.float32)
    bounds = sharded.uniform_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
    sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend,
                              edge_dropout=True)
    assert sg.cover == "pull" and (world == 1 or not sg.split_rows)
    scales = sg.dropped_scales(p, seed, first, K)
    H0 = torch.from_numpy(H0_full[lo:hi].copy()).to(dev)
    out = sg.propagate_dropped(H0, a, K, p, seed, first, scales)
    gH0 = sg.propagate_dropped_backward(torch.from_numpy(G_full[lo:hi].copy()).to(dev), a, K, p, seed, first, scales)
    for chunks in (2, 3):                                          # column chunks (exchange of one under the SpMM of the next): same columns, same sums
        out_c = sg.propagate_dropped(H0, a, K, p, seed, first, scales, chunks=chunks)
        gH0_c = sg.propagate_dropped_backward(torch.from_numpy(G_full[lo:hi].copy()).to(dev), a, K, p, seed, first, scales, chunks=chunks)
        np.testing.assert_allclose(out_c.cpu().numpy(), out.cpu().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(gH0_c.cpu().numpy(), gH0.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # ---- single process, the oracle's get_adjacency in training mode with the same (seed, stream) per iteration ---------
    adjs = [orc.get_adjacency(coo, vals, (n, n), graph_dropout=p, training=True, seed=seed, stream=first + k) for k in range(K)]
    H = H0_full.astype(np.float64)
    for ai, av in adjs:
        H = orc.ppr_iteration(ai, av.astype(np.float64), (n, n), H, H0_full.astype(np.float64), a)
    g, want_g = G_full.astype(np.float64), np.zeros((n, C))
    for ai, av in reversed(adjs):
        want_g += a * g
        g = (1 - a) * (sp.csr_matrix((av.astype(np.float64), (ai[:, 0], ai[:, 1])), shape=(n, n)).T @ g)
    want_g += g
    dropped = [float((av == 0).mean()) for _, av in adjs]
    assert all(0.4 < d < 0.6 for d in dropped), dropped            # the masks really drop about half the entries
    if world > 1:                                                  # ... and the degree scales are the global ones
        colsum = np.zeros(n, dtype=np.float64)
        ai, av = orc.get_adjacency(coo, vals, (n, n), graph_dropout=p, normalized="none", training=True, seed=seed, stream=first)
        np.add.at(colsum, ai[:, 1], av)
        want_D = np.where(colsum > 0, 1.0 / np.sqrt(np.maximum(colsum, 1e-30)), 0.0)
        np.testing.assert_allclose(scales[0].cpu().numpy(), want_D[sg.col_gid.cpu().numpy()], rtol=2e-6)
    np.testing.assert_allclose(out.cpu().numpy(), H[lo:hi], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gH0.cpu().numpy(), want_g[lo:hi], rtol=2e-4, atol=2e-5)
    if dev.type == "cuda":                                         # and against ONE GPU holding the whole graph (same masks)
        import gnntf
        from gnntf import sparse
        whole = gnntf.DeviceGraph(gnntf.SparseCOO(coo, vals, (n, n)), device=dev)
        D1 = sparse.dropped_degree_scales(whole, p, seed, first, K)
        Hf = torch.from_numpy(H0_full).to(dev).requires_grad_(True)
        single = sparse.ppr_loop(lambda k, bwd=False: sparse.dropped_adjacency(whole, p, seed, first + k, D=D1[k]), Hf, a, K)
        single.backward(torch.from_numpy(G_full).to(dev))
        np.testing.assert_allclose(out.cpu().numpy(), single.detach().cpu().numpy()[lo:hi], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gH0.cpu().numpy(), Hf.grad.cpu().numpy()[lo:hi], rtol=1e-5, atol=1e-6)
    if rank == 0:
        print("OK dropout_directed" if directed else "OK dropout", "world", world, "dropped", [round(d, 3) for d in dropped])
    dist.barrier()
    dist.destroy_process_group()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mode = sys.argv[1]
    if mode == "train":
        on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"
        return train_on_blocks(rank, world, torch.device("cuda:0" if on_gpu else "cpu"), None if on_gpu else OracleBackend())
    if mode == "gcnii":
        on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"
        return train_gcnii_on_blocks(rank, world, torch.device("cuda:0" if on_gpu else "cpu"), None if on_gpu else OracleBackend())
    if mode == "gcn":
        on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"
        return train_gcn_on_blocks(rank, world, torch.device("cuda:0" if on_gpu else "cpu"), None if on_gpu else OracleBackend())
    if mode == "train_dropout":
        on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"
        return train_on_blocks(rank, world, torch.device("cuda:0" if on_gpu else "cpu"), None if on_gpu else OracleBackend(), 0.5)
    if mode in ("dropout", "dropout_directed"):
        on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"
        return dropout_on_blocks(rank, world, torch.device("cuda:0" if on_gpu else "cpu"), None if on_gpu else OracleBackend(),
                                 mode == "dropout_directed")
    on_gpu = len(sys.argv) > 2 and sys.argv[2] == "cuda"          # GPU box: every rank shares cuda:0, libgnx.so backend
    opts = (sys.argv[3] if len(sys.argv) > 3 else "cover,split,2").split(",")
    options = dict(cover=opts[0].split("@")[0], split_rows="always" if opts[1] == "split" else False, chunks=int(opts[2]), keep_entries=True)
    if "@" in opts[0]:                                             # "cover@0.25": the weighted cover (cover_push_mask's push_weight)
        options["push_weight"] = float(opts[0].split("@")[1])
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
        sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend,
                                  **options)
        v, f = rank, 0
    elif mode == "directed":                                       # asymmetric pattern: column sums != row sums, send rows need not be boundary rows
        n = 640
        rng = np.random.default_rng(11)
        coo = np.stack([rng.integers(n, size=5000), (rng.random(5000) ** 3 * n).astype(np.int64)], 1)
        vals = (rng.random(5000).astype(np.float32) + 0.5)
        bounds = sharded.uniform_bounds(n, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
        sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]).to(dev), torch.from_numpy(vals[mine]).to(dev), bounds, backend=backend,
                                  **options)
        v, f = rank, 0
    else:
        if mode.startswith("grid"):
            pv, pf = (int(x) for x in mode[4:].split("x"))
        if mode.startswith("blocks"):                              # the bench's strong-scaling generator (one global graph)
            sg, _, (v, f, pv, pf) = sharded.build_rmat_blocks(1500, 16000, seed=1, device=dev, backend=backend,
                                                              replicate=mode == "blocks_replicated", **options)
        else:                                                      # the weak-scaling generator on a pv x pf grid
            sg, _, (v, f, pv, pf) = sharded.build_rmat_shard(500, 6000, seed=1, device=dev, backend=backend, grid=(pv, pf), **options)
        n, lo, hi = sg.n_global, sg.lo, sg.hi
    cs = C // pf                                                   # this rank's feature slice
    H0_full = np.random.default_rng(1).uniform(-1, 1, size=(n, C)).astype(np.float32)
    H0 = torch.from_numpy(H0_full[lo:hi, f * cs:(f + 1) * cs].copy()).to(dev)
    state = sg.make_state(H0)
    out = sg.propagate(state, a, K).clone()
    again = sg.propagate(state, a, K).clone()
    assert torch.equal(out, again), "propagate is not repeatable"
    zero = sg.propagate(state, a, 0)
    assert torch.equal(zero, H0), "K = 0 must return H0 in the caller's order"
    if sg.world > 1:          # pulled rows sent ahead of the pushed partial sums: two messages per peer, the same bytes in the same places
        early = sg.propagate(state, a, K, early_pull=True)
        assert torch.equal(early, out), "early_pull changed the result"
    if mode not in ("slices", "directed"):                          # symmetric unit-weight graphs: bench.py's in-run self check
        assert sg.fixed_point_error(sg.make_state(H0.clone()), a, K) < 1e-5      # (overwrites the H0 it is given)
    one = sg.propagate(sg.make_state(H0, chunks=1) if sg.world > 1 else state, a, K)
    np.testing.assert_allclose(one.cpu().numpy(), out.cpu().numpy(), rtol=1e-6, atol=1e-7)   # chunking does not change a column's sums

    # plan invariants
    assert sg.world == pv and sg.recv_counts[sg.rank] == 0 and sg.send_counts[sg.rank] == 0
    assert sg.n_buf == sg.n_local + sum(sg.recv_counts)
    assert sg.stats["pull_rows"] + sg.stats["push_rows"] == sum(sg.recv_counts) and sg.stats["send_rows"] == sum(sg.send_counts)
    assert sg.stats["pull_rows"] + sg.stats["push_rows"] <= sg.stats["pull_only_rows"]       # a cover never moves more rows than the halo
    if options["cover"] == "pull":
        assert sg.stats["push_rows"] == 0 and sg.stats["pull_rows"] == sg.stats["pull_only_rows"]
    if pv == 1:
        assert sum(sg.recv_counts) == 0                            # feature slices alone: no halo, no exchange
        if not mode.startswith("blocks"):
            assert sg.row_order is not None                        # ... and the weak generator stores the shard degree-relabelled
    plans = [None] * world
    dist.all_gather_object(plans, (v, f, sg.send_counts, sg.recv_counts))
    for pv_, pf_, sends, recvs in plans:                           # what I send to q is what q expects from me
        if pf_ == f:
            assert sends[v] == sg.recv_counts[pv_] and recvs[v] == sg.send_counts[pv_]

    # every rank's entries (global ids, normalised values) against the single-process normalisation
    grow, gcol, nvals, pushed = (None if t is None else t.cpu().numpy() for t in sg.entries)
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
    if mode in ("slices", "directed"):
        raw_coo, raw_vals = coo, vals
    else:                                                          # generators: unit weights on the union of the shards' patterns
        raw_coo, raw_vals = np.stack([g_rows, g_cols], 1), np.ones(len(g_rows), dtype=np.float32)
        key = g_rows * n + g_cols
        assert len(np.unique(key)) == len(key) and set(key.tolist()) == set((g_cols * n + g_rows).tolist())   # symmetric, no dups
        assert (g_rows != g_cols).all()
        if mode.startswith("blocks"):
            assert len(key) == 16000                               # exactly the requested number of stored entries
    ai, av = orc.get_adjacency(raw_coo, raw_vals, (n, n))
    _, _, want_vals = orc.coo_to_csr_coalesced(ai, av, (n, n))
    np.testing.assert_allclose(g_vals, want_vals, rtol=2e-6)       # normalised shard values == single-process normalisation
    want = orc.appnp_propagate(raw_coo, raw_vals, (n, n), H0_full, a=a, iterations=K)
    np.testing.assert_allclose(got_all, want, rtol=1e-4, atol=1e-5)
    assert (got_all.argmax(1) == want.argmax(1)).all()
    if on_gpu:                                                     # against ONE GPU holding the whole graph
        import gnntf
        whole = gnntf.normalize(gnntf.DeviceGraph(gnntf.SparseCOO(raw_coo, raw_vals, (n, n)), device=dev), "symmetric")
        single = gnntf.appnp_propagate(whole, torch.from_numpy(H0_full).to(dev), a, K).cpu().numpy()
        # pull-only plans keep the one-GPU summation order; pushed partial sums / relabelling change it
        exact_order = options["cover"] == "pull" and sg.row_order is None
        tol = (1e-6, 1e-7) if exact_order else (1e-5, 1e-6)
        np.testing.assert_allclose(got_all, single, rtol=tol[0], atol=tol[1])
    # the layer API on a vertex block (eval mode): Dense layers act row by row, ShardedPPRLoop propagates with the other ranks
    import gnntf
    gnntf.set_default_device(dev)
    torch.manual_seed(7)
    Wd = (torch.rand(C, 5) - 0.5).to(dev)
    local = gnntf.Trainable(torch.from_numpy(H0_full[lo:hi]).to(dev))
    head = local.add(gnntf.Dense(5, bias=False))
    local.add(sharded.ShardedPPRLoop(head, sg, a, K)) if pf == 1 else None
    if pf == 1:
        local.vars()[0].assign(Wd)
        local.training_mode(False)
        pred = local.predict(gnntf.NodeClassification(list(range(sg.n_local))))
        want_all = orc.appnp_propagate(raw_coo, raw_vals, (n, n), H0_full @ Wd.cpu().numpy(), a=a, iterations=K)
        top2 = np.sort(want_all[lo:hi], axis=1)
        clear = (top2[:, -1] - top2[:, -2]) > 1e-4                     # rows whose best class is not a float tie
        assert (pred.cpu().numpy()[clear] == want_all[lo:hi].argmax(1)[clear]).all() and clear.mean() > 0.9
    gnntf.set_default_device(None)
    if rank == 0:
        print("OK", mode, "world", world, "grid", f"{pv}x{pf}", "nnz", sg.nnz_global, "stats", sg.halo_stats() if False else sg.stats,
              "kernel", sg.graph.last_kernel())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
