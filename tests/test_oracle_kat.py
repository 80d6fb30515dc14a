"""Pins for the CPU oracle: closed forms, hand graphs, independent
re-derivations (scipy CSR, dense float64, torch.sparse) and the committed golden vectors.
The reference ships no vectors of its own (parity unpinned); these are the pins we have."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import graphs
from oracle import gnntf_oracle as orc


def dense_sym_norm(A):
    d = A.sum(axis=0)
    D = np.where(d > 0, 1 / np.sqrt(np.where(d > 0, d, 1)), 0)
    return D[:, None] * A * D[None, :]


# ---- KAT-1: closed form of K power-iteration steps -----------------------------------------
@pytest.mark.parametrize("K", [1, 3, 10])
def test_kat1_closed_form(K):
    coo, vals, shape = graphs.random_coo(150, 150, 1200, seed=5, weighted=True)
    rng = np.random.default_rng(0)
    H0 = rng.standard_normal((150, 5))
    A_hat = dense_sym_norm(orc.to_dense(coo, vals, shape))
    want = orc.appnp_closed_form(A_hat, H0, 0.1, K)
    got = orc.appnp_propagate(coo, vals, shape, H0, a=0.1, iterations=K, dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    got32 = orc.appnp_propagate(coo, vals, shape, H0, a=0.1, iterations=K, dtype=np.float32)
    np.testing.assert_allclose(got32, want, rtol=1e-4, atol=1e-5)


def test_kat1_fixed_point():
    """K -> infinity converges to a (I - (1-a) A_hat)^-1 H0."""
    coo, vals, shape = graphs.rmat_symmetric_coo(120, 600, seed=2)
    H0 = np.random.default_rng(1).standard_normal((120, 3))
    A_hat = dense_sym_norm(orc.to_dense(coo, vals, shape))
    want = 0.1 * np.linalg.solve(np.eye(120) - 0.9 * A_hat, H0)
    got = orc.appnp_propagate(coo, vals, shape, H0, a=0.1, iterations=400, dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11)


# ---- KAT-2: hand graphs ------------------------------------------------------------------------
def test_kat2_path_p3():
    idx, vals, shape = orc.graph2adj([0, 1, 2], [(0, 1), (1, 2)])
    assert idx.tolist() == [[0, 1], [1, 2], [1, 0], [2, 1]] and vals.tolist() == [1, 1, 1, 1]
    ai, av = orc.get_adjacency(idx, vals, shape, training=False, dtype=np.float64)
    s = 1 / np.sqrt(2)
    np.testing.assert_allclose(orc.to_dense(ai, av, shape), [[0, s, 0], [s, 0, s], [0, s, 0]], atol=1e-15)
    H0 = np.array([[1.0], [0.0], [0.0]])
    out = orc.ppr_iteration(ai, av, shape, H0, H0, a=0.1)
    np.testing.assert_allclose(out, [[0.1], [0.9 * s], [0.0]], atol=1e-15)


def test_kat2_star_and_isolated():
    """Star S5 plus two isolated nodes: divide_no_nan gives the isolated rows 0 weight, so
    after a step they hold a*H0 exactly (gnn.py:41)."""
    nodes = list(range(8))
    edges = [(0, i) for i in range(1, 6)]
    idx, vals, shape = orc.graph2adj(nodes, edges)
    ai, av = orc.get_adjacency(idx, vals, shape, dtype=np.float64)
    assert np.isfinite(av).all()
    A = orc.to_dense(ai, av, shape)
    np.testing.assert_allclose(A[0, 1:6], 1 / np.sqrt(5))
    assert (A[6:] == 0).all() and (A[:, 6:] == 0).all()
    H0 = np.arange(16, dtype=np.float64).reshape(8, 2)
    out = orc.appnp_propagate(idx, vals, shape, H0, a=0.25, iterations=3, dtype=np.float64)
    np.testing.assert_allclose(out[6:], 0.25 * H0[6:], atol=1e-15)


def test_kat2_duplicates_cancel():
    """A DiGraph that already stores both directions is doubled by graph2adj;
    symmetric normalisation cancels the factor 2 exactly."""
    und = [(0, 1), (1, 2), (2, 0), (2, 3)]
    both = und + [(v, u) for u, v in und]
    i1, v1, shape = orc.graph2adj(range(4), und)
    i2, v2, _ = orc.graph2adj(range(4), both)
    assert len(v2) == 2 * len(v1)
    a1 = orc.to_dense(*orc.get_adjacency(i1, v1, shape, dtype=np.float64), shape)
    a2 = orc.to_dense(*orc.get_adjacency(i2, v2, shape, dtype=np.float64), shape)
    np.testing.assert_allclose(a1, a2, rtol=1e-15)


def test_kat2_weights_and_directed():
    """Column sums (axis=0) scale BOTH sides: on a directed graph this differs from the
    textbook row/col degree form (gnn.py:41)."""
    idx, vals, shape = orc.graph2adj(range(3), [(0, 1), (0, 2), (1, 2)], weights=[2.0, 3.0, 4.0], directed=True)
    assert len(vals) == 3
    ai, av = orc.get_adjacency(idx, vals, shape, dtype=np.float64)
    A = orc.to_dense(idx, vals, shape)
    d = A.sum(axis=0)  # [0, 2, 7]
    D = np.array([0, 1 / np.sqrt(2), 1 / np.sqrt(7)])
    np.testing.assert_allclose(orc.to_dense(ai, av, shape), D[:, None] * A * D[None, :], atol=1e-15)
    assert d.tolist() == [0, 2, 7]


# ---- KAT-3: add_eye / normalized options ------------------------------------------------------
@pytest.mark.parametrize("norm", ["symmetric", "bipartite", "none"])
@pytest.mark.parametrize("eye", ["none", "before", "after"])
def test_kat3_options(norm, eye):
    coo, vals, shape = graphs.random_coo(40, 40, 200, seed=7)
    A = orc.to_dense(coo, vals, shape)
    if eye == "before":
        A = A + np.eye(40)
    d = A.sum(axis=0)
    if norm == "symmetric":
        D = np.where(d > 0, 1 / np.sqrt(np.where(d > 0, d, 1)), 0)
        A = D[:, None] * A * D[None, :]
    elif norm == "bipartite":
        D = np.where(d > 0, 1 / np.where(d > 0, d, 1), 0)
        A = D[:, None] * A
    if eye == "after":
        A = A + np.eye(40)
    ai, av = orc.get_adjacency(coo, vals, shape, normalized=norm, add_eye=eye, dtype=np.float64)
    np.testing.assert_allclose(orc.to_dense(ai, av, shape), A, rtol=1e-12, atol=1e-14)


def test_kat3_invalid_normalisation():
    coo, vals, shape = graphs.random_coo(5, 5, 10, seed=1)
    with pytest.raises(Exception, match="Invalid matrix normalization"):
        orc.get_adjacency(coo, vals, shape, normalized="row")


# ---- KAT-4: bipartite rows sum to one on symmetric graphs --------------------------------------
def test_kat4_row_stochastic():
    coo, vals, shape = graphs.rmat_symmetric_coo(200, 1500, seed=4)
    ai, av = orc.get_adjacency(coo, vals, shape, normalized="bipartite", dtype=np.float64)
    rows = orc.to_dense(ai, av, shape).sum(axis=1)
    deg = orc.to_dense(coo, vals, shape).sum(axis=1)
    np.testing.assert_allclose(rows[deg > 0], 1.0, rtol=1e-12)
    assert (rows[deg == 0] == 0).all()


# ---- independent re-derivations ------------------------------------------------------------------
def test_spmm_vs_scipy_and_torch():
    coo, vals, shape = graphs.random_coo(300, 300, 4000, seed=9)
    H = np.random.default_rng(3).standard_normal((300, 17)).astype(np.float32)
    got = orc.sparse_dense_matmul(coo, vals, shape, H)
    want = sp.coo_matrix((vals.astype(np.float64), (coo[:, 0], coo[:, 1])), shape=shape).tocsr() @ H.astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    t = torch.sparse_coo_tensor(torch.from_numpy(coo.T.copy()), torch.from_numpy(vals), shape).coalesce()
    np.testing.assert_allclose(got, torch.sparse.mm(t, torch.from_numpy(H)).numpy(), rtol=1e-5, atol=1e-5)


def test_csr_helper_matches_coo():
    coo, vals, shape = graphs.random_coo(64, 80, 900, seed=11)
    rowptr, colidx, cvals = orc.coo_to_csr_coalesced(coo, vals, shape)
    m = sp.csr_matrix((cvals, colidx, rowptr), shape=shape)
    np.testing.assert_allclose(m.toarray(), orc.to_dense(coo, vals, shape, np.float32), rtol=1e-6)
    assert rowptr[-1] == len(colidx) < len(vals)  # duplicates were summed


def test_backward_is_transpose():
    coo, vals, shape = graphs.random_coo(50, 50, 300, seed=13)
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    g = np.random.default_rng(0).standard_normal((50, 4))
    gH, gH0 = orc.ppr_iteration_backward(ai, av, shape, g, a=0.1)
    np.testing.assert_allclose(gH, 0.9 * orc.to_dense(ai, av, shape).T @ g, rtol=1e-12)
    np.testing.assert_allclose(gH0, 0.1 * g)


# ---- dropout: distribution + scaling (layered.py:47-50) -------------------------------------------
def test_dropout_rng_statistics():
    coo, vals, shape = graphs.random_coo(400, 400, 60000, seed=17, weighted=False, dup_frac=0.3)
    for p in (0.25, 0.5, 0.8):
        keep = orc.keep_mask(coo, p, seed=99, stream=3)
        assert abs(keep.mean() - (1 - p)) < 0.01
        dropped = orc.sparse_dropout(coo, vals, p, training=True, seed=99, stream=3)
        np.testing.assert_allclose(np.unique(dropped), [0, np.float32(1) / (np.float32(1) - np.float32(p))], rtol=1e-7)
    a, b = orc.keep_mask(coo, 0.5, 99, 3), orc.keep_mask(coo, 0.5, 99, 4)
    assert 0.45 < (a == b).mean() < 0.55  # streams are independent
    assert (orc.keep_mask(coo, 0.5, 99, 3) == a).all()  # and reproducible
    assert orc.sparse_dropout(coo, vals, 0.5, training=False) is vals
    assert orc.sparse_dropout(coo, vals, 0, training=True) is vals


def test_duplicate_rank():
    idx = np.array([[1, 2], [0, 0], [1, 2], [1, 2], [0, 0], [3, 1]])
    assert orc.duplicate_rank(idx).tolist() == [0, 0, 1, 2, 1, 0]


# ---- heads -------------------------------------------------------------------------------------------
def test_node_classification_head():
    logits = np.array([[1.0, 2.0, 0.5], [3.0, -1.0, 0.0], [0.0, 0.0, 1.0]])
    assert orc.node_predict(logits, [2, 0]).tolist() == [2, 1]
    assert orc.node_evaluate(logits, [0, 1, 2], [1, 0, 0]) == pytest.approx(2 / 3)
    want = torch.nn.functional.cross_entropy(torch.tensor(logits), torch.tensor([1, 0, 2])).item()
    assert orc.node_loss(logits, [0, 1, 2], [1, 0, 2]) == pytest.approx(want, rel=1e-12)
    assert orc.l2_loss(np.array([3.0, 4.0])) == 12.5


# ---- golden vectors ------------------------------------------------------------------------------------
def load_cora(golden_dir):
    z = np.load(os.path.join(golden_dir, "cora_shaped_appnp.npz"))
    n, f = int(z["n"]), int(z["f"])
    X = np.zeros((n, f), dtype=np.float32)
    X[z["x_rows"], z["x_cols"]] = 1.0
    X = X / X.sum(axis=1, keepdims=True)
    coo = z["coo"].astype(np.int64)
    weights = [(z["W1"].astype(np.float32), z["b1"].astype(np.float32)), (z["W2"].astype(np.float32), z["b2"].astype(np.float32))]
    return z, coo, np.ones(len(coo), dtype=np.float32), (n, n), X, weights


def test_golden_cora_appnp(golden_dir):
    z, coo, vals, shape, X, weights = load_cora(golden_dir)
    assert len(coo) == 21112 and shape == (2708, 2708) and X.shape == (2708, 1433)
    logits, H0 = orc.appnp_forward_eval(coo, vals, shape, X, weights, a=float(z["a"]), iterations=int(z["iterations"]))
    np.testing.assert_array_equal(H0, z["H0"])
    np.testing.assert_array_equal(logits, z["logits32"])
    np.testing.assert_allclose(logits, z["logits64"], rtol=1e-4, atol=1e-6)
    assert (np.argmax(logits, axis=1) == z["argmax"]).all()


def test_golden_arxiv_gcn(golden_dir):
    z = np.load(os.path.join(golden_dir, "arxiv_mini_gcn.npz"))
    coo = z["coo"].astype(np.int64)
    n = int(z["n"])
    weights = [(z["W1"].astype(np.float32), z["b1"].astype(np.float32)), (z["W2"].astype(np.float32), z["b2"].astype(np.float32))]
    out = orc.gcn_forward_eval(coo, np.ones(len(coo), dtype=np.float32), (n, n), z["X"].astype(np.float32), weights)
    np.testing.assert_array_equal(out, z["out32"])
    np.testing.assert_allclose(out, z["out64"], rtol=1e-4, atol=1e-5)
    assert (out >= 0).all()  # the last GCN layer keeps its relu (gcn.py:78,113)


def test_golden_dropout_masks(golden_dir):
    z = np.load(os.path.join(golden_dir, "dropout_masks.npz"))
    coo, vals, n = z["coo"].astype(np.int64), z["vals"], int(z["n"])
    for stream in (0, 7):
        np.testing.assert_array_equal(orc.keep_mask(coo, float(z["p"]), int(z["seed"]), stream), z[f"keep_{stream}"])
        _, av = orc.get_adjacency(coo, vals, (n, n), graph_dropout=float(z["p"]), training=True, seed=int(z["seed"]), stream=stream)
        np.testing.assert_array_equal(av, z[f"adj_vals_{stream}"])


# ---- property test: random COO (duplicates, empty rows, rectangular) vs scipy in float64 --------------
from hypothesis import given, settings, strategies as st


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 40), st.integers(1, 40), st.integers(0, 300), st.integers(1, 9), st.integers(0, 2 ** 31 - 1))
def test_property_spmm_matches_scipy(n_rows, n_cols, nnz, C, seed):
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(n_rows, size=nnz), rng.integers(n_cols, size=nnz)], axis=1).astype(np.int64)
    vals = rng.standard_normal(nnz)
    H = rng.standard_normal((n_cols, C))
    want = sp.coo_matrix((vals, (idx[:, 0], idx[:, 1])), shape=(n_rows, n_cols)).tocsr() @ H
    np.testing.assert_allclose(orc.sparse_dense_matmul(idx, vals, (n_rows, n_cols), H), want, rtol=1e-10, atol=1e-12)
    rowptr, colidx, cvals = orc.coo_to_csr_coalesced(idx, vals, (n_rows, n_cols))
    np.testing.assert_allclose(sp.csr_matrix((cvals, colidx, rowptr), shape=(n_rows, n_cols)) @ H, want, rtol=1e-10, atol=1e-12)
    assert (np.diff(rowptr) >= 0).all() and all((np.diff(colidx[rowptr[i]:rowptr[i + 1]]) > 0).all() for i in range(n_rows))
