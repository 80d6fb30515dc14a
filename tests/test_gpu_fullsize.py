"""Parity at BASELINE.json's full sizes, on the GPU box: config 3 (arxiv-shaped GCN) against the numpy oracle, and
configs 4 and 5 (RMAT 10M/100M C=256, RMAT 80M/1B C=128) through size-independent closed forms -- no CPU run is
possible at those sizes.  Tolerances: the float32 logits bar of BASELINE.json (rtol 1e-4) or tighter."""
import argparse

import numpy as np
import pytest
import torch

import graphs
from oracle import gnntf_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnntf():
    import gnntf
    gnntf.set_default_device("cuda:0")
    yield gnntf
    gnntf.set_default_device(None)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_config3_arxiv_full_size_gcn_vs_oracle(gnntf):
    """BASELINE config 3 at its real size: N = 169,343 vertices, 1,166,243 directed pairs symmetrised (~2.3M stored entries),
    2-layer GCN 128 -> 64 -> 40 through the layer API, against orc.gcn_forward_eval in float32."""
    n = 169_343
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 1_166_243, seed=1)
    assert 2_200_000 < len(coo) <= 2_332_486
    rng = np.random.default_rng(2)
    X = rng.standard_normal((n, 128)).astype(np.float32)
    small = lambda fi, fo: rng.uniform(-1 / np.sqrt(fo), 1 / np.sqrt(fo), size=(fi, fo)).astype(np.float32)
    weights = [(small(128, 64), rng.uniform(-0.1, 0.1, size=(1, 64)).astype(np.float32)),
               (small(64, 40), rng.uniform(-0.1, 0.1, size=(1, 40)).astype(np.float32))]
    model = gnntf.GCN(gnntf.SparseCOO(coo, vals, shape), X, num_classes=40)
    for layer, (W, b) in zip(model.layers(), weights):
        layer.W.data.copy_(dev(W)); layer.b.data.copy_(dev(b))
    model.training_mode(False)
    with torch.no_grad():
        got = model(model.features).cpu().numpy()
    want = orc.gcn_forward_eval(coo, vals, shape, X, weights, dtype=np.float32)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    decided = np.sort(want, axis=1)[:, -1] - np.sort(want, axis=1)[:, -2] > 1e-5        # rows whose top-2 logits are not a float tie
    assert (got.argmax(1)[decided] == want.argmax(1)[decided]).all() and decided.mean() > 0.5
    assert model.graph.last_kernel() in ("spmm_group16", "spmm_group32")


def eigenvector_check(gnntf, g, adj, C, K=10):
    """H0[i, c] = sqrt(deg_i) * s_c is a fixed point of H <- (1-a) A_hat H + a H0 for D^-1/2 A D^-1/2 with unit weights
    (eigenvalue 1; rows of isolated vertices are 0).  The column factors s_c are all different, so a lane -> column mix-up
    shows; the check is RELATIVE per element, so low-degree rows count as much as hubs."""
    rowptr, colidx, raw = g.csr_arrays()
    assert bool((raw == 1).all())
    deg = (rowptr[1:] - rowptr[:-1]).float()
    del rowptr, colidx, raw
    c = torch.arange(C, device="cuda", dtype=torch.float32)
    s = (0.5 + c / C) * (1 - 2 * (c.long() % 2).float())                     # distinct magnitudes, alternating signs
    H0 = deg.sqrt().unsqueeze(1) * s.unsqueeze(0)
    out = gnntf.appnp_propagate(adj, H0, a=0.1, iterations=K)
    rel = ((out - H0).abs() / H0.abs().clamp_min(1e-30))[deg > 0]
    worst = float(rel.max())
    assert worst <= 2e-5, worst
    assert torch.equal(out[deg == 0], H0[deg == 0])                           # isolated rows: exactly a * 0 + (1-a) * 0
    del out, H0, rel
    # the stored pattern is symmetric: x^T (A y) == (A x)^T y on random vectors
    x = torch.rand(g.n_rows, 8, device="cuda"); y = torch.rand(g.n_rows, 8, device="cuda")
    l = (x.double() * gnntf.spmm(adj, y).double()).sum(); r = (gnntf.spmm(adj, x).double() * y.double()).sum()
    assert abs(float(l - r)) <= 1e-6 * abs(float(l))
    # D^-1 A has unit row sums on the non-isolated rows
    bip = gnntf.normalize(g, "bipartite")
    rows = gnntf.spmm(bip, torch.ones(g.n_rows, 4, device="cuda"))
    assert torch.allclose(rows[deg > 0], torch.ones_like(rows[deg > 0]), rtol=1e-5)
    return worst


def test_config4_full_size_eigenvector(gnntf):
    """BASELINE config 4 at full size (RMAT 10M vertices / 100M stored entries, C=256, K=10)."""
    import bench
    g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), torch.device("cuda:0"))
    assert g.n_rows == 10_000_000 and g.nnz == 100_000_000
    eigenvector_check(gnntf, g, adj, 256)
    assert g.last_kernel() == "spmm_group4"


def test_config5_full_size_eigenvector(gnntf):
    """BASELINE config 5's graph on ONE GPU (RMAT 80M vertices / 1B stored entries, C=128, K=10; ~150 GB of the 288 GB):
    the one-GPU point of the strong-scaling curve, with the same closed-form checks as config 4."""
    import bench
    g, adj, _ = bench.build_single(argparse.Namespace(nodes=80_000_000, entries=1_000_000_000), torch.device("cuda:0"))
    assert g.n_rows == 80_000_000 and g.nnz == 1_000_000_000 and g.nnz_entries == 1_000_000_000
    eigenvector_check(gnntf, g, adj, 128)
    del g, adj
    torch.cuda.empty_cache()
