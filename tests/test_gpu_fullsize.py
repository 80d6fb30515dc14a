"""Parity at BASELINE.json's full sizes, on the GPU box: config 3 (arxiv-shaped GCN) against the numpy oracle, and
configs 4 and 5 (RMAT 10M/100M C=256, RMAT 80M/1B C=128) through size-independent closed forms -- no CPU run is
possible at those sizes.  Tolerances: the float32 logits bar of BASELINE.json (rtol 1e-4) or tighter."""
import argparse
import os

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
    assert model.graph.last_kernel() in ("spmm_group16+chunks", "spmm_group32+chunks")     # few hub chunks: one launch with the short rows


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
    assert g.last_kernel() == "spmm_group8"


def test_config5_full_size_eigenvector(gnntf):
    """BASELINE config 5's graph on ONE GPU (RMAT 80M vertices / 1B stored entries, C=128, K=10; ~150 GB of the 288 GB):
    the one-GPU point of the strong-scaling curve, with the same closed-form checks as config 4."""
    import bench
    g, adj, _ = bench.build_single(argparse.Namespace(nodes=80_000_000, entries=1_000_000_000), torch.device("cuda:0"))
    assert g.n_rows == 80_000_000 and g.nnz == 1_000_000_000 and g.nnz_entries == 1_000_000_000
    eigenvector_check(gnntf, g, adj, 128)
    del g, adj
    torch.cuda.empty_cache()


def test_more_rows_than_one_launch_holds(gnntf):
    """68M rows at C = 256 is one WAVE per row = 4.35e9 work-items, more than the 2^32 a single dispatch may hold: the row kernels
    are dealt in pieces (SpmmArgs::slot0).  One fused step on the fixed point sqrt(deg) x s must return it, on every row --
    before the pieces existed the launch was rejected and the output left untouched."""
    import bench
    from gnntf.sharded import max_relative_deviation
    from gnntf.sparse import _launch
    n, C = 68_000_000, 256
    g, adj, _ = bench.build_single(argparse.Namespace(nodes=n, entries=136_000_000), torch.device("cuda:0"))
    deg = torch.empty(n, dtype=torch.float32, device="cuda")
    from gnntf import _native as nat
    nat.check(nat.lib().gnx_graph_colsum(g.handle, 0.0, 0, 0, nat.ptr(deg), nat.current_stream()))
    s = 0.5 + torch.arange(C, device="cuda", dtype=torch.float32) / C
    H0 = deg.sqrt().unsqueeze(1) * s.unsqueeze(0)
    out = torch.full_like(H0, float("nan"))
    _launch(adj, H0, H0, 0.9, 0.1, 0, out=out)
    assert g.last_kernel() == "spmm_wave"
    assert max_relative_deviation(out, H0) < 1e-5                    # (NaN anywhere would fail the comparison)
    assert bool(torch.isfinite(out[-1000:]).all()) and bool(torch.isfinite(out[:1000]).all())
    del out, H0, g, adj
    torch.cuda.empty_cache()


@pytest.mark.parametrize("world,cover,n,entries,C", [(8, "cover", 8_000_000, 100_000_000, 128), (4, "pull", 2_000_000, 24_000_000, 64),
                                                     (3, "cover", 1_000_003, 12_000_000, 40)])
def test_vertex_blocks_of_one_graph_match_one_gpu(gnntf, world, cover, n, entries, C):
    """SURVEY 8(e) parity check at a realistic size: ALL P vertex blocks of one R-MAT graph live on this one GPU (the ranks are
    threads of this process, tests/thread_comm.py), each with its real halo plan, send CSR, interior / boundary handles and
    column chunks; the K = 10 propagation over the blocks must equal gnx_appnp_propagate over the whole graph."""
    from gnntf import sharded
    from thread_comm import run_ranks
    device = torch.device("cuda:0")
    u, w = sharded.rmat_relabelled_pairs(n, entries // 2, seed=1, device=device)
    H0 = torch.rand(n, C, device=device, generator=torch.Generator(device=device).manual_seed(2)) * 2 - 1
    bounds = sharded.uniform_bounds(n, world)

    def rank_body(comm):
        lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
        mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
        idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
        sg = sharded.ShardedGraph(idx, torch.ones(idx.shape[0], device=device), bounds, comm=comm, cover=cover, chunks=2,
                                  split_rows="always" if world == 4 else True)      # (auto: these blocks' interior rows are isolated vertices -> one handle)
        state = sg.make_state(H0[lo:hi])
        out = sg.propagate(state, 0.1, 10).clone()
        again = sg.propagate(state, 0.1, 10)
        assert torch.equal(out, again)
        return out, sg.stats, sg.nnz_local

    parts = run_ranks(world, rank_body)
    got = torch.cat([p[0] for p in parts])
    assert sum(p[2] for p in parts) == entries
    stats = [p[1] for p in parts]
    if cover == "cover":
        assert sum(s["pull_rows"] + s["push_rows"] for s in stats) < 0.75 * sum(s["pull_only_rows"] for s in stats)   # the cover pays
    idx = torch.cat([torch.stack([u, w], 1), torch.stack([w, u], 1)])
    whole = gnntf.normalize(gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=device), (n, n)), device=device), "symmetric")
    del idx
    want = gnntf.appnp_propagate(whole, H0, 0.1, 10)
    # Both sides are float32 sums of the same terms in a different grouping: a pull-only plan differs from the one-GPU run only in
    # how the long-row kernels deal a hub row's entries to lanes (the blocks run at the chunk width, 64 columns = 16 lanes per
    # row, the whole graph at 128 = 32 lanes) and in where the long-row threshold falls; a cover also adds its pushed partial sums
    # as separate terms.  The yardstick is therefore the float32 rounding of a row's OWN sum: eps * sum_j |A_ij| |H_j| bounds one
    # regrouping of a row, K = 10 iterations compound it.  Measured (gpurun_out/vertex_block_errors.json): see the bound below.
    absA = gnntf.sparse.Adjacency(whole.graph, whole.vals.abs())
    mag = gnntf.appnp_propagate(absA, H0.abs(), 0.1, 10)                 # sum of the absolute terms behind every element
    rel = ((got - want).abs() / mag.clamp_min(1e-30)).max().item()
    scale = want.abs().max(dim=1, keepdim=True).values.clamp_min(1e-3)
    err = ((got - want).abs() / scale).max().item()
    import json
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "vertex_block_errors.json"), "a") as f:
        f.write(json.dumps(dict(world=world, cover=cover, n=n, C=C, max_err_over_row_max=err, max_err_over_sum_of_abs_terms=rel)) + "\n")
    assert rel < 8 * 1.2e-7, (rel, err)     # a few float32 roundings of the element's own sum (measured: 1.4e-7 ... 1.8e-7, pull and cover alike)
    assert err < 2e-6, err                  # (measured 2.8e-7 ... 4.2e-7 of the row maximum; round 2 accepted 2e-4)
    assert (got.argmax(1) == want.argmax(1)).float().mean().item() > 0.9999


def test_pull_plan_keeps_the_one_gpu_summation_order_bitwise(gnntf):
    """"P-GPU logits must equal 1-GPU logits bit-for-bit if per-row reduction order is kept".  A pull-only plan keeps
    it: every row's entries stay in ascending global column order in the [regions | local | regions] buffer.  What else could
    differ is matched here -- the one-GPU reference runs each 64-column chunk on its own (the blocks' kernels run at the chunk
    width) and both structures sit above 2^20 rows (same long-row threshold) -- so the K = 10 results must be IDENTICAL."""
    from gnntf import sharded
    from thread_comm import run_ranks
    device = torch.device("cuda:0")
    world, n, entries, C = 2, 4_400_000, 52_000_000, 128
    u, w = sharded.rmat_relabelled_pairs(n, entries // 2, seed=1, device=device)
    H0 = torch.rand(n, C, device=device, generator=torch.Generator(device=device).manual_seed(2)) * 2 - 1
    bounds = sharded.uniform_bounds(n, world)

    def rank_body(comm, row_window=0):
        lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
        mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
        idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
        sg = sharded.ShardedGraph(idx, torch.ones(idx.shape[0], device=device), bounds, comm=comm, cover="pull", chunks=2, row_window=row_window)
        assert getattr(sg.graph, "row_window", 0) == row_window
        state = sg.make_state(H0[lo:hi])
        assert [c1 - c0 for c0, c1 in state.cols] == [64, 64]
        return sg.propagate(state, 0.1, 10).clone()

    got = torch.cat(run_ranks(world, rank_body))
    # the same blocks on row windows (ShardedGraph(row_window=): the launch order of a locality numbering, one window per XCD chunk):
    # another order of the launches, the same sums
    windowed = torch.cat(run_ranks(world, lambda comm: rank_body(comm, 4096)))
    assert torch.equal(windowed, got)
    idx = torch.cat([torch.stack([u, w], 1), torch.stack([w, u], 1)])
    whole = gnntf.normalize(gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=device), (n, n)), device=device), "symmetric")
    del idx
    want = torch.cat([gnntf.appnp_propagate(whole, H0[:, c0:c0 + 64].contiguous(), 0.1, 10) for c0 in (0, 64)], dim=1)
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("world,n,entries,C", [(4, 2_000_000, 24_000_000, 32), (8, 4_000_000, 50_000_000, 64)])
def test_edge_dropout_on_vertex_blocks_matches_one_gpu(gnntf, world, n, entries, C):
    """Training-mode propagation (per-iteration edge dropout + re-normalisation) over ALL P blocks of one R-MAT graph on this GPU
    against the one-GPU ppr_loop with the same seed: same masks whatever the partition, forward and dH0."""
    from gnntf import sharded, sparse
    from thread_comm import run_ranks
    device = torch.device("cuda:0")
    K, a, p, seed, first = 4, 0.1, 0.5, 77, 11
    u, w = sharded.rmat_relabelled_pairs(n, entries // 2, seed=1, device=device)
    gen = torch.Generator(device=device).manual_seed(2)
    H0 = torch.rand(n, C, device=device, generator=gen) * 2 - 1
    G = torch.rand(n, C, device=device, generator=gen) * 2 - 1
    bounds = sharded.uniform_bounds(n, world)

    def rank_body(comm):
        lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
        mu, mw = (u >= lo) & (u < hi), (w >= lo) & (w < hi)
        idx = torch.cat([torch.stack([u[mu], w[mu]], 1), torch.stack([w[mw], u[mw]], 1)])
        sg = sharded.ShardedGraph(idx, torch.ones(idx.shape[0], device=device), bounds, comm=comm, edge_dropout=True)
        scales = sg.dropped_scales(p, seed, first, K)
        out = sg.propagate_dropped(H0[lo:hi], a, K, p, seed, first, scales)
        grad = sg.propagate_dropped_backward(G[lo:hi], a, K, p, seed, first, scales)
        return out, grad

    parts = run_ranks(world, rank_body)
    got, got_grad = torch.cat([q[0] for q in parts]), torch.cat([q[1] for q in parts])
    idx = torch.cat([torch.stack([u, w], 1), torch.stack([w, u], 1)])
    whole = gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=device), (n, n)), device=device)
    del idx
    D = sparse.dropped_degree_scales(whole, p, seed, first, K)
    Hf = H0.clone().requires_grad_(True)
    want = sparse.ppr_loop(lambda k, bwd=False: sparse.dropped_adjacency(whole, p, seed, first + k, D=D[k]), Hf, a, K)
    want.backward(G)
    for x, y in ((got, want.detach()), (got_grad, Hf.grad)):
        scale = y.abs().max(dim=1, keepdim=True).values.clamp_min(1e-3)
        assert ((x - y).abs() / scale).max().item() < 1e-4
