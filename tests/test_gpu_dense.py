"""The dense ends of the path on the matrix cores (csrc/gnx_dense.hip, k_spmm_gcnii) against float64 numpy / the oracle:
Dense (layers.py:135-136), the GCNII layer (gcn.py:22-27), the NodeClassification head (graph_predictor.py:16-31) and the
sparse-input form of the first Dense.  float32 tolerance of BASELINE.json: rtol 1e-4 (+ atol for sums that cancel)."""
import numpy as np
import scipy.sparse as sp
import pytest
import torch

import graphs
from oracle import gnntf_oracle as orc

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope="module")
def gnntf():
    import gnntf
    gnntf.set_default_device("cuda:0")
    yield gnntf
    gnntf.set_default_device(None)


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("n,F,O", [(1, 1, 1), (100, 7, 3), (1000, 128, 64), (513, 1433, 64), (300, 64, 40), (2049, 64, 7), (257, 100, 300),
                                   (64, 16, 16), (130, 257, 129), (4097, 512, 256),
                                   # tall enough for the W-resident persistent kernel (n >= 4096, F <= 256 and W within LDS)
                                   (5000, 64, 40), (4100, 128, 64), (9001, 256, 64), (4097, 200, 33), (6000, 128, 128), (5000, 60, 256),
                                   (300001, 100, 7)])
@pytest.mark.parametrize("relu", [False, True])
def test_dense_mfma_matches_float64(gnntf, n, F, O, relu):
    rng = np.random.default_rng(n + F + O)
    X, W, b = rng.standard_normal((n, F)).astype(np.float32), rng.standard_normal((F, O)).astype(np.float32), rng.standard_normal((1, O)).astype(np.float32)
    got = gnntf.dense(dev(X), dev(W), dev(b), relu=relu).cpu().numpy()
    want = X.astype(np.float64) @ W.astype(np.float64) + b
    want = np.maximum(want, 0) if relu else want
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-4 * np.sqrt(F))


@pytest.mark.parametrize("n,F,O", [
    # W in registers (k_dense_wreg): (F / 4) x (O / 16) <= 256 fragments per lane, O a multiple of 16
    (20001, 256, 64), (16400, 256, 32), (33007, 128, 64), (17000, 128, 128), (1_000_003, 256, 64),
    # ... with padded widths (multiples of 4): columns of X past F staged from zeros, columns past O neither loaded nor stored
    (20003, 100, 64), (16385, 200, 40), (30000, 68, 128), (17001, 252, 4), (16500, 132, 100),
    # X through the LDS-DMA ring with W in LDS (k_dense_ring): the other tall shapes with whole 64-float K chunks
    (20001, 256, 128), (16385, 64, 64), (16400, 192, 256)])
@pytest.mark.parametrize("relu,with_bias", [(False, True), (True, True), (True, False)])
def test_dense_tall_kernels(gnntf, n, F, O, relu, with_bias):
    """The persistent kernels of tall inputs (n >= 16K rows): against float64 on a sample of rows, BITWISE against the same product
    computed in slabs of fewer than 16K rows (those take k_dense_mfma: all three kernels add the k terms in the same order), and with a
    ragged last tile.  X = rows of the identity with an asymmetric W catches a transposed fragment map exactly."""
    g = torch.Generator(device="cuda").manual_seed(n + F + O)
    X = torch.randn(n, F, device="cuda", generator=g)
    W = torch.randn(F, O, device="cuda", generator=g)
    b = torch.randn(1, O, device="cuda", generator=g) if with_bias else None
    got = gnntf.dense(X, W, b, relu=relu)
    slabs = torch.cat([gnntf.dense(X[i:i + 8192], W, b, relu=relu) for i in range(0, n, 8192)])
    assert torch.equal(got, slabs)
    rows = torch.cat([torch.arange(0, 64), torch.randint(0, n, (2048,)), torch.arange(n - 64, n)]).cuda()
    want = X[rows].double() @ W.double() + (b.double() if with_bias else 0.0)
    want = torch.relu(want) if relu else want
    np.testing.assert_allclose(got[rows].cpu().numpy(), want.cpu().numpy(), rtol=RTOL, atol=1e-4 * np.sqrt(F))
    if not relu and not with_bias:
        return
    eye = torch.zeros(n, F, device="cuda")
    eye[torch.arange(n), torch.arange(n) % F] = 1.0
    Wa = (torch.arange(F * O, device="cuda", dtype=torch.float32).reshape(F, O) * 0.5 - 7)
    out = gnntf.dense(eye, Wa, None)
    assert torch.equal(out, Wa[torch.arange(n, device="cuda") % F])


@pytest.mark.parametrize("F,O", [(256, 64), (128, 128), (256, 128), (64, 64)])
def test_dense_tall_kernels_with_pitched_operands(gnntf, F, O):
    """The C entry with leading dimensions larger than the rows (X and out as aligned column slices of wider matrices): the persistent
    kernels compute their addresses from 32-bit rows x byte pitches.  Same bits as the contiguous call; the columns beside the
    result are not touched.  Likewise gnx_dense_wgrad on aligned slices."""
    from gnntf import _native as nat
    from gnntf.sparse import _dense_wgrad
    n = 20011
    g = torch.Generator(device="cuda").manual_seed(F * O)
    wide = torch.randn(n, F + 96, device="cuda", generator=g)
    X = wide[:, 32:32 + F]                                   # pitch F + 96, offset 128 bytes
    W = torch.randn(F, O, device="cuda", generator=g)
    b = torch.randn(O, device="cuda", generator=g)
    want = gnntf.dense(X.contiguous(), W, b, relu=True)
    assert torch.equal(gnntf.dense(X, W, b, relu=True), want)
    out_wide = torch.full((n, O + 64), -7.0, device="cuda")
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_dense(nat.ptr(X), X.stride(0), n, F, nat.ptr(W), W.stride(0), O, nat.ptr(b), nat.ACT_RELU,
                                      out_wide.data_ptr() + 4 * 32, out_wide.stride(0), nat.current_stream()))
    assert torch.equal(out_wide[:, 32:32 + O], want)
    assert bool((out_wide[:, :32] == -7.0).all()) and bool((out_wide[:, 32 + O:] == -7.0).all())
    gwide = torch.randn(n, O + 32, device="cuda", generator=g)
    G = gwide[:, 16:16 + O]
    assert torch.equal(_dense_wgrad(X, G), _dense_wgrad(X.contiguous(), G.contiguous()))


def test_dense_mfma_layout_and_strides(gnntf):
    """A = I with an ASYMMETRIC W catches a transposed fragment map; strided / unaligned operands take the scalar-load path."""
    W = (np.arange(48 * 40, dtype=np.float32).reshape(48, 40) * 0.5 - 7)          # W[i][j] != W[j][i]
    eye = np.eye(48, dtype=np.float32)
    np.testing.assert_array_equal(gnntf.dense(dev(eye), dev(W)).cpu().numpy(), W)  # exact: one non-zero product per output
    rng = np.random.default_rng(0)
    big = dev(rng.standard_normal((300, 131)).astype(np.float32))
    Xv = big[:, 3:100]                                                              # ld 131, offset 3: not 16-byte aligned
    Wbig = dev(rng.standard_normal((97, 70)).astype(np.float32))
    Wv = Wbig[:, 5:55]                                                              # ldw 70
    got = gnntf.dense(Xv, Wv, None).cpu().numpy()
    want = Xv.cpu().numpy().astype(np.float64) @ Wv.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-3)
    with pytest.raises(Exception, match="expect"):
        gnntf.dense(dev(eye), dev(W[:40]))
    with pytest.raises(Exception, match="no CPU fallback"):
        gnntf.dense(torch.zeros(4, 4), torch.zeros(4, 4))


def test_dense_backward(gnntf):
    rng = np.random.default_rng(3)
    X, W, b = (rng.standard_normal(s).astype(np.float32) for s in ((500, 96), (96, 48), (1, 48)))
    gout = rng.standard_normal((500, 48)).astype(np.float32)
    grads = []
    for ours in (True, False):
        Xt, Wt, bt = (dev(t).requires_grad_() for t in (X, W, b))
        out = gnntf.dense(Xt, Wt, bt, relu=True) if ours else torch.relu(Xt @ Wt + bt)
        out.backward(dev(gout))
        grads.append([t.grad.cpu().numpy() for t in (Xt, Wt, bt)])
    for a, w in zip(*grads):
        np.testing.assert_allclose(a, w, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("n,F,O", [(1, 1, 1), (33, 5, 3), (1000, 96, 48), (5000, 256, 64), (4097, 300, 7), (2500, 1433, 64), (70000, 64, 40),
                                   (3000, 128, 200)])
def test_dense_weight_gradient_mfma(gnntf, n, F, O):
    """gnx_dense_wgrad: dW = X^T . G over row slabs on the matrix cores, slabs added in a fixed order -- against float64, twice
    (bitwise repeatable), with unaligned strided operands too."""
    from gnntf.sparse import _dense_wgrad
    rng = np.random.default_rng(n + F)
    X, G = rng.standard_normal((n, F)).astype(np.float32), rng.standard_normal((n, O)).astype(np.float32)
    got = _dense_wgrad(dev(X), dev(G))
    want = X.astype(np.float64).T @ G.astype(np.float64)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=2e-4 * np.sqrt(n))
    assert torch.equal(got, _dense_wgrad(dev(X), dev(G)))
    big = dev(np.pad(X, ((0, 0), (3, 2))))                                      # ld F + 5, offset 3
    np.testing.assert_allclose(_dense_wgrad(big[:, 3:3 + F], dev(G)).cpu().numpy(), want, rtol=RTOL, atol=2e-4 * np.sqrt(n))


@pytest.mark.parametrize("n,F,O", [(16385, 64, 64), (20003, 256, 64), (17001, 128, 128), (33002, 32, 32), (16500, 64, 256), (50007, 128, 64),
                                   (300_005, 256, 32), (1_000_003, 64, 64),
                                   # widths padded to the next of {32, 64, 128, 256}; wide layers cut into feature panels
                                   (70001, 64, 40), (20002, 100, 64), (16385, 256, 256), (30001, 256, 128), (17003, 520, 64), (40000, 128, 8), (16400, 64, 300), (20000, 260, 132),
                                   (25000, 36, 12)])
def test_dense_weight_gradient_accumulator_stationary(gnntf, n, F, O):
    """k_wgrad_acc (tall inputs, widths multiples of 4: every wave keeps a whole panel of the result in registers): against
    float64, bitwise repeatable, ragged slabs; and EXACTLY on integer-valued operands (X = rows of the identity: a transposed fragment or a
    wrong LDS swizzle moves a sum to another cell)."""
    from gnntf.sparse import _dense_wgrad
    g = torch.Generator(device="cuda").manual_seed(n + F * O)
    X = torch.randn(n, F, device="cuda", generator=g)
    G = torch.randn(n, O, device="cuda", generator=g)
    got = _dense_wgrad(X, G)
    want = sum(X[i:i + 65536].double().t() @ G[i:i + 65536].double() for i in range(0, n, 65536))
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=RTOL, atol=2e-4 * np.sqrt(n))
    assert torch.equal(got, _dense_wgrad(X, G))
    rows = torch.arange(n, device="cuda")
    eye = torch.zeros(n, F, device="cuda")
    eye[rows, (rows * 7) % F] = 1.0
    Gi = torch.randint(-3, 4, (n, O), device="cuda", generator=g).float()
    exact = torch.zeros(F, O, device="cuda", dtype=torch.float64).index_add_(0, (rows * 7) % F, Gi.double())
    assert torch.equal(_dense_wgrad(eye, Gi).double(), exact)


def hub_graph(n, m, hub_entries, seed):
    """An R-MAT graph plus one hub row/column with more than LONG_ROW (512) entries."""
    coo, vals, shape = graphs.rmat_symmetric_coo(n, m, seed=seed)
    others = np.random.default_rng(seed).choice(np.arange(1, n), size=hub_entries, replace=False)
    extra = np.concatenate([np.stack([np.zeros_like(others), others], 1), np.stack([others, np.zeros_like(others)], 1)])
    coo = np.unique(np.concatenate([coo, extra]), axis=0)
    return coo, np.ones(len(coo), dtype=np.float32), shape


@pytest.mark.parametrize("C", [16, 32, 64, 48, 128, 256])
@pytest.mark.parametrize("relu", [True, False])
def test_gcnii_step_fused(gnntf, C, relu):
    """gcn.py:22-27 in one launch (C = 16/32/64: LDS tile + MFMA; hub rows through the long-row + dense kernels) or as
    SpMM + dense (other widths), against the oracle's float64 arithmetic."""
    n = 3000
    coo, vals, shape = hub_graph(n, 20000, 900, seed=C)
    g = gnntf.DeviceGraph(gnntf.SparseCOO(coo, vals, shape), device="cuda:0")
    adj = gnntf.normalize(g, "symmetric")
    rng = np.random.default_rng(C)
    H, H0 = rng.standard_normal((n, C)).astype(np.float32), rng.standard_normal((n, C)).astype(np.float32)
    M = (0.6 * np.eye(C) + 0.4 * rng.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)
    with torch.no_grad():
        got = gnntf.gcnii_step(adj, dev(H), dev(H0), 0.1, dev(M), relu=relu).cpu().numpy()
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    want = orc.ppr_iteration(ai, av, shape, H.astype(np.float64), H0.astype(np.float64), 0.1) @ M.astype(np.float64)
    want = np.maximum(want, 0) if relu else want
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-4)
    assert g.last_kernel() == ("spmm_gcnii_mfma" if C in (16, 32, 64) else "spmm+dense_mfma")
    # with gradients the SAME launch also writes the mixed rows (dM needs them): bitwise the inference output, and dM / dH / dH0
    # against float64 algebra (g = a random upstream gradient, not all ones: the hub rows' products must be right too)
    Ht, H0t, Mt = dev(H).requires_grad_(), dev(H0).requires_grad_(), dev(M).requires_grad_()
    out = gnntf.gcnii_step(adj, Ht, H0t, 0.1, Mt, relu=relu)
    assert g.last_kernel() == ("spmm_gcnii_mfma" if C in (16, 32, 64) else "spmm+dense_mfma")
    assert np.array_equal(out.detach().cpu().numpy(), got)
    up = rng.standard_normal((n, C))
    out.backward(dev(up.astype(np.float32)))
    T = orc.ppr_iteration(ai, av, shape, H.astype(np.float64), H0.astype(np.float64), 0.1)
    gate = ((T @ M > 0) if relu else np.ones((n, C), dtype=bool)) * up.astype(np.float32).astype(np.float64)
    dT = gate @ M.T.astype(np.float64)
    A = sp.csr_matrix((av, (ai[:, 0], ai[:, 1])), shape=shape)
    np.testing.assert_allclose(Mt.grad.cpu().numpy(), T.T @ gate, rtol=1e-3, atol=2e-2)
    np.testing.assert_allclose(H0t.grad.cpu().numpy(), 0.1 * dT, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(Ht.grad.cpu().numpy(), 0.9 * (A.T @ dT), rtol=1e-3, atol=1e-3)
    # ... and the same gradients as the two-launch composition (fused SpMM+mix, then the dense kernel) that the step replaces
    H2, H02, M2 = dev(H).requires_grad_(), dev(H0).requires_grad_(), dev(M).requires_grad_()
    two = gnntf.dense(gnntf.ppr_step(adj, H2, H02, 0.1), M2, None, relu=relu)
    two.backward(dev(up.astype(np.float32)))
    np.testing.assert_allclose(out.detach().cpu().numpy(), two.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    for a_, b_ in ((Mt.grad, M2.grad), (Ht.grad, H2.grad), (H0t.grad, H02.grad)):
        np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_node_head(gnntf):
    """gather + log-softmax + CE + mean, its backward, and gather + argmax (first maximum), against the oracle / torch."""
    rng = np.random.default_rng(8)
    for C in (1, 7, 16, 40, 129):
        n, m = 2000, 700
        logits = (rng.standard_normal((n, C)) * 3).astype(np.float32)
        nodes = rng.integers(0, n, size=m)                                      # with repeats
        labels = rng.integers(0, C, size=m)
        L = dev(logits).requires_grad_()
        loss = gnntf.node_ce(L, nodes, labels)
        want = orc.node_loss(logits.astype(np.float64), nodes, labels)
        assert abs(float(loss) - want) <= 1e-5 * max(abs(want), 1.0), (C, float(loss), want)
        (loss * 1.7).backward()
        Lt = dev(logits).requires_grad_()
        (torch.nn.functional.cross_entropy(Lt[dev(nodes)], dev(labels)) * 1.7).backward()
        np.testing.assert_allclose(L.grad.cpu().numpy(), Lt.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_array_equal(gnntf.node_argmax(dev(logits), nodes).cpu().numpy(), orc.node_predict(logits, nodes))
        np.testing.assert_array_equal(gnntf.node_argmax(dev(logits)).cpu().numpy(), logits.argmax(1))
    ties = np.zeros((5, 9), dtype=np.float32); ties[1, 3] = ties[1, 7] = 2.0; ties[2, 8] = 1.0
    assert gnntf.node_argmax(dev(ties)).cpu().numpy().tolist() == [0, 3, 8, 0, 0]
    with pytest.raises(Exception, match="out of range"):
        gnntf.node_ce(dev(logits), [0, n], [0, 0])
    with pytest.raises(Exception, match="out of range"):
        gnntf.node_ce(dev(logits), [0, 1], [0, 129])
    with pytest.raises(Exception, match="out of range"):
        gnntf.node_argmax(dev(logits), [-1])
    # ids handed over as DEVICE tensors skip the host check: the kernels never dereference them -- NaN loss, -1 argmax, no gradient
    bad_nodes = torch.tensor([0, n, 5], device="cuda"); some_labels = torch.tensor([0, 0, 0], device="cuda")
    Lb = dev(logits).requires_grad_()
    bad = gnntf.node_ce(Lb, bad_nodes, some_labels)
    assert bool(torch.isnan(bad))
    assert gnntf.node_argmax(dev(logits), bad_nodes).cpu().numpy().tolist() == [int(logits[0].argmax()), -1, int(logits[5].argmax())]
    assert bool(torch.isnan(gnntf.edge_scores(dev(logits), torch.tensor([[0, 1], [2, n]], device="cuda"))[1]))
    # through the task API (graph_predictor.py:10-31)
    task = gnntf.NodeClassification(list(range(50)), labels[:50] % 7)
    small = dev(logits[:, :7].copy())
    assert abs(float(task.loss(small)) - orc.node_loss(logits[:, :7].astype(np.float64), np.arange(50), labels[:50] % 7)) < 1e-5
    assert task.predict(small).cpu().numpy().tolist() == logits[:50, :7].argmax(1).tolist()
    assert abs(task.evaluate(small) - orc.node_evaluate(logits[:, :7], np.arange(50), labels[:50] % 7)) < 1e-12


def test_sparse_input_features(gnntf):
    """Mostly-zero input features (Cora-shaped: 1433 columns, 1.3 % non-zero) reach the first Dense as a device CSR:
    X . W through the SpMM kernel == the dense product; dropout on them drops stored entries with the counter RNG."""
    coo, vals, shape, X = graphs.cora_shaped(seed=0)
    rng = np.random.default_rng(1)
    W, b = rng.standard_normal((X.shape[1], 64)).astype(np.float32), rng.standard_normal((1, 64)).astype(np.float32)
    rows = gnntf.SparseRows.from_dense(dev(X))
    assert rows.graph.nnz == np.count_nonzero(X) and rows.shape == X.shape
    want = np.maximum(X.astype(np.float64) @ W + b, 0)
    Wt, bt = dev(W).requires_grad_(), dev(b).requires_grad_()
    out = gnntf.sparse_dense(rows, Wt, bt, relu=True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    gout = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(gout))
    gate = gout * (want > 0)
    np.testing.assert_allclose(Wt.grad.cpu().numpy(), X.T.astype(np.float64) @ gate, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), gate.sum(0, keepdims=True), rtol=1e-3, atol=1e-3)
    # a pending dropout: exactly the oracle's keep mask over X's stored entries (row-major order of np.nonzero)
    xi = np.stack(np.nonzero(X), 1)
    keep = orc.keep_mask(xi, 0.5, 77, 3)
    Xd = np.zeros_like(X); Xd[xi[keep, 0], xi[keep, 1]] = X[xi[keep, 0], xi[keep, 1]] * 2
    dropped = gnntf.sparse_dense(rows.with_dropout(0.5, 77, 3), dev(W), None).cpu().numpy()
    np.testing.assert_allclose(dropped, Xd.astype(np.float64) @ W, rtol=RTOL, atol=ATOL)
    assert 0.4 < keep.mean() < 0.6
    # the model picks the sparse form by itself (APPNP: Dropout -> Dense ...), a GCN (SpMM first) does not
    model = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), X, num_classes=7)
    assert isinstance(model._input_features(), gnntf.SparseRows)
    gcn = gnntf.GCN(gnntf.SparseCOO(coo, vals, shape), X, num_classes=7)
    assert gcn._input_features() is gcn.features
    model.reset()
    with model:                                                                # training mode: input dropout on the stored entries
        a = model(model.features)
    assert a.shape == (shape[0], 7) and bool(torch.isfinite(a).all())
    dense_X = rng.standard_normal((shape[0], 30)).astype(np.float32)
    assert gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), dense_X, num_classes=7)._input_features().__class__ is torch.Tensor


def test_features_given_sparse_are_never_densified(gnntf):
    """The reference's own files hold the attribute matrix as a CSR (experiment_setup.py:273-282); datasets.load_gnn_benchmark_npz
    hands it over as a SparseCOO and the model keeps it as device SparseRows: same logits, same training losses as the dense matrix."""
    coo, vals, shape, X = graphs.cora_shaped(seed=2)
    xi = np.stack(np.nonzero(X), 1).astype(np.int64)
    sparse_X = gnntf.SparseCOO(xi, X[xi[:, 0], xi[:, 1]], X.shape)
    labels = np.random.default_rng(0).integers(0, 7, size=shape[0])
    tr = list(range(140))
    outs, losses = [], []
    for feats in (sparse_X, X):
        gnntf.set_seed(3)
        model = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), feats, num_classes=7)
        if feats is sparse_X:
            assert isinstance(model.features, gnntf.SparseRows) and model._input_features() is model.features
            assert model.features.graph.nnz == len(xi) and model.top_shape() == (shape[0], 7)
        model.train(train=gnntf.NodeClassification(tr, labels[tr]), epochs=4, patience=10)
        model.training_mode(False)
        with torch.no_grad():
            outs.append(model(model.features))
        losses.append(float(model.loss(gnntf.NodeClassification(tr, labels[tr]))))
    assert torch.equal(outs[0], outs[1]) and losses[0] == losses[1]
    with pytest.raises(Exception, match="sparse input features need"):
        gcn = gnntf.GCN(gnntf.SparseCOO(coo, vals, shape), sparse_X, num_classes=7)
        gcn(gcn.features)
    with pytest.raises(Exception, match="reorder needs dense input features"):
        gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), sparse_X, num_classes=7, reorder="degree")


def test_link_head_edge_scores(gnntf):
    """gnx_edge_scores (graph_predictor.py:122-126): logits of listed edges in one launch, with and without the DistMult
    weights, forward against float64 numpy and backward against torch autograd."""
    rng = np.random.default_rng(14)
    for C in (1, 5, 16, 40, 128):
        n, m = 3000, 5000
        F = rng.standard_normal((n, C)).astype(np.float32)
        edges = rng.integers(0, n, size=(m, 2))
        r = (rng.random((C, 1)) + 0.5).astype(np.float32)
        for weights in (None, r):
            Ft = dev(F).requires_grad_()
            rt = None if weights is None else dev(weights).requires_grad_()
            z = gnntf.edge_scores(Ft, edges, rt)
            want = orc.link_logits(F.astype(np.float64), edges, None if weights is None else weights.astype(np.float64))
            np.testing.assert_allclose(z.detach().cpu().numpy(), want, rtol=RTOL, atol=1e-4)
            g = rng.standard_normal(m).astype(np.float32)
            z.backward(dev(g))
            Fr = dev(F).requires_grad_()
            rr = None if weights is None else dev(weights).requires_grad_()
            prod = Fr[dev(edges[:, 0])] * Fr[dev(edges[:, 1])]
            (prod.sum(1) if rr is None else (prod @ rr).reshape(-1)).backward(dev(g))
            np.testing.assert_allclose(Ft.grad.cpu().numpy(), Fr.grad.cpu().numpy(), rtol=1e-3, atol=1e-4)
            if rt is not None:
                np.testing.assert_allclose(rt.grad.cpu().numpy(), rr.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    with pytest.raises(Exception, match="out of range"):
        gnntf.edge_scores(dev(F), [[0, n]])
    # through the task API, both losses and both similarities
    labels = rng.integers(0, 2, size=200).astype(np.float32); labels[:2] = [0, 1]
    e = edges[:200]
    F = (F * 0.1).astype(np.float32)                                          # keep the float32 sigmoid away from saturation (ties)
    Fd = dev(F)
    for sim in ("dot", "cos"):
        task = gnntf.LinkPrediction(e, labels, similarity=sim)
        assert abs(float(task.loss(Fd)) - orc.link_loss_diff(F.astype(np.float64), e, similarity=sim)) < 1e-4
        bce = gnntf.LinkPrediction(e, labels, similarity=sim, loss="bce")
        assert abs(float(bce.loss(Fd)) - orc.link_loss_bce(F.astype(np.float64), e, labels, similarity=sim)) < 1e-4
        z = orc.link_logits(F.astype(np.float64), e, similarity=sim)
        assert abs(task.evaluate(Fd) - orc.auc_by_pairs(labels, 1 / (1 + np.exp(-z)))) < 1e-4


def test_ngcf_model(gnntf):
    """NGCFLayer / NGCF (gcn.py:116-154) over the bipartite-normalised propagation: layer output against the oracle, the
    model's axis-0 stacked output, and a few epochs of link-prediction training through architecture.train()."""
    import networkx as nx
    import random
    rng = np.random.default_rng(21)
    n, F = 400, 12
    G = nx.Graph(); G.add_nodes_from(range(n))
    for u, v in rng.integers(0, n, size=(1600, 2)):
        if u != v:
            G.add_edge(int(u), int(v))
    X = rng.standard_normal((n, F)).astype(np.float32)
    gnntf.set_seed(2)
    model = gnntf.NGCF(gnntf.graph2adj(G), X, num_classes=8, dropout=0.0)
    model.reset()
    layers = [l for l in model.layers() if isinstance(l, gnntf.NGCFLayer)]
    assert len(layers) == 3 and model.layers()[-1].__class__ is gnntf.Concatenate
    model.training_mode(False)
    with torch.no_grad():
        out = model(model.features)
    assert out.shape == (3 * n, 8)                                            # the reference's axis-0 stacking (layers.py:98-101)
    coo = gnntf.graph2adj(G)
    idx, vals = coo.indices.cpu().numpy(), coo.values.cpu().numpy()
    l0 = layers[0]
    want = orc.ngcf_layer_eval(idx, vals, (n, n), X, *(t.detach().cpu().numpy() for t in (l0.W1, l0.b1, l0.W2, l0.b2)))
    np.testing.assert_allclose(out[:n].cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(np.linalg.norm(out[:n].cpu().numpy(), axis=1), 1.0, rtol=1e-5)
    # training: pairwise loss over a negative sampler, early stopping on held-out edges, AUC above chance
    random.seed(0)
    edges = np.array(list(G.edges()))
    perm = rng.permutation(len(edges))
    train, valid = edges[perm[:1000]], edges[perm[1000:1200]]
    model.train(train=gnntf.LinkPrediction(gnntf.negative_sampling(train, G)),
                valid=gnntf.LinkPrediction(*gnntf.negative_sampling(valid, G)()), epochs=30, patience=30)
    test_edges, test_labels = gnntf.negative_sampling(edges[perm[1200:1400]], G)()
    scores = model.predict(gnntf.LinkPrediction(test_edges))
    assert scores.shape[0] == len(test_labels) and gnntf.auc(test_labels, scores) > 0.5
    # zero-width features + structural embeddings (demos/development/library_recommendation.py:45-47)
    emb = gnntf.NGCF(gnntf.graph2adj(G), np.zeros((n, 0), dtype=np.float32), num_classes=8,
                     preprocessor=gnntf.Structural(dims=16, regularize=0, bipartite=100))
    emb.reset()
    emb.training_mode(False)
    with torch.no_grad():
        assert emb(emb.features).shape == (3 * n, 8)


def test_spectral_preserving_layer_variants(gnntf):
    """gcn.py:30-51, 92-105 through layer_type=: 2 * dropout(act(z + b) - b) over the same kernels, against numpy in eval mode."""
    coo, vals, shape = graphs.rmat_symmetric_coo(900, 7000, seed=6)
    rng = np.random.default_rng(6)
    X = rng.standard_normal((900, 24)).astype(np.float32)
    ai, av = orc.get_adjacency(coo, vals, shape, dtype=np.float64)
    gcn = gnntf.GCN(gnntf.SparseCOO(coo, vals, shape), X, num_classes=5, latent_dims=[16], layer_type=gnntf.GCNSpectralPreservingLayer)
    gcn.reset()
    H = X.astype(np.float64)
    for layer in gcn.layers():
        with torch.no_grad():
            layer.b.copy_(dev(rng.uniform(-0.3, 0.3, size=tuple(layer.b.shape)).astype(np.float32)))
        W, b = layer.W.detach().cpu().numpy().astype(np.float64), layer.b.detach().cpu().numpy().astype(np.float64)
        H = 2 * (np.maximum(orc.sparse_dense_matmul(ai, av, shape, H) @ W + b, 0) - b)
    gcn.training_mode(False)
    with torch.no_grad():
        np.testing.assert_allclose(gcn(gcn.features).cpu().numpy(), H, rtol=RTOL, atol=1e-4)
    model = gnntf.GCNII(gnntf.SparseCOO(coo, vals, shape), X, num_classes=5, latent_dims=[32], iterations=3,
                        layer_type=gnntf.GCNIISpectralPreservingLayer)
    model.reset()
    convs = [l for l in model.layers() if isinstance(l, gnntf.GCNIISpectralPreservingLayer)]
    dense = [l for l in model.layers() if isinstance(l, gnntf.Dense)]
    for l in convs:
        with torch.no_grad():
            l.W.copy_(dev((rng.standard_normal((32, 32)) * 0.2).astype(np.float32)))
            l.bias.copy_(dev(rng.uniform(-0.2, 0.2, size=(1, 32)).astype(np.float32)))
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    H0 = np.maximum(X.astype(np.float64) @ f64(dense[0].W) + f64(dense[0].b), 0)
    H = H0
    for k, l in enumerate(convs):
        beta = np.log1p(0.5 / (k + 1))
        T = orc.ppr_iteration(ai, av, shape, H, H0, 0.1)
        H = 2 * (np.maximum(T @ ((1 - beta) * np.eye(32) + beta * f64(l.W)) + f64(l.bias), 0) - f64(l.bias))
    want = H @ f64(dense[1].W) + f64(dense[1].b)
    model.training_mode(False)
    with torch.no_grad():
        np.testing.assert_allclose(model(model.features).cpu().numpy(), want, rtol=RTOL, atol=1e-4)
