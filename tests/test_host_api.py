"""Host-side mirror of the reference's Layer / Layered / Trainable / Predictor protocol
(reference gnntf/core/nn/layered.py, variables.py, trainable.py, layers.py,
graph_predictor.py, graph_manipulation.py) -- everything that does not need the GPU."""
import networkx as nx
import numpy as np
import pytest
import torch

import gnntf
from oracle import gnntf_oracle as orc


@pytest.fixture(autouse=True)
def cpu_default():
    gnntf.set_default_device("cpu")
    yield
    gnntf.set_default_device(None)


MLP = gnntf.MLP   # reference gnntf/core/nn/architectures/mlp.py:6-12


def test_layer_protocol_and_shapes():
    arch = gnntf.Layered((10, 6))
    assert arch.top_shape() == (10, 6) and arch.is_training()
    d1 = arch.add(gnntf.Dense(4, activation=gnntf.relu))
    d2 = arch.add(gnntf.Dense())          # outputs default to the incoming width (layers.py:126-127)
    assert d1.output_shape == (10, 4) and d2.output_shape == (10, 4) and arch.top_layer() is d2
    assert len(d1.vars) == 2 and len(arch.vars()) == 4
    assert [v.normalization for v in arch.vars()] == ["small", "zero", "small", "zero"]
    arch.reset()
    W = arch.vars()[0].var
    assert W.shape == (6, 4) and float(W.detach().abs().max()) <= 0.5 + 1e-6      # U(+-1/sqrt(fan_out))
    assert float(arch.vars()[1].var.abs().max()) == 0
    out = arch(torch.ones(10, 6))
    assert out.shape == (10, 4) and d1.value.shape == (10, 4) and (d1.value >= 0).all()
    assert d2.value is out


def test_layer_errors_match_reference_messages():
    class NoBuild(gnntf.Layer):
        pass

    class NoShape(gnntf.Layer):
        def __build__(self, arch):
            return None

    class NoForward(gnntf.Layer):
        def __build__(self, arch):
            return arch.top_shape()

    arch = gnntf.Layered((3, 3))
    with pytest.raises(Exception, match="Layer should implment a __build__ method"):
        arch.add(NoBuild())
    with pytest.raises(Exception, match="Layer __build__ should return an output shape"):
        arch.add(NoShape())
    arch.add(NoForward())
    with pytest.raises(Exception, match="Layer should implement a __forward__ method"):
        arch(torch.zeros(3, 3))
    with pytest.raises(Exception, match="Invalid normalization type"):
        gnntf.WrappedVariable((2, 2), normalization="nope").reset()
    for method in ("predict", "loss", "evaluate"):
        with pytest.raises(Exception, match="Predictors need to implement"):
            getattr(gnntf.Predictor(), method)(None)


def test_training_mode_quirk():
    """Training mode starts True and only turns False when a `with` block exits (layered.py:9,37-42)."""
    arch = gnntf.Layered((4, 4))
    assert arch.is_training()
    x = torch.ones(200, 50)
    assert (arch.dropout(x, 0.5) == 0).any()
    with arch as params:
        assert params == [] and arch.is_training()
    assert not arch.is_training()
    assert arch.dropout(x, 0.5) is x
    arch.training_mode(True)
    assert arch.dropout(x, 0) is x and arch.is_training()


def test_variable_sharing_and_init_schemes():
    gen = gnntf.VariableGenerator()
    a = gen.create_var((3, 3), "eye", shared_name="s")
    b = gen.create_var((3, 3), "ones", shared_name="s")
    assert a is b and len(gen.vars()) == 1
    gen.create_var((4, 9), 0.25)
    gen.create_var((4, 9), "bernouli", trainable=False, regularize=False)
    gen.reset()
    assert torch.equal(gen.vars()[0].var, torch.eye(3))
    assert float(gen.vars()[1].var.abs().max()) <= 0.25
    assert set(np.round(gen.vars()[2].numpy() * 3, 5).ravel().tolist()) <= {-1.0, 1.0}
    assert gen.vars()[2].regularize == 0.0 and not gen.vars()[2].var.requires_grad
    snap = gen.vars()[1].identity()
    gen.vars()[1].assign(torch.zeros(4, 9))
    assert float(gen.vars()[1].var.abs().sum()) == 0 and float(snap.abs().sum()) > 0
    with pytest.raises(TypeError):
        gen.create_var()          # the reference's APPNP(a=None) path raises the same way (filter.py:35)


def test_node_classification_matches_oracle():
    logits = torch.tensor(np.random.default_rng(0).standard_normal((30, 5)), dtype=torch.float32)
    nodes, labels = [3, 7, 11, 29], np.array([0, 4, 2, 2])
    task = gnntf.NodeClassification(nodes, labels)
    assert task.predict(logits).tolist() == orc.node_predict(logits.numpy(), nodes).tolist()
    assert float(task.loss(logits)) == pytest.approx(orc.node_loss(logits.numpy().astype(np.float64), nodes, labels), rel=1e-5)
    assert task.evaluate(logits) == pytest.approx(orc.node_evaluate(logits.numpy(), nodes, labels))
    with pytest.raises(Exception, match="Evaluation requires node labels"):
        gnntf.NodeClassification(nodes).loss(logits)
    assert gnntf.acc(task.predict(logits), labels) == pytest.approx(task.evaluate(logits))


def test_graph2adj_matches_oracle():
    G = nx.DiGraph()
    G.add_nodes_from(["c", "a", "b", "z"])
    G.add_edge("a", "b", weight=2.5)
    G.add_edge("c", "a")
    G.add_edge("b", "c", weight=0.5)
    for directed in (False, True):
        adj = gnntf.graph2adj(G, directed=directed)
        weights = [d.get("weight", 1.0) for _, _, d in G.edges(data=True)]
        idx, vals, shape = orc.graph2adj(list(G), list(G.edges()), weights, directed=directed)
        assert adj.indices.numpy().tolist() == idx.tolist()
        assert adj.values.numpy().tolist() == vals.tolist() and adj.dense_shape == shape
    assert gnntf.graph2indices(G) == [[0, 1], [1, 2], [2, 0]]
    back = gnntf.adj2graph(range(4), gnntf.graph2adj(G, directed=True))
    assert sorted(back.edges()) == [(0, 1), (1, 2), (2, 0)]


def test_trainable_loop_on_cpu_tensors():
    """train(): reset, Adam, L2 on regularised vars, early stopping, best-weights restore, eval
    mode afterwards, cached predict (trainable.py:41-103) -- on an MLP (no propagation layer)."""
    gnntf.set_seed(0)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((300, 8)).astype(np.float32)
    labels = (X[:, 0] + X[:, 1] > 0).astype(np.int64)
    model = MLP(X, 2, latent_dims=[16], dropout=0.1)
    train, valid, test = list(range(0, 150)), list(range(150, 225)), list(range(225, 300))
    model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]),
                patience=20, epochs=200, learning_rate=0.05)
    assert not model.is_training()
    pred = model.predict(gnntf.NodeClassification(test))
    assert gnntf.acc(pred, labels[test]) > 0.85
    first = model._fast_predict
    model.evaluate(gnntf.NodeClassification(test, labels[test]))
    assert model._fast_predict is first       # memoised (trainable.py:26-29)
    model.reset()
    assert model._fast_predict is None


def test_flow_layers():
    arch = gnntf.Layered((5, 3))
    a = arch.add(gnntf.Dense(3))
    b = arch.add(gnntf.Activation("tanh"))
    c = arch.add(gnntf.Resume(a))
    t = arch.add(gnntf.Tradeoff([a, b]))
    arch.reset()
    out = arch(torch.ones(5, 3))
    np.testing.assert_allclose(out.detach().numpy(), (0.5 * a.value + 0.5 * b.value).detach().numpy(), rtol=1e-6)
    assert c.value is a.value and t.output_shape == (5, 3)
    with pytest.raises(Exception, match="Mismatching trade-off dimentions"):
        gnntf.Layered((5, 3), [gnntf.Dense(2)]).add(gnntf.Tradeoff([a, gnntf.Layered((5, 3)).add(gnntf.Dense(2))]))
    # un-equal gates: sigmoid(w_i) / sum sigmoid(w) (layers.py:117-122); fixed weights are taken as they are
    fixed = arch.add(gnntf.Tradeoff([a, b], weights=[0.0, 2.0]))
    s0, s1 = 0.5, 1 / (1 + np.exp(-2.0))
    np.testing.assert_allclose(fixed(arch, None).detach().numpy(), ((s0 * a.value + s1 * b.value) / (s0 + s1)).detach().numpy(), rtol=1e-6)
    # Branch restarts from a given matrix; Concatenate stacks along axis 0 while declaring a wider axis 1 (layers.py:93-101)
    side = torch.arange(15.0).reshape(5, 3)
    br = arch.add(gnntf.Branch(side))
    assert br(arch, torch.zeros(1)) is side and br.output_shape == (5, 3)
    cat = arch.add(gnntf.Concatenate(a))
    assert cat.output_shape == (5, 6) and tuple(cat(arch, side).shape) == (10, 3) and torch.equal(cat.value[:5], side)
    both = arch.add(gnntf.Concatenate([a, b]))
    assert torch.equal(both(arch, None), torch.cat([a.value, b.value], dim=0))
    with pytest.raises(Exception, match="Mismatching first dimension to concatenate"):
        arch.add(gnntf.Concatenate(gnntf.Layered((4, 3)).add(gnntf.Dense(3))))
    # the parametrised activations (layers.py:139-172): parameter counts, initial behaviour, formulas
    x = torch.linspace(-2, 2, 15).reshape(5, 3)
    for name, n_vars, want in (("scale", 1, x), ("kernel", 6, torch.log(torch.exp(x) + 2)), ("softthresh", 1, x),
                               ("softmax", 0, torch.softmax(x, dim=1)), ("exp", 0, torch.exp(x)), ("linear", 0, x)):
        probe = gnntf.Layered((5, 3))
        act = probe.add(gnntf.Activation(name))
        probe.reset()
        assert len(act.vars) == n_vars, name
        np.testing.assert_allclose(act(probe, x).detach().numpy(), want.numpy(), rtol=1e-6, atol=1e-7, err_msg=name)
    thr = gnntf.Layered((5, 3))
    np.testing.assert_allclose(thr.add(gnntf.Activation("softthresh", threshold=0.5))(thr, x).numpy(),
                               (torch.relu(x - 0.5) - torch.relu(0.5 - x)).numpy())
    assert gnntf.Layered((5, 3)).add(gnntf.Activation(torch.sign)).activation is torch.sign       # any callable in place of a name


def test_npz_dataset_roundtrip(tmp_path):
    """Local-file replacement of the reference's dgl_setup tuple (experiment_setup.py:179)."""
    G = nx.path_graph(5)
    labels = np.array([0, 1, 0, 1, 0])
    X = np.eye(5, dtype=np.float32)
    path = str(tmp_path / "toy.npz")
    gnntf.save_npz(path, G, labels, X, [0, 1], [2], [3, 4])
    adj, l2, f2, train, valid, test = gnntf.load_npz(path)
    want = gnntf.graph2adj(G)
    assert adj.indices.tolist() == want.indices.tolist() and adj.dense_shape == (5, 5)
    assert l2.tolist() == labels.tolist() and (f2 == X).all() and (train, valid, test) == ([0, 1], [2], [3, 4])


def test_container_lets_a_layer_take_the_following_ones_along():
    """Layer.__run__ (not in the reference): the container's loop (layered.py:52-55) offers every layer the chance to execute
    a run of layers in one go; the default declines, ``fuse_runs = False`` never asks, and either way the output and every
    layer's ``.value`` are what the plain loop gives."""
    class Twice(gnntf.Layer):
        def __build__(self, arch):
            return arch.top_shape()

        def __forward__(self, arch, x):
            return 2 * x

        def __run__(self, arch, x, stack, at):
            run = [l for l in stack[at:at + 3] if isinstance(l, Twice)]
            if len(run) < 2:
                return None
            calls.append(len(run))
            for k, layer in enumerate(run):
                layer.value = x * 2 ** (k + 1)
            return len(run), run[-1].value

    calls = []
    arch = gnntf.Layered((2, 2), [Twice(), Twice(), Twice(), Twice()])
    x = torch.ones(2, 2)
    assert torch.equal(arch(x), 16 * x) and calls == [3]              # three in one go, the fourth alone
    assert [float(l.value[0, 0]) for l in arch.layers()] == [2, 4, 8, 16]
    arch.fuse_runs = False
    assert torch.equal(arch(x), 16 * x) and calls == [3]


def test_locality_order_groups_communities():
    """gnntf.ordering.locality_order (GNN(reorder="locality")): a permutation; on a planted-partition graph with shuffled labels
    most entries end up between vertices that are close in the new numbering; vertices without entries trail."""
    from gnntf import ordering
    rng = np.random.default_rng(0)
    n, k, size = 6000, 30, 190
    comm = np.full(n, -1)
    members = rng.permutation(n)[: k * size].reshape(k, size)                       # 300 vertices stay without entries
    for c in range(k):
        comm[members[c]] = c
    src = members[rng.integers(k, size=60000), rng.integers(size, size=60000)]
    inside = rng.random(60000) < 0.85
    dst = np.where(inside, members[comm[src], rng.integers(size, size=60000)], members[rng.integers(k, size=60000), rng.integers(size, size=60000)])
    idx = torch.from_numpy(np.concatenate([np.stack([src, dst], 1), np.stack([dst, src], 1)]))
    order = ordering.locality_order(idx, n)
    assert sorted(order.tolist()) == list(range(n))
    newid = torch.empty_like(order)
    newid[order] = torch.arange(n)
    near = lambda a, b: float(((a - b).abs() < 2 * size).float().mean())
    assert near(newid[idx[:, 0]], newid[idx[:, 1]]) > 0.5 > 3 * near(idx[:, 0], idx[:, 1])
    degree = torch.bincount(idx[:, 0], minlength=n)
    assert bool((degree[order[-(n - k * size):]] == 0).all()) and bool((degree[order[: k * size]] > 0).all())
    labels = ordering.propagate_labels(idx[:, 0], idx[:, 1], n)
    assert bool((labels[degree == 0] == torch.arange(n)[degree == 0]).all())          # no neighbours: the own label stays
    # the acceptance rule: a share well above what a random numbering gives, on a graph of more than a few windows
    assert ordering.found_communities(0.36, 10_000_000, 4096) and not ordering.found_communities(0.002, 10_000_000, 4096)
    assert not ordering.found_communities(1.0, 2708, 4096) and not ordering.found_communities(0.37, 20_000, 4096)
    assert not ordering.found_communities(0.35, 200_000, 4096, baseline_share=0.3)            # hubs close under the degree order too: not communities
    assert ordering.degree_order_share(idx, n, 2 * size) < 0.5 * ordering.share_within(idx, newid, 2 * size)
    assert ordering.share_within(idx, newid, 2 * size) > 0.5 and ordering.share_within(idx[:0], newid, 10) == 0.0


# ---- the reference's own on-disk format (experiment_setup.py:273-282) and its splits (:183-201) -----------------------------------
def _gnn_benchmark_file(tmp_path, n=60, classes=3, seed=0):
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    A = sps.random(n, n, density=0.08, random_state=1, format="csr", dtype=np.float64)
    A.data[:] = rng.integers(1, 3, size=A.nnz)
    X = sps.random(n, 40, density=0.05, random_state=2, format="csr", dtype=np.float64)
    labels = rng.integers(0, classes, size=n)
    labels[:5] = -1                                                            # unlabelled nodes
    path = str(tmp_path / "toy.npz")
    np.savez(path, **{"adj_matrix.data": A.data, "adj_matrix.indices": A.indices, "adj_matrix.indptr": A.indptr, "adj_matrix.shape": A.shape,
                      "attr_matrix.data": X.data, "attr_matrix.indices": X.indices, "attr_matrix.indptr": X.indptr, "attr_matrix.shape": X.shape,
                      "labels": labels})
    return path, A, X, labels


def test_gnn_benchmark_npz_is_read_as_the_reference_reads_it(tmp_path):
    """load_npz on the gnn-benchmark layout: the adjacency is what graph2adj makes of the DiGraph the reference builds from
    adj_matrix (one arc per stored entry with its weight, then the reversed arcs APPENDED), the attribute matrix stays sparse."""
    import gnntf
    from gnntf import datasets
    path, A, X, labels = _gnn_benchmark_file(tmp_path)
    adj, got_labels, feats, train, valid, test = gnntf.load_npz(path)
    assert isinstance(adj, gnntf.SparseCOO) and isinstance(feats, gnntf.SparseCOO)
    assert tuple(adj.dense_shape) == A.shape and tuple(feats.dense_shape) == X.shape
    coo = A.tocoo()
    idx = np.asarray(adj.indices)
    assert idx.shape == (2 * A.nnz, 2)
    np.testing.assert_array_equal(idx[:A.nnz], np.stack([coo.row, coo.col], 1))             # row by row, in stored order
    np.testing.assert_array_equal(idx[A.nnz:], idx[:A.nnz, ::-1])                          # the reversed arcs, appended
    np.testing.assert_array_equal(np.asarray(adj.values)[:A.nnz], A.data.astype(np.float32))
    dense = np.zeros(X.shape, dtype=np.float32)
    f_idx = np.asarray(feats.indices)
    dense[f_idx[:, 0], f_idx[:, 1]] = np.asarray(feats.values)
    np.testing.assert_array_equal(dense, X.toarray().astype(np.float32))
    np.testing.assert_array_equal(got_labels, labels)
    assert (train, valid, test) == datasets.custom_splits(labels, 20, 500, 0)
    directed = datasets.load_gnn_benchmark_npz(path, directed=True)[0]
    assert np.asarray(directed.indices).shape == (A.nnz, 2)
    bad = dict(np.load(path))
    bad["adj_matrix.indptr"] = bad["adj_matrix.indptr"][:-1]
    np.savez(str(tmp_path / "bad.npz"), **bad)
    with pytest.raises(Exception, match="not a consistent CSR"):
        gnntf.load_npz(str(tmp_path / "bad.npz"))


def test_custom_splits_follow_the_reference_recipe():
    """experiment_setup.py:183-201 restated in the test: random.seed(seed); shuffle; first k per class train; the other labelled nodes
    shuffled again -> validation, test.  Same generator stream as the reference's module-level random under the same seed."""
    import random
    from gnntf.datasets import custom_splits
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 4, size=300)
    labels[rng.integers(0, 300, size=20)] = -1
    for seed, per_class, n_valid in ((0, 20, 50), (5, 7, None)):
        random.seed(seed)
        order = list(range(300))
        random.shuffle(order)
        count, want_train = {}, []
        for pos in order:
            if labels[pos] == -1:
                continue
            if count.get(labels[pos], 0) < per_class:
                want_train.append(pos)
                count[labels[pos]] = count.get(labels[pos], 0) + 1
        rest = list(set(pos for pos in range(300) if labels[pos] != -1) - set(want_train))
        random.shuffle(rest)
        k = n_valid if n_valid is not None else len(count) * per_class
        train, valid, test = custom_splits(labels, per_class, n_valid, seed)
        assert train == want_train and valid == rest[:k] and test == rest[k:]
        assert len(train) == 4 * per_class and not set(train) & set(valid) and not set(valid) & set(test)
        assert all(labels[i] != -1 for i in train + valid + test) and len(train + valid + test) == int((labels != -1).sum())


def test_row_widths_of_the_k_loops_fill_their_lines():
    """sparse.friendly_width: the width the K-iteration loops run at -- 16-byte aligned rows, and no row touching more 128-byte lines
    than its size needs (lines_per_row: average over the start offsets back-to-back rows take)."""
    from gnntf.sparse import friendly_width, lines_per_row
    n = 1 << 20
    assert [friendly_width(C, n) for C in (1, 6, 7, 8, 9, 16, 17, 24, 32)] == [1, 6, 8, 8, 16, 16, 32, 32, 32]
    assert lines_per_row(8) == 1.0 and lines_per_row(32) == 1.0 and lines_per_row(24) == 1.5        # 96-byte rows: every other one straddles
    assert lines_per_row(40) == 2.0 and lines_per_row(48) == 2.0 and lines_per_row(64) == 2.0        # 160 / 192 / 256 bytes: never a third line
    assert lines_per_row(44) == 2.25 and lines_per_row(56) == 2.5 and lines_per_row(60) == 2.75
    assert [friendly_width(C, n) for C in (33, 36, 40, 41, 44, 47, 48, 49, 52, 56, 60, 63, 64)] == [36, 36, 40, 48, 48, 48, 48, 64, 64, 64, 64, 64, 64]
    assert [friendly_width(C, n) for C in (65, 72, 96, 100, 120, 128, 129, 256, 300)] == [68, 72, 96, 100, 128, 128, 132, 256, 304]
    for C in range(1, 400):
        Cp = friendly_width(C, n)
        assert Cp >= C and (C <= 6 or Cp % 4 == 0) and Cp <= max(32, (C + 31) // 32 * 32)
        if C > 6:
            assert lines_per_row(Cp) <= lines_per_row((C + 3) // 4 * 4) + 1e-12                      # never more lines than the plain multiple of 4
    assert friendly_width(56, 1000) == 56                                                              # small graphs: launch-bound, no padding
