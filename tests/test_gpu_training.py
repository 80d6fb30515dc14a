"""One whole training step of APPNP on the HIP path -- loss and every parameter gradient -- against an independent dense
float64 computation (torch autograd on the CPU over the oracle's dropped + re-normalised adjacencies of each iteration)."""
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


class FixedMasks:
    """Feature-dropout masks drawn once on the host, handed out in call order: the same masks drive the HIP model
    (patched into Layered.dropout) and the dense reference, so the two computations are comparable bit for bit."""

    def __init__(self, seed):
        self.rng, self.masks, self.cursor = np.random.default_rng(seed), [], 0

    def mask(self, shape, p):
        if self.cursor == len(self.masks):
            self.masks.append((self.rng.random(shape) >= p).astype(np.float64) / (1.0 - p))
        m = self.masks[self.cursor]
        self.cursor += 1
        return m

    def rewind(self):
        self.cursor = 0


@pytest.mark.parametrize("fused", [False, True])
def test_whole_model_training_step_matches_dense_float64(gnntf, fused):
    """trainable.py:69-79: training-mode forward (input Dropout 0.5, Dense(relu, dropout 0.6), Dense, K x PPRIteration with
    per-iteration edge dropout 0.5 + renormalisation), CE + L2, backward: compare loss, dW1, db1, dW2, db2."""
    n, F, hidden, classes, K, a, wd = 600, 40, 16, 5, 10, 0.1, 5e-4
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 4000, seed=3)
    coo = np.concatenate([coo, coo[:300]])                                        # stored duplicates: dropout acts per stored entry
    vals = np.concatenate([vals, np.full(300, 0.5, dtype=np.float32)])
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, F)).astype(np.float32)
    labels = rng.integers(0, classes, size=n)
    train = np.arange(0, 200)
    gnntf.set_seed(17)
    model = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), X, num_classes=classes, latent_dims=[hidden], iterations=K, a=a, fused=fused)
    model.reset()
    dense = [l for l in model.layers() if isinstance(l, gnntf.Dense)]
    params = [dense[0].W, dense[0].b, dense[1].W, dense[1].b]
    with torch.no_grad():
        dense[0].b.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, size=(1, hidden)).astype(np.float32)))
        dense[1].b.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, size=(1, classes)).astype(np.float32)))
    masks = FixedMasks(23)
    model.dropout = lambda feats, p=0.5: feats if (not model.is_training() or p == 0) else \
        feats * torch.from_numpy(masks.mask(tuple(feats.shape), p)).to(feats.device, feats.dtype)
    task = gnntf.NodeClassification(train, labels[train])
    from gnntf.training import _Objective
    from gnntf import metrics
    seed, first_stream = metrics.current_seed(), model._mask_calls
    with model:
        loss = _Objective(model, task, wd)()
        loss.backward()
    got = [float(loss)] + [p.grad.cpu().numpy().astype(np.float64) for p in params]

    # ---- the same step in dense float64 on the CPU -------------------------------------------------------------------------
    masks.rewind()
    W1, b1, W2, b2 = [torch.tensor(p.detach().cpu().numpy().astype(np.float64), requires_grad=True) for p in params]
    H = torch.from_numpy(X.astype(np.float64)) * torch.from_numpy(masks.mask((n, F), 0.5))              # Dropout(0.5), filter.py:30
    H = torch.relu(H @ W1 + b1)
    H = H * torch.from_numpy(masks.mask((n, hidden), 0.6))                                              # Dense(..., dropout=0.6), filter.py:31-32
    H0 = H @ W2 + b2                                                                                    # Dense(num_classes) has dropout 0, filter.py:33
    Hk = H0
    for k in range(K):                                                                                  # filter.py:17-22, a fresh adjacency per iteration
        ai, av = orc.get_adjacency(coo, vals, shape, graph_dropout=0.5, training=True, seed=seed, stream=first_stream + k, dtype=np.float64)
        A = torch.from_numpy(orc.to_dense(ai, av, shape, dtype=np.float64))
        Hk = (A @ Hk) * (1 - a) + H0 * a
    logp = torch.log_softmax(Hk[torch.from_numpy(train)], dim=1)
    want_loss = torch.nn.functional.cross_entropy(logp, torch.from_numpy(labels[train]))               # graph_predictor.py:24-25
    want_loss = want_loss + wd * ((W1 ** 2).sum() / 2 + (b1 ** 2).sum() / 2)                            # regularize=False on the output Dense (filter.py:33)
    want_loss.backward()
    want = [float(want_loss)] + [t.grad.numpy() for t in (W1, b1, W2, b2)]
    assert abs(got[0] - want[0]) <= 1e-5 * abs(want[0]), (got[0], want[0])
    for name, g_, w_ in zip(("dW1", "db1", "dW2", "db2"), got[1:], want[1:]):
        scale = np.abs(w_).max()
        np.testing.assert_allclose(g_, w_, rtol=1e-3, atol=2e-5 * scale, err_msg=name)
    assert np.abs(want[1]).max() > 1e-4 and np.abs(want[3]).max() > 1e-4                                # the gradients are not trivially zero


def test_captured_training_equals_eager(gnntf):
    """train(capture=True): one training step + one validation forward recorded as hipGraphs and replayed per epoch.  With edge
    dropout as the only randomness (counter RNG: the captured launches keep their stream ids, a device counter advances them
    per replay) the captured run must follow the eager run: same masks, same parameters up to the optimizer's float32 rounding."""
    coo, vals, shape = graphs.rmat_symmetric_coo(800, 6000, seed=4)
    rng = np.random.default_rng(4)
    X = rng.standard_normal((800, 20)).astype(np.float32)
    labels = rng.integers(0, 4, size=800)
    train, valid = np.arange(0, 200), np.arange(200, 400)

    def build():
        gnntf.set_seed(11)
        torch.manual_seed(3)
        model = gnntf.GNN(gnntf.SparseCOO(coo, vals, shape), X)
        model.add(gnntf.Dense(16, activation=gnntf.relu))                       # no feature dropout: torch's generator stays out of it
        H0 = model.add(gnntf.Dense(4, regularize=False))
        for _ in range(4):
            model.add(gnntf.PPRIteration(H0, 0.1, graph_dropout=0.5))
        return model

    results = []
    for capture in (False, True):
        model = build()
        torch.manual_seed(5)                                                    # reset() draws the initial weights from here
        model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]),
                    epochs=12, patience=50, capture=capture)
        results.append([v.var.detach().cpu().numpy().copy() for v in model.vars()] + [model._mask_calls])
    assert results[0][-1] == results[1][-1] == 12 * 4                           # 4 edge-dropout masks per step, 12 steps
    for eager, captured in zip(results[0][:-1], results[1][:-1]):
        np.testing.assert_allclose(captured, eager, rtol=2e-3, atol=2e-5)
    # the reference model with all its dropouts: trains, predicts, and a sampler-driven task is refused with a clear message
    for fused in (False, True):                   # the reference's layer list (fused at execution) and the collapsed PPRLoop layer
        gnntf.set_seed(0)
        appnp = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), X, num_classes=4, fused=fused)
        appnp.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]),
                    epochs=20, patience=20, capture=True)
        assert appnp.predict(gnntf.NodeClassification(list(range(400, 800)))).shape[0] == 400 and not appnp.is_training()
        assert len(appnp.layers()) == (4 if fused else 13)
    # a task that draws new edges on the host at every call cannot be replayed: refused with a clear message, nothing left behind
    import networkx as nx
    G = nx.Graph(); G.add_nodes_from(range(800)); G.add_edges_from((int(u), int(v)) for u, v in coo if u < v)
    ngcf = gnntf.NGCF(gnntf.SparseCOO(coo, vals, shape), X, num_classes=8)
    sampler = gnntf.negative_sampling([list(e) for e in list(G.edges())[:200]], G)
    with pytest.raises(Exception, match="cannot be recorded as a device graph"):
        ngcf.train(train=gnntf.LinkPrediction(sampler), epochs=2, capture=True)
    ngcf.train(train=gnntf.LinkPrediction(sampler), epochs=2, patience=2)         # the eager loop still works afterwards
