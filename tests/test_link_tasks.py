"""Host-side logic of the link-prediction tasks, the ranking metrics and the NGCF building blocks (reference
gnntf/core/gnn/graph_predictor.py:34-203, gnntf/measures.py:17-45, gnntf/core/gnn/gnn.py:5-26) -- everything that does
not need the GPU, on CPU tensors."""
import random

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


def small_graph():
    G = nx.Graph()
    G.add_nodes_from(range(30))
    rng = np.random.default_rng(0)
    for u, v in rng.integers(0, 30, size=(70, 2)):
        if u != v:
            G.add_edge(int(u), int(v))
    return G


def test_ranking_metrics_against_definitions():
    rng = np.random.default_rng(1)
    for _ in range(20):
        labels = rng.integers(0, 2, size=40)
        labels[:2] = [0, 1]
        scores = np.round(rng.random(40), 1)                                  # many ties
        assert abs(gnntf.auc(labels, scores) - orc.auc_by_pairs(labels, scores)) < 1e-12
    from sklearn import metrics as skm
    fpr, tpr, _ = skm.roc_curve(labels, scores, pos_label=1)                  # what the reference integrates (measures.py:17-19)
    assert abs(gnntf.auc(labels, scores) - skm.auc(fpr, tpr)) < 1e-12
    labels = np.array([1, 0, 1, 0, 0, 1]); scores = np.array([.9, .8, .7, .1, .2, .3])
    assert gnntf.prec(labels, scores, 2) == 0.5 and gnntf.rec(labels, scores, 2) == pytest.approx(1 / 3)
    assert gnntf.f1(labels, scores, 2) == pytest.approx(0.4) and gnntf.avprec(labels, scores, 3) == pytest.approx((1 + 1 / 3) / 2)
    assert gnntf.avprec(np.zeros(4), np.arange(4.), 2) == 0 and gnntf.f1(np.array([1, 0, 0, 0]), np.arange(4.), 2) == 0


def test_negative_sampling_and_recommend_all():
    random.seed(3)
    G = small_graph()
    positive = [list(e) for e in list(G.edges())[:12]]
    sampler = gnntf.negative_sampling(positive, G, samples=2)
    edges, labels = sampler()
    assert edges.shape == (36, 2) and labels.tolist() == [1., 0., 0.] * 12
    for i, (u, v) in enumerate(positive):
        assert edges[3 * i].tolist() == [u, v]
        for s in (1, 2):
            a, w = edges[3 * i + s]
            assert a == u and w not in (u, v) and not G.has_edge(u, w)
    again, _ = sampler()
    assert again is edges                                                      # the reference re-fills one array in place
    pooled = gnntf.negative_sampling(positive, G, samples=1, negative_nodes=list(range(15)), pool=4)
    e2, _ = pooled()
    assert set(e2[1::2, 1]) <= set(range(15))
    cand, lab = gnntf.recommend_all(positive[0][0], graph=G)
    node = positive[0][0]
    assert sum(lab) == G.degree(node) and len(cand) == len(lab)
    assert all(not G.has_edge(node, v) for (_, v), l in zip(cand, lab) if l == 0)


def test_link_prediction_on_cpu_tensors():
    rng = np.random.default_rng(5)
    F = torch.from_numpy(rng.standard_normal((30, 6)).astype(np.float32))
    edges = rng.integers(0, 30, size=(20, 2))
    labels = rng.integers(0, 2, size=20).astype(np.float32)
    labels[:2] = [0, 1]
    for sim in ("dot", "cos"):
        task = gnntf.LinkPrediction(edges, labels, similarity=sim, loss="diff")
        z = task.predict(F, to_logits=True).numpy()
        np.testing.assert_allclose(z, orc.link_logits(F.numpy(), edges, similarity=sim), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(task.predict(F).numpy(), 1 / (1 + np.exp(-z)), rtol=1e-5)
        assert abs(float(task.loss(F)) - orc.link_loss_diff(F.numpy(), edges, similarity=sim)) < 1e-5
        bce = gnntf.LinkPrediction(edges, labels, similarity=sim, loss="bce")
        assert abs(float(bce.loss(F)) - orc.link_loss_bce(F.numpy(), edges, labels, similarity=sim)) < 1e-5
        assert abs(task.evaluate(F) - orc.auc_by_pairs(labels, 1 / (1 + np.exp(-z.astype(np.float64))))) < 1e-6
    arch = gnntf.Layered((30, 6))
    with_r = gnntf.LinkPrediction(edges, labels, gnn=arch)                     # shared DistMult weights (graph_predictor.py:112)
    assert with_r.r.shape == (6, 1) and gnntf.LinkPrediction(edges, labels, gnn=arch).r is with_r.r
    arch.reset()
    assert float(with_r.r.detach().min()) == 1.0                               # "ones"
    np.testing.assert_allclose(with_r.predict(F, to_logits=True).detach().numpy(), orc.link_logits(F.numpy(), edges), rtol=1e-5, atol=1e-6)


def test_mean_link_prediction_report(capsys):
    random.seed(0)
    G = small_graph()
    held_out = np.array([list(e) for e in list(G.edges())[:10]])
    F = torch.from_numpy(np.random.default_rng(2).standard_normal((30, 5)).astype(np.float32))
    task = gnntf.MeanLinkPrediction(held_out, graph=G, k=3, positive_nodes=[int(held_out[0][0])])
    value = task.evaluate(F)
    assert 0 <= value <= 1 and "per-node ranking" in capsys.readouterr().out
    with pytest.raises(Exception, match="Node not found"):
        gnntf.MeanLinkPrediction(held_out, graph=G, positive_nodes=[max(G) + 7]).evaluate(F)


def test_structural_and_ngcf_shapes_on_cpu():
    arch = gnntf.Layered((12, 0))
    s = arch.add(gnntf.Structural(dims=4, bipartite=5, regularize=0))
    assert s.output_shape == (12, 4) and [tuple(v.var.shape) for v in arch.vars()] == [(5, 4), (7, 4)]
    arch.reset()
    out = arch(torch.zeros(12, 0))
    assert out.shape == (12, 4) and torch.equal(out[:5], s.embeddings) and torch.equal(out[5:], s.embeddings2)
    arch2 = gnntf.Layered((12, 3))
    s2 = arch2.add(gnntf.Structural(dims=2, l2_contraint=True))
    arch2.reset()
    out2 = arch2(torch.ones(12, 3))
    assert out2.shape == (12, 5) and torch.allclose(out2[:, :2].norm(dim=1), torch.ones(12), atol=1e-6) and bool((out2[:, 2:] == 1).all())
