"""Link-prediction tasks over node embeddings: edge samplers, LinkPrediction and its ranking evaluation.

API and behaviour follow reference gnntf/core/gnn/graph_predictor.py:34-203 (same class names, constructor arguments, label
layout of the samplers -- one positive followed by ``samples`` negatives -- loss definitions and evaluation metrics); the
structure is this build's own.  On device embeddings the edge logits come from ONE launch of the link-head kernel
(gnx_edge_scores: gather of both endpoint rows + product + optional DistMult weights + reduction); CPU embeddings (host-logic
tests of the protocol) use torch ops."""
from __future__ import annotations

import random

import numpy as np
import torch

from . import metrics, sparse
from .training import Predictor


def _linked(graph, u, v):
    return graph.has_edge(u, v) or graph.has_edge(v, u)


def recommend_all(node, graph=None, positive_edges=None, negative_nodes=None):
    """Every candidate edge of ``node`` with its 0/1 label (graph_predictor.py:34-49): the positive edges that touch the node,
    then (node, v) for every candidate v that is not linked to it."""
    if positive_edges is None:
        positive_edges = [[node, neighbor] for neighbor in graph.neighbors(node)]
    candidates = list(graph) if negative_nodes is None else negative_nodes
    touching = [[u, v] for u, v in positive_edges if node in (u, v)]
    strangers = [[node, v] for v in candidates if v != node and (graph is None or not _linked(graph, node, v))]
    return np.array(touching + strangers), [1] * len(touching) + [0] * len(strangers)


class negative_sampling:
    """Callable edge sampler (graph_predictor.py:52-98): each call returns (edges, labels) where every positive edge (u, v) is
    followed by ``samples`` freshly drawn corrupted edges (u, v') with v' neither u, v nor a neighbour of u; labels are
    1, 0, ..., 0 per group.  ``pool``: draw the negatives of a source node from a fixed pre-sampled pool of that size."""

    def __init__(self, positive_edges, graph, samples=1, negative_nodes=None, pool=None):
        self.positive_edges, self.graph, self.samples, self.pool = positive_edges, graph, samples, pool
        self.negative_nodes = list(graph) if negative_nodes is None else negative_nodes
        group = 1 + samples
        self.labels = np.tile(np.array([1.] + [0.] * samples), len(positive_edges))
        self.edges = np.full((group * len(positive_edges), 2), -1, dtype=int)
        for i, (u, v) in enumerate(positive_edges):
            self.edges[i * group:(i + 1) * group, 0] = u
            self.edges[i * group, 1] = v
        self._negative_pool = None
        if pool is not None:
            self._negative_pool = {u: [self._draw(u, None, self.negative_nodes) for _ in range(pool)]
                                   for u in set(u for u, _ in positive_edges)}

    def _draw(self, u, v, candidates):
        while True:
            w = random.choice(candidates)
            if w != u and w != v and not _linked(self.graph, u, w):
                return w

    def __call__(self):
        group = 1 + self.samples
        for i, (u, v) in enumerate(self.positive_edges):
            candidates = self.negative_nodes if self._negative_pool is None else self._negative_pool[u]
            for s in range(1, group):
                self.edges[i * group + s, 1] = self._draw(u, v, candidates)
        return self.edges, self.labels


def _edge_logits(features, edges, r):
    if features.is_cuda:
        return sparse.edge_scores(features, edges, r)
    e = torch.as_tensor(np.asarray(edges), dtype=torch.int64)
    prod = features[e[:, 0]] * features[e[:, 1]]
    return prod.sum(dim=1) if r is None else (prod @ r).reshape(-1)


class LinkPrediction(Predictor):
    """graph_predictor.py:101-151.  ``edges`` is an [m, 2] array or a sampler (called again before every use);
    ``similarity`` "dot" or "cos"; ``loss`` "diff" (pairwise: -mean log sigmoid(logit_even - logit_odd), for samplers with one
    negative per positive) or anything else for binary cross entropy on the labels; ``gnn``: adds the shared DistMult weights."""

    def __init__(self, edges, labels=None, gnn=None, similarity="dot", loss="diff", regularize=0, batch_size=float('inf')):
        self.edge_sampler = edges if callable(edges) else None
        if self.edge_sampler is not None:
            edges, labels = self.edge_sampler()
        self.batch_size = batch_size
        self.edges = np.array(edges)
        self.loss_func = loss
        self.labels = None if labels is None else np.asarray(labels, dtype=np.float32).reshape(-1, 1)
        self.r = None if gnn is None else gnn.create_var(shape=(gnn.top_shape()[1], 1), regularize=0, shared_name="distmult",
                                                          normalization="ones", trainable=True)
        self.similarity = similarity
        self.regularize = regularize

    def _update_labels(self):
        if self.edge_sampler is not None:
            edges, labels = self.edge_sampler()
            self.edges = edges
            self.labels = None if labels is None else np.asarray(labels, dtype=np.float32).reshape(-1, 1)

    def _embed(self, features):
        return torch.nn.functional.normalize(features, dim=1, eps=1e-12) if self.similarity == "cos" else features

    def predict(self, features, to_logits=False):
        self._update_labels()
        logits = _edge_logits(self._embed(features), self.edges, self.r)
        return logits if to_logits else torch.sigmoid(logits)

    def loss(self, features):
        self._update_labels()
        features = self._embed(features)
        if self.loss_func == "diff":
            edges = self.edges
            take = min(self.batch_size, len(edges))
            if take != len(edges):
                edges = edges[random.sample(range(len(edges)), take), :]
            logits = _edge_logits(features, edges, self.r)
            return -torch.nn.functional.logsigmoid(logits[0::2] - logits[1::2]).mean()
        logits = _edge_logits(features, self.edges, self.r)
        target = torch.as_tensor(self.labels, dtype=torch.float32, device=logits.device).reshape(-1)
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, target)

    def evaluate(self, features):
        self._update_labels()
        return metrics.auc(self.labels.reshape(-1), self.predict(features))


class MeanLinkPrediction(LinkPrediction):
    """graph_predictor.py:154-203: per-node ranking quality -- for every positive node, its held-out edges against all
    non-linked candidates; prints mean AUC / MAP / precision / recall / F1 at k and the coverage of the top-k lists,
    returns the mean F1."""

    def __init__(self, *args, graph, positive_nodes=None, negative_nodes=None, k=5, **kwargs):
        super().__init__(*args, **kwargs)
        self.positive_nodes, self.negative_nodes, self.k, self.graph = positive_nodes, negative_nodes, k, graph
        self.parsed_edges = dict()
        for u, v in self.edges:
            self.parsed_edges.setdefault(u, list())
            self.parsed_edges.setdefault(v, list())
            self.parsed_edges[u].append(v)
            self.parsed_edges[v].append(u)

    def evaluate(self, features):
        k = self.k
        sources = list(self.parsed_edges) if self.positive_nodes is None else self.positive_nodes
        # default candidates: every node that appears in the held-out edges.  (The reference iterates the dict's KEYS here,
        # graph_predictor.py:178, which raises TypeError for integer nodes; the neighbour lists are what was meant.)
        candidates = set(v for neighbors in self.parsed_edges.values() for v in neighbors) if self.negative_nodes is None \
            else set(self.negative_nodes)
        scores = {name: [] for name in ("auc", "avprec", "prec", "rec", "f1")}
        recommended = set()
        for node in sources:
            if node not in self.parsed_edges:
                raise Exception("Node not found")
            held_out = [[node, v] for v in self.parsed_edges[node]]
            strangers = [[node, v] for v in candidates if v != node and not _linked(self.graph, node, v)]
            self.labels = np.array([1.] * len(held_out) + [0] * len(strangers))
            self.edges = np.array(held_out + strangers)
            prediction = metrics._np(self.predict(features))
            scores["auc"].append(metrics.auc(self.labels, prediction))
            for name in ("avprec", "prec", "rec", "f1"):
                scores[name].append(getattr(metrics, name)(self.labels, prediction, k))
            recommended.update(self.edges[i][1] for i in np.argsort(prediction)[-k:])
        mean = {name: float(np.mean(v)) for name, v in scores.items()}
        print("per-node ranking: AUC %.3f  MAP %.3f  precision %.3f  recall %.3f  F1 %.3f  coverage %.3f"
              % (mean["auc"], mean["avprec"], mean["prec"], mean["rec"], mean["f1"], len(recommended) / len(candidates)))
        return np.mean(scores["f1"])
