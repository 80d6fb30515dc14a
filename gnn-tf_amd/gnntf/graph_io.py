"""networkx <-> SparseCOO.  Mirrors reference gnntf/core/gnn/graph_manipulation.py:5-31:
the adjacency is an unsorted COO, undirected graphs are symmetrised by APPENDING the
reversed edge list (so a DiGraph that already stores both directions yields every entry
twice -- the device path sums duplicates like TensorFlow does)."""
from __future__ import annotations

import numpy as np

from .sparse import SparseCOO


def create_nx_graph(nodes, edges):
    import networkx as nx
    graph = nx.DiGraph()
    if nodes is not None:
        graph.add_nodes_from(nodes)
    graph.add_edges_from((u, v) for u, v in edges)
    return graph


def adj2graph(nodes, adj):
    return create_nx_graph(nodes, adj.indices.cpu().numpy())


def graph2indices(G):
    node2id = {u: idx for idx, u in enumerate(G)}
    return [[node2id[u], node2id[v]] for u, v in G.edges()]


def graph2adj(G, directed=False):
    pairs = np.asarray(graph2indices(G), dtype=np.int64).reshape(-1, 2)
    weights = np.asarray([data.get("weight", 1.) for _, _, data in G.edges(data=True)], dtype=np.float32)
    if not directed:
        pairs = np.concatenate([pairs, pairs[:, ::-1]], axis=0)
        weights = np.concatenate([weights, weights])
    return SparseCOO(pairs, weights, (len(G), len(G)))
