"""networkx <-> SparseCOO, the host end of A0.

Behavioural contract: reference gnntf/core/gnn/graph_manipulation.py:5-31 -- vertices are numbered in the graph's iteration
order, the adjacency is an UNSORTED COO holding one entry per stored edge with its "weight" attribute (1.0 when absent), and an
undirected reading (the default) APPENDS the reversed edge list, so a DiGraph that already stores both directions yields every
entry twice (the device path sums duplicates like TensorFlow's sparse ops do).  The structure is this build's own: one pass over
``edges(data=True)`` into numpy arrays, no Python list of index pairs.
"""
from __future__ import annotations

import numpy as np

from .sparse import SparseCOO


def _numbered_edges(G):
    """(int64 [m, 2] vertex numbers, float32 [m] weights) of G's stored edges, in G's edge order."""
    number = dict(zip(G, range(len(G))))
    m = G.number_of_edges()
    pairs, weights = np.empty((m, 2), dtype=np.int64), np.empty(m, dtype=np.float32)
    for k, (u, v, attributes) in enumerate(G.edges(data=True)):
        pairs[k, 0], pairs[k, 1] = number[u], number[v]
        weights[k] = attributes.get("weight", 1.)
    return pairs, weights


def graph2indices(G):
    """[[row, col], ...] of G's stored edges (graph_manipulation.py:19-21)."""
    return _numbered_edges(G)[0].tolist()


def graph2adj(G, directed=False):
    """The adjacency of a networkx graph as a SparseCOO (graph_manipulation.py:24-31)."""
    pairs, weights = _numbered_edges(G)
    if not directed:                                   # symmetrise by appending: duplicates are kept on purpose
        pairs, weights = np.concatenate([pairs, pairs[:, ::-1]]), np.concatenate([weights, weights])
    return SparseCOO(pairs, weights, (len(G), len(G)))


def create_nx_graph(nodes, edges):
    """A DiGraph over ``nodes`` (may be None) with one arc per listed pair (graph_manipulation.py:5-12)."""
    import networkx as nx
    graph = nx.DiGraph()
    graph.add_nodes_from(() if nodes is None else nodes)
    graph.add_edges_from(map(tuple, edges))
    return graph


def adj2graph(nodes, adj):
    """Back from a SparseCOO to networkx (graph_manipulation.py:15-16)."""
    return create_nx_graph(nodes, np.asarray(adj.indices.cpu() if hasattr(adj.indices, "cpu") else adj.indices))
