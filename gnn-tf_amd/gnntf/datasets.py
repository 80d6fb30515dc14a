"""Local dataset files in place of the reference's DGL download helper
(reference experiments/experiment_setup.py:153-181, whose returned tuple layout
``(G, labels, features, train, valid, test)`` is kept).  There is no network in the build or on the
GPU box, so data comes from a pre-converted ``.npz``:

    indices  int   [nnz, 2]  directed (row, col) pairs as graph2adj would store them
    values   float [nnz]     optional (default 1.0)
    shape    int   [2]       optional (default: max index + 1, square)
    features float [N, F]
    labels   int   [N]
    train, valid, test  int index arrays
"""
from __future__ import annotations

import numpy as np

from .sparse import SparseCOO


def load_npz(path):
    """Returns (adjacency SparseCOO, labels, features, train, valid, test)."""
    z = np.load(path)
    idx = np.asarray(z["indices"], dtype=np.int64).reshape(-1, 2)
    vals = np.asarray(z["values"], dtype=np.float32) if "values" in z else np.ones(len(idx), dtype=np.float32)
    n = int(idx.max()) + 1 if len(idx) else 0
    shape = tuple(int(x) for x in z["shape"]) if "shape" in z else (max(n, len(z["labels"])),) * 2
    return (SparseCOO(idx, vals, shape), np.asarray(z["labels"]), np.asarray(z["features"], dtype=np.float32),
            z["train"].tolist(), z["valid"].tolist(), z["test"].tolist())


def save_npz(path, G, labels, features, train, valid, test):
    """Converts the reference's ``dgl_setup`` tuple (with a networkx graph) into that file."""
    from .graph_io import graph2adj
    adj = G if isinstance(G, SparseCOO) else graph2adj(G)
    np.savez_compressed(path, indices=adj.indices.cpu().numpy(), values=adj.values.cpu().numpy(), shape=np.asarray(adj.dense_shape),
                        features=np.asarray(features, dtype=np.float32), labels=np.asarray(labels), train=np.asarray(train),
                        valid=np.asarray(valid), test=np.asarray(test))
