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

or from the gnn-benchmark ``.npz`` the reference itself reads (experiment_setup.py:273-282): ``adj_matrix.{data,indices,indptr,shape}``,
``attr_matrix.{data,indices,indptr,shape}``, ``labels`` -- load_npz recognises it by its keys.  The attribute matrix stays sparse (a
SparseCOO the model turns into device SparseRows: Cora's 2708 x 1433 at 1.3 % density is never densified); the splits come from
custom_splits with the reference's defaults, as experiment_setup.py:49 draws them.
"""
from __future__ import annotations

import numpy as np

from .sparse import SparseCOO


def _csr_entries(z, name):
    """(int64 [nnz, 2] (row, col), float32 [nnz], shape) of the CSR stored under ``name``.* -- row by row, a row's entries in stored order."""
    data, indices, indptr = np.asarray(z[name + ".data"]), np.asarray(z[name + ".indices"]), np.asarray(z[name + ".indptr"])
    shape = tuple(int(x) for x in z[name + ".shape"])
    if len(indptr) != shape[0] + 1 or len(data) != len(indices) or int(indptr[-1]) != len(indices):
        raise Exception(f"load_npz: {name} is not a consistent CSR")
    rows = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(indptr))
    return np.stack([rows, indices.astype(np.int64)], 1), data.astype(np.float32), shape


def custom_splits(labels, examples_per_class=20, num_validation=500, seed=0):
    """experiment_setup.py:183-201: in a seeded shuffle of the nodes, the first ``examples_per_class`` of every class (label -1 =
    unlabelled, skipped) are the training nodes; the other labelled nodes, shuffled again, give ``num_validation`` validation
    nodes (None: as many as there are training nodes) and the test nodes.  Returns three index lists.  The shuffles are Python's
    ``random`` under ``seed`` as in the reference, so the same file and seed give the same lists."""
    import random
    labels = np.asarray(labels)
    rng = random.Random(seed)            # (the reference seeds the module-level generator: same stream, no global side effect here)
    order = list(range(labels.shape[0]))
    rng.shuffle(order)
    taken, training_idx = dict(), list()
    for pos in order:
        label = labels[pos].item()
        if label == -1:
            continue
        if taken.get(label, 0) < examples_per_class:
            training_idx.append(pos)
            taken[label] = taken.get(label, 0) + 1
    test_idx = list(set(pos for pos in range(labels.shape[0]) if labels[pos] != -1) - set(training_idx))
    rng.shuffle(test_idx)
    if num_validation is None:
        num_validation = len(taken) * examples_per_class
    return training_idx, test_idx[:num_validation], test_idx[num_validation:]


def load_gnn_benchmark_npz(path, directed=False, examples_per_class=20, num_validation=500, seed=0):
    """The gnn-benchmark file the reference reads (experiment_setup.py:273-282).  The reference turns ``adj_matrix`` into a DiGraph
    (one arc per stored entry, weight = the stored value) and graph2adj then APPENDS the reversed arcs (graph_manipulation.py:24-31):
    the adjacency returned here is that unsorted COO -- for a file that stores both directions every entry twice, which the device
    path sums like TensorFlow's sparse ops (and the symmetric normalisation does not see: SURVEY.md 3.4).  ``attr_matrix`` is
    returned as a SparseCOO, not densified.  Returns (adjacency, labels, features, train, valid, test)."""
    z = np.load(path, allow_pickle=True)
    pairs, weights, shape = _csr_entries(z, "adj_matrix")
    if shape[0] != shape[1]:
        raise Exception("load_npz: adj_matrix is not square")
    if not directed:
        pairs, weights = np.concatenate([pairs, pairs[:, ::-1]]), np.concatenate([weights, weights])
    f_pairs, f_vals, f_shape = _csr_entries(z, "attr_matrix")
    labels = np.asarray(z["labels"])
    if f_shape[0] != shape[0] or labels.shape[0] != shape[0]:
        raise Exception("load_npz: adj_matrix, attr_matrix and labels disagree about the number of nodes")
    train, valid, test = custom_splits(labels, examples_per_class, num_validation, seed)
    return SparseCOO(pairs, weights, shape), labels, SparseCOO(f_pairs, f_vals, f_shape), train, valid, test


def load_npz(path):
    """Returns (adjacency SparseCOO, labels, features, train, valid, test) from either file layout of the module docstring."""
    z = np.load(path, allow_pickle=True)
    if "adj_matrix.data" in z.files:
        return load_gnn_benchmark_npz(path)
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
