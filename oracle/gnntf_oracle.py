"""CPU oracle: a numpy restatement of gnntf's propagation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnn-tf_amd/`` imports this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker.  The product path is the HIP library behind ``include/gnx.h``.

PARITY UNPINNED: the reference (/root/reference, gnntf 0.0.20) ships no golden vectors,
no known-answer tests and no fixtures for this path, and its
arithmetic lives in an un-vendored, un-pinned TensorFlow that is absent from this image,
so the reference itself cannot be run here.  This file restates the reference's Python
line by line and the documented TensorFlow op semantics it relies on; it is pinned by
closed-form known-answer tests (tests/test_oracle_kat.py) and by two independent
re-derivations (scipy CSR, dense float64), not by reference outputs.

Every function cites the reference file:line (relative to /root/reference) it follows.
All functions take/return numpy arrays; ``dtype`` selects float32 (what the reference
computes in) or float64 (for closed-form checks).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# Counter-based dropout RNG.
# TensorFlow's tf.nn.dropout stream (layered.py:45,50) cannot be reproduced (unpinned TF,
# stateful generator), so the build defines its own stateless generator; the HIP kernel
# in gnn-tf_amd/csrc/gnx_prep.hip implements the very same integer arithmetic, which makes
# training-mode masks bit-identical between this oracle and the GPU and independent of
# how the graph is sharded.  Only the distribution (Bernoulli(1-p), kept values scaled by
# 1/(1-p)) is reference behaviour.
# --------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_C_STREAM = np.uint64(0xD1342543DE82EF95)
_C_ROW = np.uint64(0x9E3779B97F4A7C15)
_C_COL = np.uint64(0xC2B2AE3D27D4EB4F)
_C_DUP = np.uint64(0x165667B19E3779F9)
_F1 = np.uint64(0xBF58476D1CE4E5B9)
_F2 = np.uint64(0x94D049BB133111EB)


def _fin(z):
    z = z ^ (z >> np.uint64(30))
    z = z * _F1
    z = z ^ (z >> np.uint64(27))
    z = z * _F2
    z = z ^ (z >> np.uint64(31))
    return z


def hash_u24(seed, stream, row, col, dup):
    """24-bit uniform integer per (seed, stream, row, col, dup); wraps mod 2**64."""
    with np.errstate(over="ignore"):
        seed = np.uint64(seed)
        stream = np.uint64(stream)
        row = np.asarray(row).astype(np.uint64)
        col = np.asarray(col).astype(np.uint64)
        dup = np.asarray(dup).astype(np.uint64)
        k = seed ^ (stream * _C_STREAM)
        x = _fin(k + row * _C_ROW)
        x = x ^ (col * _C_COL)
        x = _fin(x + dup * _C_DUP)
        return (x >> np.uint64(40)).astype(np.uint32)


def dropout_threshold(p: float) -> int:
    """Integer threshold: an entry is KEPT iff hash_u24 >= threshold."""
    return int(np.float64(p) * 16777216.0)


def duplicate_rank(indices):
    """Rank of every COO entry among the entries with the same (row, col), in input order."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    nnz = indices.shape[0]
    if nnz == 0:
        return np.zeros(0, dtype=np.int64)
    ncol = int(indices[:, 1].max()) + 1
    key = indices[:, 0] * ncol + indices[:, 1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    head = np.ones(nnz, dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    start = np.maximum.accumulate(np.where(head, np.arange(nnz), 0))
    rank_sorted = np.arange(nnz) - start
    rank = np.empty(nnz, dtype=np.int64)
    rank[order] = rank_sorted
    return rank


def keep_mask(indices, p, seed, stream):
    """Boolean keep mask per COO entry for edge dropout with rate p."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    u = hash_u24(seed, stream, indices[:, 0], indices[:, 1], duplicate_rank(indices))
    return u >= np.uint32(dropout_threshold(p))


# --------------------------------------------------------------------------------------
# A0  graph2adj  (gnntf/core/gnn/graph_manipulation.py:19-31)
# --------------------------------------------------------------------------------------
def graph2adj(nodes, edges, weights=None, directed=False):
    """COO adjacency as the reference builds it.

    ``nodes``: iterable of node ids in graph iteration order (graph_manipulation.py:20:
    position = enumerate(G)); ``edges``: iterable of (u, v) in G.edges() order (:21);
    ``weights``: per-edge "weight" attribute, default 1.0 (:27).  Undirected (default)
    appends the reversed list and repeats the values (:28-30).  The result is an
    UNSORTED COO that may hold duplicates (:31) -- e.g. a DiGraph that already stores
    both directions yields every entry twice.
    """
    node2id = {u: i for i, u in enumerate(nodes)}
    idx = [[node2id[u], node2id[v]] for u, v in edges]
    vals = [1.0] * len(idx) if weights is None else [float(w) for w in weights]
    if not directed:
        idx = idx + [[v, u] for u, v in idx]
        vals = vals + vals
    indices = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(vals, dtype=np.float32)
    n = len(node2id)
    return indices, values, (n, n)


# --------------------------------------------------------------------------------------
# A1  Layered.sparse_dropout  (gnntf/core/nn/layered.py:47-50)
# --------------------------------------------------------------------------------------
def sparse_dropout(indices, values, p, training, seed=0, stream=0):
    """values <- tf.nn.dropout(values, p): kept entries are scaled by 1/(1-p), dropped
    entries stay as explicit zeros; identity when p == 0 or not training (:48-49)."""
    values = np.asarray(values)
    if p == 0 or not training:
        return values
    keep = keep_mask(indices, p, seed, stream)
    scale = values.dtype.type(1.0) / (values.dtype.type(1.0) - values.dtype.type(p))
    return np.where(keep, values * scale, values.dtype.type(0))


# --------------------------------------------------------------------------------------
# TensorFlow op semantics the reference relies on
# --------------------------------------------------------------------------------------
def sparse_reduce_sum_axis0(indices, values, shape):
    """tf.sparse.reduce_sum(graph, axis=0): column sums d[j] = sum_i A[i, j]; duplicate
    entries add up (gnn.py:41,44)."""
    out = np.zeros(shape[1], dtype=values.dtype)
    np.add.at(out, np.asarray(indices)[:, 1], values)
    return out


def divide_no_nan(x, y):
    """tf.math.divide_no_nan: x / y, and 0 where y == 0 (gnn.py:41,44)."""
    y = np.asarray(y)
    out = np.zeros_like(y)
    np.divide(x, y, out=out, where=(y != 0))
    return out


def sparse_dense_matmul(indices, values, shape, H):
    """tf.sparse.sparse_dense_matmul(A, H): P[i, :] = sum over COO entries (i, j, v) of
    v * H[j, :]; accepts unsorted COO and sums duplicates; accumulation follows the COO
    entry order (filter.py:19, gcn.py:88)."""
    indices = np.asarray(indices)
    H = np.asarray(H)
    dt = np.result_type(values.dtype, H.dtype)
    P = np.zeros((shape[0], H.shape[1]), dtype=dt)
    np.add.at(P, indices[:, 0], values[:, None].astype(dt) * H[indices[:, 1]].astype(dt))
    return P


# --------------------------------------------------------------------------------------
# A2  GNN.get_adjacency  (gnntf/core/gnn/gnn.py:36-50)
# --------------------------------------------------------------------------------------
def _add_eye(indices, values, shape):
    """tf.sparse.add(graph, tf.sparse.eye(N)) (gnn.py:39,49): as a COO with the diagonal
    appended (duplicates are summed by every consumer, so this is equivalent)."""
    n = shape[0]
    eye = np.stack([np.arange(n, dtype=np.int64)] * 2, axis=1)
    return (np.concatenate([np.asarray(indices, dtype=np.int64).reshape(-1, 2), eye]),
            np.concatenate([values, np.ones(n, dtype=values.dtype)]))


def get_adjacency(indices, values, shape, graph_dropout=0.5, normalized="symmetric",
                  add_eye="none", training=False, seed=0, stream=0, dtype=np.float32):
    """Returns (indices, values) of the normalised adjacency.

    Order of operations exactly as gnn.py:37-50: edge dropout (:37) -> optional +I before
    (:38-39) -> symmetric: D = divide_no_nan(1, sqrt(colsum)), v_ij <- D[i] v_ij D[j]
    (:40-42) | bipartite: D = divide_no_nan(1, colsum), v_ij <- D[i] v_ij (:43-45) | none
    | anything else raises (:46-47) -> optional +I after (:48-49).  Note both forms use
    COLUMN sums (axis=0) for the row scaling too.
    """
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values).astype(dtype)
    values = sparse_dropout(indices, values, graph_dropout, training, seed, stream)
    if add_eye == "before":
        indices, values = _add_eye(indices, values, shape)
    if normalized == "symmetric":
        D = divide_no_nan(dtype(1.0), np.sqrt(sparse_reduce_sum_axis0(indices, values, shape)))
        values = D[indices[:, 0]] * values * D[indices[:, 1]]
    elif normalized == "bipartite":
        D = divide_no_nan(dtype(1.0), sparse_reduce_sum_axis0(indices, values, shape))
        values = D[indices[:, 0]] * values
    elif normalized != "none":
        raise Exception("Invalid matrix normalization")
    if add_eye == "after":
        indices, values = _add_eye(indices, values, shape)
    return indices, values.astype(dtype)


# --------------------------------------------------------------------------------------
# A3-A5  PPRIteration.__forward__  (gnntf/core/gnn/architectures/filter.py:17-22)
# --------------------------------------------------------------------------------------
def ppr_iteration(adj_indices, adj_values, shape, H, H0, a=0.1, activation=None):
    """One APPNP power-iteration step on an already normalised adjacency:
    propagated = A_hat . H (:19); out = propagated*(1-a) + H0*a (:20-21); then
    activation(dropout(out)) with the defaults identity / rate 0 (:22, :8)."""
    dt = H.dtype.type
    propagated = sparse_dense_matmul(adj_indices, adj_values, shape, H)
    out = propagated * dt(1 - a) + H0 * dt(a)
    return out if activation is None else activation(out)


def appnp_propagate(indices, values, shape, H0, a=0.1, iterations=10, graph_dropout=0.5,
                    normalized="symmetric", add_eye="none", training=False, seed=0,
                    first_stream=0, dtype=np.float32, activation=None):
    """The K-iteration hot loop (filter.py:34-35 adds K PPRIteration layers; each one
    calls get_adjacency again, filter.py:18) starting from H = H0 (the Dense layer's
    value feeds the first iteration, layered.py:52-55).  ``activation``: APPNP's
    argument of that name, handed to every PPRIteration (filter.py:28,35 -> :22)."""
    H0 = np.asarray(H0).astype(dtype)
    H = H0
    for k in range(iterations):
        ai, av = get_adjacency(indices, values, shape, graph_dropout, normalized, add_eye,
                               training, seed, first_stream + k, dtype)
        H = ppr_iteration(ai, av, shape, H, H0, a, activation)
    return H


def appnp_closed_form(dense_adj, H0, a, iterations):
    """KAT-1: H_K = (1-a)^K A^K H0 + a sum_{k<K} (1-a)^k A^k H0,
    evaluated with dense float64 algebra."""
    A = np.asarray(dense_adj, dtype=np.float64)
    H0 = np.asarray(H0, dtype=np.float64)
    term = H0.copy()
    acc = np.zeros_like(H0)
    for k in range(iterations):
        acc += a * (1 - a) ** k * term
        term = A @ term
    return acc + (1 - a) ** iterations * term


def to_dense(indices, values, shape, dtype=np.float64):
    A = np.zeros(shape, dtype=dtype)
    np.add.at(A, (np.asarray(indices)[:, 0], np.asarray(indices)[:, 1]), np.asarray(values).astype(dtype))
    return A


# --------------------------------------------------------------------------------------
# A7  Dense / Dropout  (gnntf/core/nn/layers.py:125-136, 175-181; layered.py:44-45)
# --------------------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, x.dtype.type(0))


def dense_forward(X, W, b, activation=None):
    """activation(X . W + b) (layers.py:136); the feature dropout around it is identity
    in eval mode (layered.py:45)."""
    out = X @ W + b
    return out if activation is None else activation(out)


# --------------------------------------------------------------------------------------
# A6  APPNP eval forward  (filter.py:27-35)   /   A8  GCN eval forward  (gcn.py:77-113)
# --------------------------------------------------------------------------------------
def appnp_forward_eval(indices, values, shape, X, weights, a=0.1, iterations=10, dtype=np.float32):
    """Dropout(0.5) [identity in eval] -> Dense(latent, relu) per latent dim ->
    Dense(num_classes) = H0 -> K x PPRIteration (filter.py:30-35).
    ``weights`` = [(W, b), ...] for the Dense stack, the last pair producing H0."""
    H = np.asarray(X).astype(dtype)
    for W, b in weights[:-1]:
        H = dense_forward(H, W.astype(dtype), b.astype(dtype), relu)
    W, b = weights[-1]
    H0 = dense_forward(H, W.astype(dtype), b.astype(dtype))
    return appnp_propagate(indices, values, shape, H0, a, iterations, training=False, dtype=dtype), H0


def gcn_forward_eval(indices, values, shape, X, weights, dtype=np.float32):
    """GCNLayer: relu((A_hat . X) . W + b), aggregation FIRST at the input width
    (gcn.py:87-89); every layer including the last keeps the relu (gcn.py:78,113)."""
    ai, av = get_adjacency(indices, values, shape, training=False, dtype=dtype)
    H = np.asarray(X).astype(dtype)
    for W, b in weights:
        agg = sparse_dense_matmul(ai, av, shape, H)
        H = relu(agg @ W.astype(dtype) + b.astype(dtype))
    return H


def gcnii_forward_eval(indices, values, shape, X, dense_in, conv_W, dense_out, a=0.1, l=0.5, dtype=np.float32):
    """GCNII in eval mode (gcn.py:54-74): Dropout [identity] -> Dense(relu) = H0 -> per layer k:
    relu(((1-a) A_hat.H + a H0) . ((1-b) I + b W_k)), b = log1p(l/(k+1)) (gcn.py:22-27) -> Dense."""
    ai, av = get_adjacency(indices, values, shape, training=False, dtype=dtype)
    W, b = dense_in
    H0 = dense_forward(np.asarray(X).astype(dtype), W.astype(dtype), b.astype(dtype), relu)
    H = H0
    for k, Wk in enumerate(conv_W):
        beta = dtype(np.log1p(l / (k + 1)))
        tradeoff = ppr_iteration(ai, av, shape, H, H0, a)
        H = relu(tradeoff @ ((dtype(1) - beta) * np.eye(Wk.shape[1], dtype=dtype) + beta * Wk.astype(dtype)))
    W, b = dense_out
    return dense_forward(H, W.astype(dtype), b.astype(dtype))


# --------------------------------------------------------------------------------------
# A10  NodeClassification  (gnntf/core/gnn/graph_predictor.py:10-31)
# --------------------------------------------------------------------------------------
def node_predict(logits, nodes):
    """argmax(logits[nodes], axis=1) (graph_predictor.py:16-17)."""
    return np.argmax(np.asarray(logits)[np.asarray(nodes)], axis=1).astype(np.int64)


def log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def node_loss(logits, nodes, labels):
    """SparseCategoricalCrossentropy(from_logits=True)(labels, log_softmax(logits[nodes]))
    (graph_predictor.py:24-25): softmax of a log-softmax is the softmax, so this is the
    plain mean cross entropy."""
    lp = log_softmax(log_softmax(np.asarray(logits)[np.asarray(nodes)]))
    return -lp[np.arange(len(nodes)), np.asarray(labels)].mean()


def node_evaluate(logits, nodes, labels):
    """1 - count_nonzero(pred - labels)/len (graph_predictor.py:27-31)."""
    pred = node_predict(logits, nodes)
    return 1 - np.count_nonzero(pred - np.asarray(labels)) / pred.shape[0]


def l2_loss(x):
    """tf.nn.l2_loss = sum(x**2)/2 (trainable.py:77, layered.py:86)."""
    return (np.asarray(x) ** 2).sum() / 2


# --------------------------------------------------------------------------------------
# Backward of one propagation step (what tf.GradientTape derives for filter.py:19-21);
# used to check the HIP transposed SpMM.
# --------------------------------------------------------------------------------------
def ppr_iteration_backward(adj_indices, adj_values, shape, grad_out, a=0.1):
    """Given dL/d(out) for out = (1-a) A H + a H0, returns (dL/dH, dL/dH0):
    dL/dH = (1-a) A^T g, dL/dH0 = a g."""
    idx = np.asarray(adj_indices)
    dt = grad_out.dtype.type
    t_idx = idx[:, ::-1]
    gH = sparse_dense_matmul(t_idx, adj_values, (shape[1], shape[0]), grad_out) * dt(1 - a)
    return gH, grad_out * dt(a)


# --------------------------------------------------------------------------------------
# Helpers shared by tests / fixtures: coalesced CSR (what the HIP path stores)
# --------------------------------------------------------------------------------------
def coo_to_csr_coalesced(indices, values, shape):
    """Sort by (row, col), sum duplicates in input order.  Returns rowptr int64[n+1],
    colidx int32[nnz_c], vals[nnz_c]."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    values = np.asarray(values)
    n, m = shape
    key = indices[:, 0] * m + indices[:, 1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    vs = values[order]
    if len(ks) == 0:
        return np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=values.dtype)
    head = np.ones(len(ks), dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    slot = np.cumsum(head) - 1
    vals = np.zeros(int(slot[-1]) + 1, dtype=values.dtype)
    np.add.at(vals, slot, vs)
    uk = ks[head]
    rows = uk // m
    cols = (uk % m).astype(np.int32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, cols, vals


# --------------------------------------------------------------------------------------
# (f4 tail)  NGCFLayer  (gnntf/core/gnn/architectures/gcn.py:116-135)  and the link head
#            (gnntf/core/gnn/graph_predictor.py:101-151)
# --------------------------------------------------------------------------------------
def leaky_relu(x, slope=0.2):
    """tf.nn.leaky_relu with its default alpha = 0.2 (gcn.py:117)."""
    return np.where(x > 0, x, x * x.dtype.type(slope))


def l2_normalize_rows(x, eps=1e-12):
    """tf.math.l2_normalize(x, axis=1): x / sqrt(max(sum(x^2), eps))."""
    return x / np.sqrt(np.maximum((x * x).sum(axis=1, keepdims=True), eps))


def ngcf_layer_eval(indices, values, shape, X, W1, b1, W2, b2, dtype=np.float64):
    """gcn.py:130-135 in eval mode: A = bipartite-normalised adjacency (gnn.py:43-45, taken at build time gcn.py:127);
    out = l2_normalize(leaky_relu((X * A.X).W1 + b1) + leaky_relu((A.X).W2 + b2))."""
    ai, av = get_adjacency(indices, values, shape, normalized="bipartite", training=False, dtype=dtype)
    X = np.asarray(X).astype(dtype)
    agg = sparse_dense_matmul(ai, av, shape, X)
    out = leaky_relu((X * agg) @ W1.astype(dtype) + b1.astype(dtype)) + leaky_relu(agg @ W2.astype(dtype) + b2.astype(dtype))
    return l2_normalize_rows(out)


def link_logits(features, edges, r=None, similarity="dot"):
    """graph_predictor.py:122-126: sum over columns of F[u] * F[v] (optionally through the DistMult weights r [C, 1])."""
    F = np.asarray(features)
    if similarity == "cos":
        F = l2_normalize_rows(F)
    edges = np.asarray(edges)
    prod = F[edges[:, 0]] * F[edges[:, 1]]
    return prod.sum(axis=1) if r is None else (prod @ np.asarray(r)).reshape(-1)


def link_loss_diff(features, edges, r=None, similarity="dot"):
    """graph_predictor.py:138-141: -mean(log_sigmoid(logit[0::2] - logit[1::2]))."""
    z = link_logits(features, edges, r, similarity)
    d = z[0::2] - z[1::2]
    return float(np.mean(np.log1p(np.exp(-np.abs(d))) + np.maximum(-d, 0)))


def link_loss_bce(features, edges, labels, r=None, similarity="dot"):
    """graph_predictor.py:142-146: BinaryCrossentropy(from_logits=True), mean over the edges."""
    z = link_logits(features, edges, r, similarity)
    y = np.asarray(labels, dtype=z.dtype).reshape(-1)
    return float(np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))))


def auc_by_pairs(labels, scores):
    """Area under the ROC curve from its definition: P(score+ > score-) + 0.5 P(tie) over all positive/negative pairs
    (what sklearn's roc_curve + auc integrate, measures.py:17-19)."""
    labels, scores = np.asarray(labels).reshape(-1), np.asarray(scores, dtype=np.float64).reshape(-1)
    pos, neg = scores[labels == 1], scores[labels != 1]
    wins = (pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()
    return float(wins / (len(pos) * len(neg)))
