"""The TensorFlow op semantics the oracle restates (the reference's arithmetic lives in an un-vendored, unpinned
`tensorflow`, setup.py:26-28), anchored on the worked examples and definitions of TensorFlow's public API documentation -- the only
published known answers that exist for this path.  Each case names the op, the documented statement it encodes and the reference
call site that depends on it."""
import numpy as np
import pytest

from oracle import gnntf_oracle as orc


def test_sparse_reduce_sum_documented_example():
    """tf.sparse.reduce_sum docs: x = [[1, ?, 1], [?, 1, ?]] (? implicitly zero) -> reduce_sum(x, 0) == [1, 1, 1]
    (and reduce_sum(x) == 3, reduce_sum(x, 1) == [2, 1]).  Used with axis=0 at gnn.py:41,44."""
    idx = np.array([[0, 0], [0, 2], [1, 1]])
    vals = np.array([1, 1, 1], dtype=np.float32)
    cols = orc.sparse_reduce_sum_axis0(idx, vals, (2, 3))
    assert cols.tolist() == [1, 1, 1] and cols.sum() == 3
    assert orc.sparse_reduce_sum_axis0(idx[:, ::-1], vals, (3, 2)).tolist() == [2, 1]        # axis 1 of x = axis 0 of its transpose


def test_divide_no_nan_documented_behaviour():
    """tf.math.divide_no_nan docs (a safe divide that returns 0 where the denominator is zero): divide_no_nan(3.0, 0.0) == 0.0 where 3.0 / 0.0 is inf;
    otherwise the plain quotient.  gnn.py:41,44 rely on it for isolated vertices (their degree scale is 0, not inf)."""
    out = orc.divide_no_nan(np.float32(3.0), np.array([0.0, 2.0, -4.0, 0.0], dtype=np.float32))
    assert out.tolist() == [0.0, 1.5, -0.75, 0.0]
    assert np.isfinite(orc.divide_no_nan(np.float32(1.0), np.sqrt(np.zeros(5, dtype=np.float32)))).all()


def test_l2_loss_documented_definition():
    """tf.nn.l2_loss docs: output = sum(t ** 2) / 2 (no square root).  trainable.py:77 (weight decay), layered.py:86."""
    t = np.array([[1.0, -2.0], [3.0, 0.5]])
    assert orc.l2_loss(t) == (1 + 4 + 9 + 0.25) / 2
    assert orc.l2_loss(np.array([3.0, 4.0])) == 12.5                  # not the norm 5, not its square 25


def test_dropout_documented_scaling():
    """tf.nn.dropout docs: with probability rate an element is set to 0, the remaining ones are scaled up by 1 / (1 - rate) so that the
    expected value is preserved -- the documented example turns a tensor of ones at rate 0.5 into 0s and 2s.
    layered.py:50 applies it to the values of the sparse adjacency: kept entries x 1/(1-p), dropped entries explicit zeros, identity
    outside training."""
    idx = np.stack([np.arange(4000) % 50, np.arange(4000) // 50], 1)
    ones = np.ones(4000, dtype=np.float32)
    out = orc.sparse_dropout(idx, ones, 0.5, training=True, seed=1, stream=0)
    assert set(np.unique(out).tolist()) == {0.0, 2.0}
    assert abs(out.mean() - 1.0) < 0.08                               # expected value preserved
    out = orc.sparse_dropout(idx, ones, 0.8, training=True, seed=1, stream=0)
    assert set(np.unique(out).tolist()) == {0.0, np.float32(1.0) / (np.float32(1.0) - np.float32(0.8))}
    assert orc.sparse_dropout(idx, ones, 0.5, training=False) is ones and orc.sparse_dropout(idx, ones, 0, training=True) is ones


def test_sparse_dense_matmul_documented_contract():
    """tf.sparse.sparse_dense_matmul (rank-2 SparseTensor times dense matrix; the op does not validate or reorder the indices, and its
    COO kernel accumulates entry by entry): the product equals to_dense(A) @ B with duplicates summed, in any entry order.
    filter.py:19, gcn.py:88 (graph2adj emits unsorted COO with duplicates)."""
    idx = np.array([[1, 0], [0, 2], [1, 0], [0, 0]])                  # unsorted, (1, 0) twice
    vals = np.array([2.0, 3.0, 5.0, 7.0])
    B = np.arange(6, dtype=np.float64).reshape(3, 2)
    dense = np.zeros((2, 3))
    np.add.at(dense, (idx[:, 0], idx[:, 1]), vals)
    assert dense[1, 0] == 7.0
    np.testing.assert_array_equal(orc.sparse_dense_matmul(idx, vals, (2, 3), B), dense @ B)
    np.testing.assert_array_equal(orc.sparse_dense_matmul(idx[::-1], vals[::-1], (2, 3), B), dense @ B)


def test_sparse_categorical_crossentropy_from_logits_definition():
    """Keras SparseCategoricalCrossentropy(from_logits=True): mean over the batch of -log softmax(logits)[label]; the reference feeds
    it log_softmax(logits) (graph_predictor.py:24-25), and softmax(log_softmax(x)) == softmax(x), so the loss is the plain mean CE.
    Documented worked example (Keras API docs): y_true = [1, 2], y_pred = [[0.05, 0.95, 0], [0.1, 0.8, 0.1]] as PROBABILITIES gives
    1.177; fed as logits of those probabilities' logs the same number must come out."""
    probs = np.array([[0.05, 0.95, 1e-30], [0.1, 0.8, 0.1]])
    loss = orc.node_loss(np.log(probs), [0, 1], [1, 2])
    assert abs(loss - 1.177) < 5e-4
    x = np.random.default_rng(0).standard_normal((6, 5))
    np.testing.assert_allclose(orc.log_softmax(orc.log_softmax(x)), orc.log_softmax(x), atol=1e-12)
