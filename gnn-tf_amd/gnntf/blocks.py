"""Generic layers the propagation path plugs into: Dense, Dropout, Activation and the flow
layers.  Mirrors reference gnntf/core/nn/layers.py:68-181 (the Keras adapter ``Wrap`` and
the toy ``LSTM`` there are TensorFlow-specific / unused and are not part of this path).
On device tensors the dense transform runs on this repo's matrix-core kernel (gnx_dense, float32 MFMA); mostly-zero
input features reach the first Dense as a device CSR and are transformed by the SpMM kernel instead (sparse.SparseRows).
CPU tensors (host-logic tests of the protocol) use plain torch ops.
"""
from __future__ import annotations

import torch

from . import sparse
from .protocol import Layer, Layered


def linear(x):
    return x


relu = torch.relu


def is_relu(fn) -> bool:
    """Whether an activation argument is the relu (under any of its torch names): what the kernels can run in their epilogue."""
    return fn is torch.relu or fn is torch.nn.functional.relu


def affine(features, W, b, activation=linear):
    """activation(features . W + b) -- layers.py:136.  Device tensors: gnx_dense (MFMA) with the bias and a relu fused;
    SparseRows: the SpMM kernel over the rows of W; CPU tensors: torch."""
    bias = b if isinstance(b, torch.Tensor) else None
    fused_act = activation is relu or activation is linear
    if isinstance(features, sparse.SparseRows):
        out = sparse.sparse_dense(features, W, bias, relu=activation is relu)
    elif features.is_cuda:
        out = sparse.dense(features, W, bias, relu=activation is relu)
    else:
        return activation(torch.matmul(features, W) + b)
    return out if fused_act else activation(out)


class Dense(Layer):
    """layers.py:125-136: dropout(activation(X.W + b))."""

    def __build__(self, architecture: Layered, outputs: int = None, activation=linear, bias: bool = True,
                  dropout: float = 0, regularize: bool = True):
        if outputs is None:
            outputs = architecture.top_shape()[1]
        self.W = architecture.create_var((architecture.top_shape()[1], outputs), regularize=regularize)
        self.b = architecture.create_var((1, outputs), "zero", regularize=regularize) if bias else 0
        self.activation = activation
        self.dropout = dropout
        return (architecture.top_shape()[0], outputs)

    def __forward__(self, architecture: Layered, features):
        return architecture.dropout(affine(features, self.W, self.b, self.activation), self.dropout)


class Dropout(Layer):
    """layers.py:175-181."""

    def __build__(self, gcn, rate: float = 0.5):
        self.rate = rate
        return gcn.top_shape()

    def __forward__(self, gcn, features):
        return gcn.dropout(features, self.rate)


def _scalar(architecture, init="zero"):
    """A trainable [1, 1] parameter outside the weight decay (the parametrised activations' knobs)."""
    return architecture.create_var((1, 1), init, regularize=False)


def _scaled(architecture, **kwargs):
    gain = _scalar(architecture)
    return lambda x: x * (1 + gain)


def _log_sum_of_three(architecture, **kwargs):
    slopes = [_scalar(architecture, "ones" if i == 0 else "zero") for i in range(3)]
    offsets = [_scalar(architecture) for _ in range(3)]
    return lambda x: torch.logsumexp(torch.stack([x * k + c for k, c in zip(slopes, offsets)]), dim=0)


def _soft_threshold(architecture, threshold=None, **kwargs):
    theta = _scalar(architecture) if threshold is None else threshold
    return lambda x: torch.relu(x - theta) - torch.relu(theta - x)


# name -> factory(architecture, **kwargs) of the elementwise function (layers.py:139-172's names and parameter counts)
ACTIVATIONS = {
    "relu": lambda architecture, **kw: torch.relu,
    "linear": lambda architecture, **kw: linear,
    "tanh": lambda architecture, **kw: torch.tanh,
    "exp": lambda architecture, **kw: torch.exp,
    "softmax": lambda architecture, **kw: (lambda x: torch.softmax(x, dim=1)),
    "scale": _scaled,
    "kernel": _log_sum_of_three,
    "softthresh": _soft_threshold,
}


class Activation(Layer):
    """layers.py:139-172: a named elementwise function (some with trainable scalars), or any callable given in its place."""

    def __build__(self, architecture: Layered, activation: str = "relu", **kwargs):
        make = ACTIVATIONS.get(activation) if isinstance(activation, str) else None
        self.activation = make(architecture, **kwargs) if make is not None else activation
        return architecture.top_shape()

    def __forward__(self, gcn, features):
        return self.activation(features)


# ---- flow layers (layers.py:68-122): they re-route values between layers and own no kernel of this path -------------------
def _sources(H0):
    """The layers a flow layer reads from: one layer or a list of them."""
    return list(H0) if isinstance(H0, (list, tuple)) else [H0]


class _Tap(Layer):
    """A layer whose output is something that already exists: ``pick()`` says what; the incoming features are ignored."""

    def __forward__(self, architecture: Layered, features):
        return self.pick()


class Branch(_Tap):
    """layers.py:68-74: restarts the flow from a given feature matrix."""

    def __build__(self, architecture: Layered, features):
        self.features = features
        self.pick = lambda: self.features
        return tuple(features.shape)


class Resume(_Tap):
    """layers.py:77-83: continues from another layer's cached value."""

    def __build__(self, architecture: Layered, H0: Layer):
        self.H0 = H0
        self.pick = lambda: self.H0.value
        return H0.output_shape


class Concatenate(Layer):
    """layers.py:86-101: joins the incoming features with another layer's value, or a LIST of layers' values with each other --
    along axis 0, although the declared output shape widens axis 1 (the reference's behaviour; NGCF's link
    tasks index the rows of the first block)."""

    def __build__(self, architecture: Layered, H0):
        self.H0 = H0
        rows, width = architecture.top_shape()
        odd = [H for H in _sources(H0) if H.output_shape[0] != rows]
        if odd:
            raise Exception("Mismatching first dimension to concatenate between shapes " + str(architecture.top_shape())
                            + " and " + str(odd[0].output_shape))
        return (rows, width + _sources(H0)[0].output_shape[1])

    def __forward__(self, architecture: Layered, features):
        parts = [H.value for H in _sources(self.H0)]
        return torch.cat(parts if isinstance(self.H0, list) else [features] + parts, dim=0)


class Tradeoff(Layer):
    """layers.py:104-122: average of other layers' values with sigmoid gates, normalised to sum to one."""

    def __build__(self, architecture: Layered, layers, weights=None, trainable=True):
        if len({tuple(layer.output_shape) for layer in layers}) > 1:
            raise Exception("Mismatching trade-off dimentions")
        self.layers = layers
        self.weights = weights if weights is not None else [architecture.create_var((1, 1), "zero", trainable=trainable) for _ in layers]
        return layers[0].output_shape

    def __forward__(self, architecture: Layered, features):
        values = [layer.value for layer in self.layers]
        gates = torch.sigmoid(torch.stack([torch.as_tensor(w, dtype=values[0].dtype, device=values[0].device).reshape(()) for w in self.weights]))
        share = gates / gates.sum()
        return sum(values[i] * share[i] for i in range(len(values)))
