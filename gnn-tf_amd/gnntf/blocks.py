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


class Activation(Layer):
    """layers.py:139-172."""

    def __build__(self, architecture: Layered, activation: str = "relu", **kwargs):
        if activation == "relu":
            fn = torch.relu
        elif activation == "linear":
            fn = linear
        elif activation == "tanh":
            fn = torch.tanh
        elif activation == "exp":
            fn = torch.exp
        elif activation == "softmax":
            fn = lambda x: torch.softmax(x, dim=1)
        elif activation == "scale":
            scale = architecture.create_var((1, 1), "zero", regularize=False)
            fn = lambda x: x * (1 + scale)
        elif activation == "kernel":
            s = [architecture.create_var((1, 1), "ones" if i == 0 else "zero", regularize=False) for i in range(6)]
            fn = lambda x: torch.log(torch.exp(x * s[0] + s[3]) + torch.exp(x * s[1] + s[4]) + torch.exp(x * s[2] + s[5]))
        elif activation == "softthresh":
            theta = kwargs['threshold'] if 'threshold' in kwargs else architecture.create_var((1, 1), "zero", regularize=False)
            fn = lambda x: torch.relu(x - theta) - torch.relu(theta - x)
        else:
            fn = activation
        self.activation = fn
        return architecture.top_shape()

    def __forward__(self, gcn, features):
        return self.activation(features)


class Branch(Layer):
    """layers.py:68-74: restarts the flow from a given feature matrix."""

    def __build__(self, architecture: Layered, features):
        self.features = features
        return tuple(self.features.shape)

    def __forward__(self, architecture: Layered, features):
        return self.features


class Resume(Layer):
    """layers.py:77-83: continues from another layer's cached value."""

    def __build__(self, architecture: Layered, H0: Layer):
        self.H0 = H0
        return H0.output_shape

    def __forward__(self, architecture: Layered, features):
        return self.H0.value


class Concatenate(Layer):
    """layers.py:86-101 (including its axis-0 concatenation)."""

    def __build__(self, architecture: Layered, H0):
        self.H0 = H0
        first = H0[0] if isinstance(H0, list) else H0
        for H in (H0 if isinstance(H0, list) else [H0]):
            if architecture.top_shape()[0] != H.output_shape[0]:
                raise Exception("Mismatching first dimension to concatenate between shapes " + str(architecture.top_shape())
                                + " and " + str(H.output_shape))
        return (architecture.top_shape()[0], architecture.top_shape()[1] + first.output_shape[1])

    def __forward__(self, architecture: Layered, features):
        if isinstance(self.H0, list):
            return torch.cat([H.value for H in self.H0], dim=0)
        return torch.cat([features, self.H0.value], dim=0)


class Tradeoff(Layer):
    """layers.py:104-122: sigmoid-weighted average of other layers' values."""

    def __build__(self, architecture: Layered, layers, weights=None, trainable=True):
        shape = layers[0].output_shape
        for layer in layers:
            if layer.output_shape != shape:
                raise Exception("Mismatching trade-off dimentions")
        self.layers = layers
        self.weights = [architecture.create_var((1, 1), "zero", trainable=trainable) for _ in layers] if weights is None else weights
        return shape

    def __forward__(self, architecture: Layered, features):
        gates = [torch.sigmoid(torch.as_tensor(w)) for w in self.weights]
        total = sum(gates)
        return sum(g * layer.value / total for g, layer in zip(gates, self.layers))
