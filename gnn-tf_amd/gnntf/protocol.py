"""The Layer protocol and the Layered container -- the drop-in boundary of the hot path.

Mirrors reference gnntf/core/nn/layered.py:5-86: deferred layer construction
(__late_init__ -> __build__ returns the output shape), cached ``.value`` per layer, a
training-mode flag that starts True and is cleared on leaving ``with architecture:``,
feature dropout and edge ("sparse") dropout gated on that flag.
"""
from __future__ import annotations

import torch

from .params import VariableGenerator


class Layered(VariableGenerator):
    def __init__(self, input_shape, layers=list()):
        super().__init__()
        self.__layers = list()
        self.__training_mode = True      # layered.py:9 -- True until the first `with` block exits
        self.input_shape = tuple(input_shape)
        for layer in layers:
            self.add(layer)

    def layers(self):
        return self.__layers

    def top_shape(self):
        return self.__layers[-1].output_shape if self.__layers else self.input_shape

    def top_layer(self):
        return self.__layers[-1]

    def add(self, layer):
        if layer not in self.__layers:
            layer.__late_init__(self)
        self.__layers.append(layer)
        return layer

    def is_training(self):
        return self.__training_mode

    def training_mode(self, training_mode):
        self.__training_mode = training_mode

    def __enter__(self):
        self.__training_mode = True
        return [var.var for var in self.vars() if var.trainable]

    def __exit__(self, type, value, tb):
        self.__training_mode = False

    def dropout(self, features, dropout=0.5):
        """tf.nn.dropout(features, rate) in training mode, identity otherwise (layered.py:44-45)."""
        if self.__training_mode and dropout != 0:
            return torch.nn.functional.dropout(features, p=float(dropout), training=True)
        return features

    def sparse_dropout(self, G, dropout=0.5):
        """Edge dropout (layered.py:47-50).  ``G`` is an Adjacency over raw values; in
        training mode a new Adjacency is returned whose values were dropped per stored COO
        entry on the device; otherwise G itself."""
        if dropout == 0 or not self.__training_mode:
            return G
        from .sparse import normalize
        seed, stream = self._next_mask_stream()
        return normalize(G.graph, "none", "none", dropout, seed, stream)

    # counter RNG bookkeeping for edge-dropout masks (one fresh stream id per call)
    def _next_mask_stream(self, n=1):
        from . import metrics
        count = getattr(self, "_mask_calls", 0)
        self._mask_calls = count + n
        return metrics.current_seed(), count

    def __call__(self, features):
        for layer in self.__layers:
            features = layer(self, features)
        return features


class Layer(object):
    def __init__(self, *args, output_regularize: float = 0, **kwargs):
        self.__args = args
        self.__kwargs = kwargs
        self.output_regularize = output_regularize

    def __late_init__(self, architecture: VariableGenerator):
        before = set(architecture.vars())
        self.output_shape = self.__build__(architecture, *self.__args, **self.__kwargs)
        if self.output_shape is None:
            raise Exception("Layer __build__ should return an output shape")
        self.vars = set(architecture.vars()) - before
        self.__args = None
        self.__kwargs = None

    def __build__(self, architecture: VariableGenerator, *args, **kwargs):
        raise Exception("Layer should implment a __build__ method")

    def __forward__(self, architecture: VariableGenerator, features):
        raise Exception("Layer should implement a __forward__ method")

    def __call__(self, architecture: VariableGenerator, features):
        self.value = self.__forward__(architecture, features)
        return self.value

    def loss(self):
        if self.output_regularize == 0:
            return 0
        return self.output_regularize * (self.value ** 2).sum() / 2   # tf.nn.l2_loss
