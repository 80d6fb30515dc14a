"""The Layer protocol and the Layered container -- the drop-in boundary of the hot path.

API mirror of reference gnntf/core/nn/layered.py:5-86 (same public names, arguments, messages and
quirks), written for torch tensors:

  * a Layer is constructed with its arguments deferred; ``Layered.add`` binds it to the architecture
    by calling ``__build__(architecture, *args, **kwargs)``, which must return the output shape
    (layered.py:59-71); ``layer(architecture, x)`` runs ``__forward__`` and caches the result in
    ``layer.value`` (layered.py:79-81), which other layers (PPRIteration's H0) read;
  * the container keeps a training-mode flag that starts **True** and is only cleared when a
    ``with architecture as variables:`` block exits (layered.py:9,37-42) -- so a forward before any
    training runs with dropout on, exactly like the reference;
  * feature dropout and edge ("sparse") dropout are gated on that flag (layered.py:44-50).
"""
from __future__ import annotations

import torch

from .params import VariableGenerator


class Layer(object):
    """Base class of every layer.  Subclasses implement ``__build__`` and ``__forward__``."""

    def __init__(self, *args, output_regularize: float = 0, **kwargs):
        self._pending = (args, kwargs)          # consumed by __late_init__
        self.output_regularize = output_regularize

    # -- to be provided by subclasses ------------------------------------------------------------
    def __build__(self, architecture: VariableGenerator, *args, **kwargs):
        raise Exception("Layer should implment a __build__ method")

    def __forward__(self, architecture: VariableGenerator, features):
        raise Exception("Layer should implement a __forward__ method")

    # -- protocol ------------------------------------------------------------------------------------
    def __late_init__(self, architecture: VariableGenerator):
        """Binds the layer: runs __build__ with the deferred arguments, records the output shape and
        which variables the layer created (layered.py:64-71)."""
        args, kwargs = self._pending
        known = set(architecture.vars())
        shape = self.__build__(architecture, *args, **kwargs)
        if shape is None:
            raise Exception("Layer __build__ should return an output shape")
        self.output_shape = shape
        self.vars = set(architecture.vars()) - known
        self._pending = None

    def __call__(self, architecture: VariableGenerator, features):
        self.value = self.__forward__(architecture, features)
        return self.value

    def __run__(self, architecture: VariableGenerator, features, stack, at):
        """Optional: execute this layer TOGETHER with layers that follow it in ``stack`` (it sits at index ``at``) and return
        (number of layers executed, output), having set their ``.value``; None (the default) = run layer by layer.  Not in the
        reference: the container's loop (layered.py:52-55) is the same either way, a run only saves launches and intermediates."""
        return None

    def loss(self):
        """output_regularize * tf.nn.l2_loss(value) = output_regularize * sum(value^2)/2 (layered.py:83-86)."""
        if self.output_regularize == 0:
            return 0
        return self.output_regularize * (self.value ** 2).sum() / 2


class Layered(VariableGenerator):
    """Sequential container of layers + variable registry + training-mode switch."""

    def __init__(self, input_shape, layers=list()):
        super().__init__()
        self.input_shape = tuple(input_shape)
        self._stack = []
        self._training = True                    # layered.py:9
        self._mask_calls = 0                     # edge-dropout mask streams handed out so far
        for layer in layers:
            self.add(layer)

    # -- structure -------------------------------------------------------------------------------------
    def add(self, layer):
        if not any(layer is known for known in self._stack):
            layer.__late_init__(self)
        self._stack.append(layer)
        return layer

    def layers(self):
        return self._stack

    def top_layer(self):
        return self._stack[-1]

    def top_shape(self):
        return self._stack[-1].output_shape if self._stack else self.input_shape

    fuse_runs = True                             # let layers execute runs of their kind in one go (Layer.__run__); False: strictly layer by layer

    def __call__(self, features):
        return self.run(features)

    def run(self, features, first=0):
        """The container's loop (layered.py:52-55) from layer index ``first`` on: ``run(x)`` is ``__call__(x)``; ``run(h, first=i)``
        continues from a value of layer ``i - 1`` the caller already holds (not in the reference: benchmarks time the propagation
        layers of a model through it without re-running the pre-MLP)."""
        stack, at = self._stack, first
        while at < len(stack):                   # layered.py:52-55, except that a layer may take the ones that follow it along
            done = stack[at].__run__(self, features, stack, at) if self.fuse_runs else None
            if done is None:
                features = stack[at](self, features)
                at += 1
            else:
                at, features = at + done[0], done[1]
        return features

    # -- training mode -----------------------------------------------------------------------------------
    def is_training(self):
        return self._training

    def training_mode(self, training_mode):
        self._training = training_mode

    def __enter__(self):
        self._training = True
        return [v.var for v in self.vars() if v.trainable]

    def __exit__(self, type, value, tb):
        self._training = False

    # -- dropout -------------------------------------------------------------------------------------------
    def dropout(self, features, dropout=0.5):
        """tf.nn.dropout(features, rate) while training, identity otherwise (layered.py:44-45)."""
        if not self._training or dropout == 0:
            return features
        from .sparse import SparseRows
        if isinstance(features, SparseRows):            # mostly-zero input features: drop stored entries (zeros stay zero either way)
            return features.with_dropout(float(dropout), *self._next_mask_stream())
        return torch.nn.functional.dropout(features, p=float(dropout), training=True)

    def sparse_dropout(self, G, dropout=0.5):
        """Edge dropout (layered.py:47-50).  ``G`` is an Adjacency; while training a NEW Adjacency is
        returned whose raw values were dropped per stored COO entry on the device, otherwise G itself."""
        if dropout == 0 or not self._training:
            return G
        from .sparse import normalize
        seed, stream = self._next_mask_stream()
        return normalize(G.graph, "none", "none", dropout, seed, stream)

    def _next_mask_stream(self, n=1):
        """(seed, first stream id) of the next ``n`` edge-dropout masks of the counter RNG."""
        from . import metrics
        first = self._mask_calls
        self._mask_calls = first + n
        return metrics.current_seed(), first
