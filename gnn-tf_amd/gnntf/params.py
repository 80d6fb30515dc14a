"""Parameter registry of an architecture: WrappedVariable + VariableGenerator.

Behavioural contract: reference gnntf/core/nn/variables.py:4-66 -- the constructor arguments of a variable, the initialisation
schemes by name ('small' = U(+-1/sqrt(fan_out)) is the default, a float = U(+-that), 'zero', 'eye', 'ones', 'xavier', 'he',
'bernouli'), ``create_var(..., shared_name=)`` handing out ONE tensor per shared name, ``vars()`` listing every variable in
creation order and ``reset()`` re-drawing all of them.  The structure is this build's own: torch tensors, initialisers as a table.
"""
from __future__ import annotations

import math

import torch

_default_device = None


def set_default_device(device):
    """Device new variables / features are placed on (default: cuda if present, else cpu)."""
    global _default_device
    _default_device = torch.device(device) if device is not None else None


def default_device() -> torch.device:
    if _default_device is not None:
        return _default_device
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _symmetric_uniform(shape, bound, device):
    return (torch.rand(shape, device=device) * 2 - 1) * bound


# name -> f(shape, device): the value a reset() draws (variables.py:17-36; xavier / he are Keras's GlorotUniform / HeUniform limits)
_INITIALISERS = {
    "zero": lambda shape, dev: torch.zeros(shape, device=dev),
    "ones": lambda shape, dev: torch.ones(shape, device=dev),
    "eye": lambda shape, dev: torch.eye(shape[1], device=dev),
    "small": lambda shape, dev: _symmetric_uniform(shape, shape[1] ** -0.5, dev),
    "xavier": lambda shape, dev: _symmetric_uniform(shape, math.sqrt(6.0 / (shape[0] + shape[1])), dev),
    "he": lambda shape, dev: _symmetric_uniform(shape, math.sqrt(6.0 / shape[0]), dev),
    "bernouli": lambda shape, dev: (torch.round(torch.rand(shape, device=dev)) * 2 - 1) * shape[1] ** -0.5,
}


class WrappedVariable(object):
    """One parameter: ``var`` is the raw tensor the layers compute with; ``regularize`` its weight-decay factor (a float, so that
    ``regularize=False`` switches the decay off), ``normalization`` the initialiser it is reset with."""

    def __init__(self, shape, normalization='small', trainable=True, regularize=True, name=None):
        self.name, self.normalization = name, normalization
        self.trainable, self.regularize = trainable, float(regularize)
        self.var = torch.zeros(tuple(shape), dtype=torch.float32, device=default_device(), requires_grad=bool(trainable))

    def reset(self):
        """Re-draws the value in place."""
        scheme, shape, dev = self.normalization, tuple(self.var.shape), self.var.device
        if isinstance(scheme, float):
            fresh = _symmetric_uniform(shape, scheme, dev)
        elif scheme in _INITIALISERS:
            fresh = _INITIALISERS[scheme](shape, dev)
        else:
            raise Exception("Invalid normalization type")
        self.assign(fresh)

    def assign(self, value):
        with torch.no_grad():
            self.var.copy_(torch.as_tensor(value, dtype=torch.float32, device=self.var.device).reshape(self.var.shape))

    def identity(self):
        """A detached copy (what the early-stopping snapshot keeps)."""
        return self.var.detach().clone()

    def numpy(self):
        return self.identity().cpu().numpy()

    def apply_gradient(self, optimizer, gradient):
        if gradient is not None:
            self.var.grad = gradient
            optimizer.step()


class VariableGenerator(object):
    """Hands out parameters and remembers them: everything an architecture trains is created through ``create_var``."""

    def __init__(self):
        self._created = []          # every WrappedVariable, in creation order
        self._by_shared_name = {}   # shared_name -> the raw tensor handed out under it

    def create_var(self, *args, shared_name=None, **kwargs):
        """The raw tensor of a new variable -- or, for a ``shared_name`` seen before, the tensor created under that name."""
        known = self._by_shared_name.get(shared_name) if shared_name is not None else None
        if known is not None:
            return known
        wrapped = WrappedVariable(*args, **kwargs)
        self._created.append(wrapped)
        if shared_name is not None:
            self._by_shared_name[shared_name] = wrapped.var
        return wrapped.var

    def vars(self):
        return self._created

    def reset(self):
        for wrapped in self._created:
            wrapped.reset()
