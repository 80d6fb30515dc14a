"""Parameter registry of an architecture: WrappedVariable + VariableGenerator.

Mirrors reference gnntf/core/nn/variables.py:4-66 with torch tensors in place of
tf.Variable: same constructor arguments, same initialisation schemes, same sharing rule.
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


def _uniform(shape, bound, device):
    return (torch.rand(shape, device=device) * 2 - 1) * bound


class WrappedVariable(object):
    """variables.py:4-45.  ``var`` is the raw tensor layers compute with."""

    def __init__(self, shape, normalization='small', trainable=True, regularize=True, name=None):
        self.var = torch.zeros(tuple(shape), dtype=torch.float32, device=default_device(), requires_grad=bool(trainable))
        self.trainable = trainable
        self.regularize = float(regularize)
        self.name = name
        self.normalization = normalization

    def apply_gradient(self, optimizer, gradient):
        if gradient is None:
            return
        self.var.grad = gradient
        optimizer.step()

    def reset(self):
        """Re-draws the value in place (variables.py:17-36)."""
        shape, dev, kind = tuple(self.var.shape), self.var.device, self.normalization
        if isinstance(kind, float):
            value = _uniform(shape, kind, dev)
        elif kind == 'zero':
            value = torch.zeros(shape, device=dev)
        elif kind == 'eye':
            value = torch.eye(shape[1], device=dev)
        elif kind == 'ones':
            value = torch.ones(shape, device=dev)
        elif kind == 'xavier':  # keras GlorotUniform: limit sqrt(6 / (fan_in + fan_out))
            value = _uniform(shape, math.sqrt(6.0 / (shape[0] + shape[1])), dev)
        elif kind == 'he':      # keras HeUniform: limit sqrt(6 / fan_in)
            value = _uniform(shape, math.sqrt(6.0 / shape[0]), dev)
        elif kind == 'bernouli':
            value = (torch.round(torch.rand(shape, device=dev)) * 2 - 1) / shape[1] ** 0.5
        elif kind == 'small':
            value = _uniform(shape, 1. / (shape[1] ** 0.5), dev)
        else:
            raise Exception("Invalid normalization type")
        self.assign(value)

    def identity(self):
        return self.var.detach().clone()

    def numpy(self):
        return self.var.detach().cpu().numpy()

    def assign(self, value):
        with torch.no_grad():
            self.var.copy_(torch.as_tensor(value, dtype=torch.float32, device=self.var.device).reshape(self.var.shape))


class VariableGenerator(object):
    """variables.py:48-66."""

    def __init__(self):
        self.__vars = list()
        self.__named_vars = dict()

    def vars(self):
        return self.__vars

    def create_var(self, *args, shared_name=None, **kwargs):
        if shared_name is not None and shared_name in self.__named_vars:
            return self.__named_vars[shared_name]
        var = WrappedVariable(*args, **kwargs)
        self.__vars.append(var)
        if shared_name is not None:
            self.__named_vars[shared_name] = var.var
        return var.var

    def reset(self):
        for var in self.__vars:
            var.reset()
