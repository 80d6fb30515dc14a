"""set_seed / acc of reference gnntf/measures.py:7-14 (the link-prediction metrics there
are outside the propagation path)."""
from __future__ import annotations

import random

import numpy as np
import torch

_seed = 0


def set_seed(seed):
    """measures.py:7-10: seeds python, numpy and the tensor library; also the counter RNG
    that draws edge-dropout masks on the device."""
    global _seed
    _seed = int(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def current_seed() -> int:
    return _seed


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def acc(predictions, labels):
    """measures.py:13-14."""
    predictions, labels = _np(predictions), _np(labels)
    return 1 - np.count_nonzero(predictions - labels) / predictions.shape[0]
