"""Metrics of reference gnntf/measures.py:7-45: set_seed, acc, and the ranking metrics the link-prediction tasks report
(auc; avprec / rec / prec / f1 at k).  auc is computed from the rank statistic (ties get their mean rank), which equals the
area under sklearn's ROC curve that the reference integrates."""
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


def auc(labels, predictions):
    """measures.py:17-19: area under the ROC curve = P(score of a positive > score of a negative) + half the ties."""
    labels, predictions = _np(labels).reshape(-1), _np(predictions).reshape(-1).astype(np.float64)
    pos = labels == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(predictions, kind="mergesort")
    ranks = np.empty(len(predictions), dtype=np.float64)
    sorted_scores = predictions[order]
    start = 0
    while start < len(order):                               # mean rank over each run of equal scores
        stop = start
        while stop + 1 < len(order) and sorted_scores[stop + 1] == sorted_scores[start]:
            stop += 1
        ranks[order[start:stop + 1]] = (start + stop) / 2.0 + 1.0
        start = stop + 1
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def _top(predictions, k):
    return np.argsort(_np(predictions).reshape(-1))[-k:]


def avprec(labels, predictions, k=5):
    """measures.py:22-27: sum over the top-k (best first) of label / position, over the number of positives in the top-k."""
    labels, top = _np(labels).reshape(-1), _top(predictions, k)
    gain = sum(labels[i] / (position + 1) for position, i in enumerate(reversed(top)))
    return 0 if gain == 0 else gain / np.sum(labels[top])


def rec(labels, predictions, k=5):
    """measures.py:30-32."""
    labels = _np(labels).reshape(-1)
    return np.sum(labels[_top(predictions, k)]) / np.sum(labels)


def prec(labels, predictions, k=5):
    """measures.py:35-37."""
    return np.mean(_np(labels).reshape(-1)[_top(predictions, k)])


def f1(labels, predictions, k=5):
    """measures.py:40-45."""
    precision, recall = prec(labels, predictions, k), rec(labels, predictions, k)
    return 0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)
