"""NodeClassification, the consumer of the propagated logits.
Mirrors reference gnntf/core/gnn/graph_predictor.py:10-31 (the link-prediction tasks in
that file belong to a different path)."""
from __future__ import annotations

import numpy as np
import torch

from . import sparse
from .training import Predictor


def _index(nodes, device):
    return torch.as_tensor(np.asarray(nodes), dtype=torch.int64, device=device)


class NodeClassification(Predictor):
    def __init__(self, nodes, labels=None, loss_transform=None):
        self.nodes = nodes
        self.labels = labels
        self.loss_transform = loss_transform
        self._on_device = dict()       # (what, device, size) -> checked, uploaded index list: the same lists serve every epoch

    def _device_index(self, what, values, features, upper):
        key = (what, features.device, upper)
        if key not in self._on_device:
            self._on_device[key] = sparse.DeviceIndex(values, features.device, upper, "node id" if what == "nodes" else "label")
        return self._on_device[key]

    def predict(self, features):
        """argmax over the rows of ``nodes`` (graph_predictor.py:16-17); on the device one fused gather+argmax launch."""
        if features.is_cuda:
            return sparse.node_argmax(features, self._device_index("nodes", self.nodes, features, features.shape[0]))
        return torch.argmax(features[_index(self.nodes, features.device)], dim=1)

    def loss(self, features):
        if self.labels is None:
            raise Exception("Evaluation requires node labels")
        if self.loss_transform is not None:
            features = self.loss_transform(features)
        if features.is_cuda:          # gather + log-softmax + cross entropy fused (gnx_node_ce)
            return sparse.node_ce(features, self._device_index("nodes", self.nodes, features, features.shape[0]),
                                  self._device_index("labels", self.labels, features, features.shape[1]))
        predictions = torch.log_softmax(features[_index(self.nodes, features.device)], dim=1)
        # SparseCategoricalCrossentropy(from_logits=True) on top of the log-softmax (graph_predictor.py:24-25)
        return torch.nn.functional.cross_entropy(predictions, _index(self.labels, features.device))

    def evaluate(self, features):
        if self.labels is None:
            raise Exception("Evaluation requires node labels")
        predictions = self.predict(features)
        wrong = torch.count_nonzero(predictions - _index(self.labels, features.device)).item()
        return 1 - wrong / predictions.shape[0]
