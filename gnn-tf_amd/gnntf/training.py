"""Predictor ABC and Trainable: cached full-graph forward, predict/loss/evaluate, and the
full-batch training loop with L2 regularisation, early stopping and best-weights restore.

Mirrors reference gnntf/core/nn/trainable.py:5-103 (same signatures and bookkeeping);
tf.GradientTape + keras Adam become torch autograd + torch.optim.Adam (epsilon 1e-7 as in
Keras).  The propagation layers inside the forward run on the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch

from .params import default_device
from .protocol import Layered


class Predictor(object):
    def predict(self, features):
        raise Exception("Predictors need to implement a predict method")

    def loss(self, features):
        raise Exception("Predictors need to implement a loss method")

    def evaluate(self, features):
        raise Exception("Predictors need to implement an evaluate method")


def _as_features(features):
    if isinstance(features, torch.Tensor):
        return features.to(default_device(), torch.float32)
    return torch.as_tensor(np.asarray(features), dtype=torch.float32).to(default_device())


class Trainable(Layered):
    def __init__(self, features):
        features = _as_features(features)
        super().__init__(tuple(features.shape))
        self.features = features
        self._fast_predict = None

    def reset(self):
        super().reset()
        self._fast_predict = None

    def _cached_forward(self):
        if self._fast_predict is None:
            with torch.no_grad():
                self._fast_predict = self(self.features)
        return self._fast_predict

    def predict(self, predictor: Predictor):
        return predictor.predict(self._cached_forward())

    def loss(self, predictor: Predictor):
        return predictor.loss(self._cached_forward())

    def evaluate(self, predictor: Predictor):
        return predictor.evaluate(self._cached_forward())

    def train(self,
              train: Predictor,
              valid: Predictor = None,
              test: Predictor = None,
              patience: int = 100,
              learning_rate: float = 0.01,
              regularization: float = 5.E-4,
              verbose: bool = False,
              epochs: int = 2000,
              degradation=lambda epoch: 1,
              batches: int = 1,
              optimizer=None):
        self.reset()
        params = [var.var for var in self.vars() if var.trainable]
        if optimizer is None:
            optimizer = torch.optim.Adam(params, lr=learning_rate, eps=1e-7)
        elif callable(optimizer) and not isinstance(optimizer, torch.optim.Optimizer):
            optimizer = optimizer(params)
        if valid is None:
            valid = train
        min_loss = float('inf')
        min_loss_vars = [var.identity() for var in self.vars()]
        patience_remaining = patience
        for epoch in range(epochs):
            self._fast_predict = None
            loss = 0
            for _ in range(batches):
                with self as vars:
                    optimizer.zero_grad(set_to_none=True)
                    batch_loss = train.loss(self(self.features))
                    for layer in self.layers():
                        if layer.output_regularize != 0:
                            batch_loss = batch_loss + layer.loss()
                    for var in self.vars():
                        if var.regularize != 0:
                            batch_loss = batch_loss + regularization * var.regularize * (var.var ** 2).sum() / 2
                    (batch_loss * degradation(epoch)).backward()
                    optimizer.step()
                    loss = loss + float(batch_loss.detach())

            # patience mechanism (trainable.py:82-100); the exit of the `with` above left eval mode on
            with torch.no_grad():
                output = self(self.features)
                valid_loss = float(valid.loss(output))
            patience_remaining -= 1
            if verbose and valid_loss < min_loss:
                train_acc = float(train.evaluate(output))
                test_acc = float("nan") if test is None else float(test.evaluate(output))
                valid_acc = float(valid.evaluate(output))
                print(f'\rEpoch {epoch}  patience {patience_remaining}  Train loss {float(loss):.3f} Validation loss {valid_loss:.3f}  Train {train_acc:.3f} Validation {valid_acc:.3f}  Test {test_acc:.3f}', end='')
            if valid_loss < min_loss:
                min_loss, min_loss_vars = valid_loss, [var.identity() for var in self.vars()]
                patience_remaining = patience
            if patience_remaining == 0:
                break
        for var, best_var in zip(self.vars(), min_loss_vars):
            var.assign(best_var)
        self._fast_predict = None
        print('\r')
