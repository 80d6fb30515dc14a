"""Predictor ABC and Trainable: memoised full-graph forward and the full-batch training loop.

Behavioural contract: reference gnntf/core/nn/trainable.py:5-103 -- the ``train()`` signature, the
objective (task loss + layer output penalties + L2 on regularised variables, scaled by
``degradation(epoch)`` only inside the gradient), validation-loss early stopping with a patience
counter that is re-armed on every improvement, and restoring the best variables at the end.
The structure is this build's own: the epoch is three small pieces (``_Objective``, ``_BestSoFar``,
``_fit_epoch``) around torch autograd + torch.optim.Adam (epsilon 1e-7, Keras's default, in place
of tf.GradientTape + keras Adam).  The propagation layers inside the forward run on the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch

from .params import default_device
from .protocol import Layered


class Predictor(object):
    """What a task exposes to an architecture (trainable.py:5-13)."""

    def predict(self, features):
        raise Exception("Predictors need to implement a predict method")

    def loss(self, features):
        raise Exception("Predictors need to implement a loss method")

    def evaluate(self, features):
        raise Exception("Predictors need to implement an evaluate method")


def _as_features(features):
    from .sparse import DeviceGraph, SparseCOO, SparseRows
    if isinstance(features, SparseCOO):                 # a sparse attribute matrix (datasets.load_gnn_benchmark_npz): never densified
        return SparseRows(DeviceGraph(features, device=default_device()))
    if isinstance(features, SparseRows):
        return features
    if isinstance(features, torch.Tensor):
        return features.to(default_device(), torch.float32)
    return torch.as_tensor(np.asarray(features), dtype=torch.float32).to(default_device())


class _Objective:
    """The scalar one optimisation step minimises (trainable.py:71-77): task loss on the training-mode
    forward, plus ``layer.loss()`` of every layer with an output penalty, plus
    ``regularization * var.regularize * sum(var^2)/2`` (= tf.nn.l2_loss) of every regularised variable."""

    def __init__(self, model: "Trainable", task: Predictor, weight_decay: float, replicas: int = 1):
        """``replicas``: how many processes hold a copy of the variables and ADD their gradients up before each step (models over
        vertex blocks, sharded_layers.SummedGradients).  The task loss and the output penalties are sums over rows, of which every
        process holds its own; the weight decay is a property of the variables, which all of them hold -- each contributes 1/replicas
        of it, so that ``regularization`` means the same thing for one process and for P."""
        self.model, self.task = model, task
        self.penalised_layers = [layer for layer in model.layers() if layer.output_regularize != 0]
        self.decayed = [(weight_decay * v.regularize / max(int(replicas), 1), v.var) for v in model.vars() if v.regularize != 0]

    def __call__(self):
        total = self.task.loss(self.model(self.model.features))
        for layer in self.penalised_layers:
            total = total + layer.loss()
        for coeff, tensor in self.decayed:
            total = total + coeff * tensor.square().sum() / 2
        return total


class _BestSoFar:
    """Early-stopping bookkeeping (trainable.py:60-62, 86, 96-102): the lowest validation loss seen, a
    snapshot of every variable taken at that moment, and a countdown re-armed by each improvement."""

    def __init__(self, variables, patience: int):
        self.variables, self.patience = variables, patience
        self.loss = float("inf")
        self.snapshot = [v.identity() for v in variables]
        self.countdown = patience

    def observe(self, loss: float) -> bool:
        """Consumes one epoch of patience; True when ``loss`` is a new minimum (snapshot refreshed)."""
        self.countdown -= 1
        if not loss < self.loss:
            return False
        self.loss = loss
        self.snapshot = [v.identity() for v in self.variables]
        return True

    def rearm(self):
        self.countdown = self.patience

    @property
    def exhausted(self) -> bool:
        return self.countdown == 0

    def restore(self):
        for v, best in zip(self.variables, self.snapshot):
            v.assign(best)


class Trainable(Layered):
    SPARSE_FEATURES_BELOW = 0.05        # input features with a smaller share of non-zeros reach the first Dense as a device CSR

    def __init__(self, features):
        features = _as_features(features)
        super().__init__(tuple(features.shape))
        self.features = features
        self._fast_predict = None
        self._sparse_rows = None        # built on first use (device features only)

    def __call__(self, features):
        if features is self.features:
            features = self._input_features()
        return super().__call__(features)

    def _input_features(self):
        """self.features, or its SparseRows form when it is mostly zeros, lives on the device and the stack starts with
        [Dropout]* Dense (the pre-MLP of filter.py:30-33) -- the only layers that know what to do with it."""
        from .blocks import Dense, Dropout
        from .sparse import SparseRows
        X = self.features
        if isinstance(X, SparseRows):                   # given sparse: only [Dropout]* Dense can take it
            head = [layer for layer in self.layers() if not isinstance(layer, Dropout)]
            if not head or type(head[0]) is not Dense:
                raise Exception("sparse input features need a model whose first layers are [Dropout]* Dense")
            return X
        if self._sparse_rows is None:
            head = [layer for layer in self.layers() if not isinstance(layer, Dropout)]
            eligible = (X.is_cuda and X.dim() == 2 and X.numel() > 0 and head and type(head[0]) is Dense
                        and float(torch.count_nonzero(X)) < self.SPARSE_FEATURES_BELOW * X.numel())
            self._sparse_rows = SparseRows.from_dense(X) if eligible else False
        return self._sparse_rows if self._sparse_rows is not False else X

    def reset(self):
        super().reset()
        self._fast_predict = None

    # ---- memoised full-graph forward (trainable.py:26-39) --------------------------------------------
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

    # ---- training ------------------------------------------------------------------------------------------
    def _make_optimizer(self, optimizer, learning_rate):
        params = [v.var for v in self.vars() if v.trainable]
        if optimizer is None:
            return torch.optim.Adam(params, lr=learning_rate, eps=1e-7)
        if callable(optimizer) and not isinstance(optimizer, torch.optim.Optimizer):
            return optimizer(params)           # a factory taking the parameter list
        return optimizer

    def _fit_epoch(self, objective: _Objective, optimizer, scale, batches: int) -> float:
        """``batches`` full-batch steps in training mode; returns the summed (unscaled) objective."""
        seen = 0.0
        for _ in range(batches):
            with self:                          # training mode on; the exit leaves eval mode (layered.py:37-42)
                optimizer.zero_grad(set_to_none=True)
                value = objective()
                (value * scale).backward()
                optimizer.step()
            seen += float(value.detach())
        return seen

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
              optimizer=None,
              capture: bool = False):
        """trainable.py:41-52.  ``capture=True`` (not in the reference; device models with fixed index lists only) records
        ONE training step and ONE validation forward as hipGraphs and replays them every epoch: on small graphs an epoch is
        ~200 kernel launches of a few microseconds each, and the host loop around them is what takes the time."""
        self.reset()
        if capture:
            return self._train_captured(train, valid, test, patience, learning_rate, regularization, verbose, epochs, degradation,
                                        batches, optimizer)
        optimizer = self._make_optimizer(optimizer, learning_rate)
        judge = train if valid is None else valid
        objective = _Objective(self, train, regularization, replicas=getattr(optimizer, "replicas", 1))
        best = _BestSoFar(self.vars(), patience)
        for epoch in range(epochs):
            self._fast_predict = None
            fitted = self._fit_epoch(objective, optimizer, degradation(epoch), batches)
            with torch.no_grad():               # eval-mode forward: validation decides (trainable.py:83-84)
                logits = self(self.features)
                held_out = float(judge.loss(logits))
            if best.observe(held_out):
                if verbose:
                    self._report(epoch, best.countdown, fitted, held_out, logits, train, judge, test)
                best.rearm()
            if best.exhausted:
                break
        best.restore()
        self._fast_predict = None
        print('\r')

    # ---- the same loop with the device work of an epoch replayed from two hipGraphs -------------------------------------------
    def _dropout_graphs(self):
        """Every DeviceGraph whose kernels draw edge / input-dropout masks for this model."""
        graphs = [self.graph] if hasattr(self, "graph") else []
        rows = self._input_features()
        if rows is not self.features:
            graphs.append(rows.graph)
        return graphs

    def _train_captured(self, train, valid, test, patience, learning_rate, regularization, verbose, epochs, degradation, batches,
                        optimizer):
        if not self.features.is_cuda:
            raise Exception("train(capture=True) needs the model on the GPU")
        if optimizer is not None and isinstance(optimizer, torch.optim.Optimizer):
            raise Exception("train(capture=True) builds its own capturable optimizer: pass a factory or nothing")
        device = self.features.device
        params = [v.var for v in self.vars() if v.trainable]
        make_optimizer = (lambda: optimizer(params)) if optimizer is not None else \
            (lambda: torch.optim.Adam(params, lr=learning_rate, eps=1e-7, capturable=True))
        judge = train if valid is None else valid
        objective = _Objective(self, train, regularization)
        scale = torch.ones((), dtype=torch.float32, device=device)           # degradation(epoch), refreshed before every replay
        counter = torch.zeros(1, dtype=torch.int64, device=device)           # added to every dropout stream id on the device
        graphs = self._dropout_graphs()
        for g in graphs:
            g.set_dropout_counter(counter)

        def train_step(opt):
            with self:
                opt.zero_grad(set_to_none=True)
                value = objective()
                (value * scale).backward()
                opt.step()
            return value

        def validate():
            with torch.no_grad():
                logits = self(self.features)
                return logits, judge.loss(logits)

        try:
            # warm-up on a side stream (lazy allocations of the library, index uploads, autograd workspaces), then undo it:
            # parameters, the mask numbering and the optimizer state start the captured run exactly where an eager run starts
            start = [v.identity() for v in self.vars()]
            first_mask = self._mask_calls
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            optimizer = make_optimizer()
            with torch.cuda.stream(side):
                for _ in range(2):                                          # also creates the optimizer's state tensors: their
                    train_step(optimizer)                                   # initialisation must not end up inside the graph
                    validate()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            for v, value in zip(self.vars(), start):
                v.assign(value)
            masks_per_step = (self._mask_calls - first_mask) // 2
            self._mask_calls = first_mask
            for state in optimizer.state.values():                          # back to a fresh optimizer: moments and step counts zero
                for item in state.values():
                    if isinstance(item, torch.Tensor):
                        item.zero_()
            optimizer.zero_grad(set_to_none=True)
            step_graph, eval_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(step_graph):
                fitted_value = train_step(optimizer)
                counter.add_(masks_per_step)
            self._mask_calls = first_mask                                    # the device counter does the numbering from here on
            with torch.cuda.graph(eval_graph):
                logits, held_out_value = validate()
        except Exception as error:
            for g in graphs:
                g.set_dropout_counter(None)
            raise Exception("train(capture=True): this model / task cannot be recorded as a device graph (" + str(error) + ")")
        counter.zero_()
        best = _BestSoFar(self.vars(), patience)
        steps = 0
        for epoch in range(epochs):
            self._fast_predict = None
            scale.fill_(float(degradation(epoch)))
            for _ in range(batches):
                step_graph.replay()
                steps += 1
            eval_graph.replay()
            held_out = float(held_out_value)                                # the one synchronisation of the epoch
            if best.observe(held_out):
                if verbose:
                    self._report(epoch, best.countdown, float(fitted_value) * batches, held_out, logits, train, judge, test)
                best.rearm()
            if best.exhausted:
                break
        best.restore()
        for g in graphs:
            g.set_dropout_counter(None)
        self._mask_calls = first_mask + steps * masks_per_step
        self._training = False
        self._fast_predict = None
        print('\r')

    @staticmethod
    def _report(epoch, countdown, fitted, held_out, logits, train, judge, test):
        scores = dict(train=float(train.evaluate(logits)), valid=float(judge.evaluate(logits)),
                      test=float("nan") if test is None else float(test.evaluate(logits)))
        print("\r[epoch %d | patience %d] loss: train %.3f, valid %.3f | acc: train %.3f, valid %.3f, test %.3f"
              % (epoch, countdown, fitted, held_out, scores["train"], scores["valid"], scores["test"]), end='')
