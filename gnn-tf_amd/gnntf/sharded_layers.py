"""The layer API over ONE vertex block of a graph (gnntf/sharded.py holds the block itself): every rank builds the same layer
stack over the rows of its block; the layers here stand where the single-GPU model has PPRIteration / PPRLoop (filter.py:17-35),
GCNLayer (gcn.py:77-89) and GCNIILayer (gcn.py:7-27), and exchange halo rows with the other ranks inside their forward and
backward.  SummedGradients and BlockNodeClassification make ``architecture.train()`` (trainable.py:41-103) act like one process
holding all rows.  The reference has no distributed code; this is the multi-GPU form of the same layers."""
from __future__ import annotations

import torch

from . import sparse
from .protocol import Layer

class _BlockLoop(torch.autograd.Function):
    """The K-iteration loop over a vertex block as one autograd node.  The loop is a polynomial F(A_hat) applied to H0
    (H_K = (1-a)^K A^K H0 + a sum_{k<K} (1-a)^k A^k H0), so dH0 = F(A_hat)^T g -- and for the symmetric A_hat of an undirected
    graph that is F(A_hat) g: the backward IS the same sharded propagation, applied to the output gradient."""

    @staticmethod
    def forward(ctx, H0, layer):
        ctx.layer = layer
        return layer._propagate("forward", H0.detach())

    @staticmethod
    def backward(ctx, g):
        return ctx.layer._propagate("backward", g.contiguous()), None


class _BlockDroppedLoop(torch.autograd.Function):
    """The K training-mode iterations with per-iteration edge dropout over a vertex block.  Linear in H0, and every A_k is
    regenerated from the counter RNG, so nothing but the degree scales (K x n_buf floats) is kept for the backward."""

    @staticmethod
    def forward(ctx, H0, layer, seed, first):
        sg, p, K = layer.dropout_graph, layer.graph_dropout, layer.iterations
        scales = sg.dropped_scales(p, seed, first, K)
        ctx.layer, ctx.seed, ctx.first, ctx.scales = layer, seed, first, scales
        return sg.propagate_dropped(H0.detach(), layer.restart_probability, K, p, seed, first, scales)

    @staticmethod
    def backward(ctx, g):
        layer = ctx.layer
        return layer.dropout_graph.propagate_dropped_backward(g.contiguous(), layer.restart_probability, layer.iterations,
                                                              layer.graph_dropout, ctx.seed, ctx.first, ctx.scales), None, None, None


class ShardedPPRLoop(Layer):
    """The K PPRIteration layers of APPNP (filter.py:34-35) for a model that holds ONE vertex block of the graph: every rank
    builds the same layer stack over the rows of its block (the Dense layers before it act row by row, so they need no
    communication), and this layer propagates the block's H0 with the other ranks through ``ShardedGraph.propagate``.

        sg = ShardedGraph(my_entries, vals, bounds)
        model = gnntf.Trainable(features_of_my_rows)
        model.add(gnntf.Dense(64, activation=gnntf.relu)); H0 = model.add(gnntf.Dense(num_classes))
        model.add(ShardedPPRLoop(H0, sg, 0.1, 10))
        local_labels = model.predict(gnntf.NodeClassification(my_local_node_ids))

    Training works too when the adjacency is symmetric and constant (an undirected graph, no edge dropout): the backward of
    the loop is then the loop itself applied to the gradient (see _BlockLoop), the parameter gradients of the row-wise layers
    are summed over the ranks by ``SummedGradients`` and the task is a ``BlockNodeClassification``.

    ``graph_dropout`` > 0 (APPNP's default is 0.5, filter.py:8) needs ``dropout_graph``: the same rows built with
    ``ShardedGraph(..., edge_dropout=True)``.  In training mode every iteration then drops and re-normalises its edges exactly as
    the one-GPU PPRLoop does -- same masks, whatever the partition -- and the backward sends the halo rows of A_k^T g back to
    their owners; directed graphs are fine on this path.  Every rank must have called gnntf.set_seed with the same seed."""

    def __build__(self, architecture, H0: Layer, graph: "ShardedGraph", restart_probability: float = 0.1, iterations: int = 10,
                  symmetric: bool = True, graph_dropout: float = 0.0, dropout_graph: "ShardedGraph" = None):
        if architecture.top_shape()[0] != graph.n_local:
            raise Exception("ShardedPPRLoop: the architecture must hold this rank's %d rows" % graph.n_local)
        if graph_dropout != 0 and (dropout_graph is None or not dropout_graph.edge_dropout or dropout_graph.n_local != graph.n_local):
            raise Exception("ShardedPPRLoop: graph_dropout needs dropout_graph = ShardedGraph(same rows, edge_dropout=True)")
        self.H0, self.graph, self.restart_probability, self.iterations = H0, graph, restart_probability, iterations
        self.symmetric, self.graph_dropout, self.dropout_graph = symmetric, graph_dropout, dropout_graph
        self._states = dict()
        return architecture.top_shape()

    def _propagate(self, which, H0):
        C = H0.shape[1]
        H0 = sparse._padded(H0.to(torch.float32), sparse.friendly_width(C, self.graph.n_global))      # odd class counts run at a line-friendly row width
        state = self._states.get(which)
        if state is None or tuple(state.H0.shape) != tuple(H0.shape):
            state = self._states[which] = self.graph.make_state(H0.clone())
        elif self.graph.row_order is not None:
            state.H0_user.copy_(H0)
            state.H0.copy_(H0.index_select(0, self.graph.row_order))
        else:
            state.H0.copy_(H0)
        out = self.graph.propagate(state, self.restart_probability, self.iterations)
        return out.clone() if out.shape[1] == C else out[:, :C].contiguous()

    def __forward__(self, architecture, features):
        H0 = self.H0.value
        if self.graph_dropout != 0 and architecture.is_training():
            seed, first = architecture._next_mask_stream(self.iterations)
            return _BlockDroppedLoop.apply(H0, self, seed, first)
        if torch.is_grad_enabled() and H0.requires_grad:
            if not self.symmetric:
                raise Exception("ShardedPPRLoop: gradients need a symmetric adjacency (the backward reuses the forward propagation)")
            return _BlockLoop.apply(H0, self)
        return self._propagate("forward", H0.detach())


class _BlockSpMM(torch.autograd.Function):
    """A_hat . X over a vertex block (one halo exchange) as an autograd node; for the symmetric A_hat of an undirected graph the
    backward is the same product applied to the gradient."""

    @staticmethod
    def forward(ctx, X, layer):
        ctx.layer = layer
        return layer._aggregate("forward", X.detach())

    @staticmethod
    def backward(ctx, g):
        return ctx.layer._aggregate("backward", g.contiguous()), None


class ShardedGCNLayer(Layer):
    """GCNLayer (gcn.py:77-89), dropout(activation((A_hat . X) . W + b)), for a model that holds ONE vertex block: the aggregation
    is one iteration of the vertex-block propagation with a = 0 (pack, pairwise exchange, fused SpMM), the transform acts row by
    row.  Constant adjacency (graph_dropout = 0, what the reference's GCN uses in eval mode and GCNII always); training needs a
    symmetric A_hat, like ShardedPPRLoop, and the same SummedGradients / BlockNodeClassification pair."""

    def __build__(self, architecture, graph: "ShardedGraph", outputs: int, activation=None, bias: bool = True, dropout: float = 0,
                  symmetric: bool = True):
        from .blocks import relu
        if architecture.top_shape()[0] != graph.n_local:
            raise Exception("ShardedGCNLayer: the architecture must hold this rank's %d rows" % graph.n_local)
        self.graph, self.symmetric = graph, symmetric
        self.W = architecture.create_var((architecture.top_shape()[1], outputs))
        self.b = architecture.create_var((1, outputs), "zero") if bias else 0
        self.activation = relu if activation is None else activation
        self.dropout = dropout
        self._states = dict()
        return (architecture.top_shape()[0], outputs)

    def _aggregate(self, which, X):
        X = X.to(torch.float32).contiguous()
        state = self._states.get(which)
        if state is None or tuple(state.H0.shape) != tuple(X.shape):
            state = self._states[which] = self.graph.make_state(X.clone())
        elif self.graph.row_order is not None:
            state.H0_user.copy_(X)
            state.H0.copy_(X.index_select(0, self.graph.row_order))
        else:
            state.H0.copy_(X)
        return self.graph.propagate(state, 0.0, 1).clone()            # (1 - 0) A_hat X + 0 * X

    def __forward__(self, architecture, features):
        from .blocks import affine
        if torch.is_grad_enabled() and features.requires_grad:
            if not self.symmetric:
                raise Exception("ShardedGCNLayer: gradients need a symmetric adjacency (the backward reuses the forward product)")
            aggregated = _BlockSpMM.apply(features, self)
        else:
            aggregated = self._aggregate("forward", features.detach())
        return architecture.dropout(affine(aggregated, self.W, self.b, self.activation), self.dropout)


class _BlockMixStep(torch.autograd.Function):
    """(1-a) A_hat H + a H0 over a vertex block with H and H0 distinct (the aggregation of a GCNII layer); symmetric A_hat:
    dH = (1-a) A_hat g -- the same step applied to g with a zero mix term -- and dH0 = a g."""

    @staticmethod
    def forward(ctx, H, H0, layer):
        ctx.layer = layer
        return layer._step("forward", H.detach(), H0.detach())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gH = ctx.layer._step("backward", g, torch.zeros_like(g)) if ctx.needs_input_grad[0] else None
        gH0 = g * ctx.layer.a if ctx.needs_input_grad[1] else None
        return gH, gH0, None


class ShardedGCNIILayer(Layer):
    """GCNIILayer (gcn.py:7-27), dropout(act(((1-a) A_hat H + a H0) . ((1-b) I + b W))), b = beta_transformer(l / (k+1)), for a
    model that holds ONE vertex block: the aggregation is one iteration of the vertex-block propagation started from this
    layer's input with H0's value as the mix term; the C x C transform acts row by row.  Constant adjacency (graph_dropout = 0);
    gradients need a symmetric A_hat, as for the other block layers."""

    def __build__(self, architecture, graph: "ShardedGraph", H0: Layer, a: float, l: float, k: int = 0, activation=None,
                  beta_transformer=None, dropout: float = 0.5, regularization=True, symmetric: bool = True):
        import math
        from .blocks import linear
        if architecture.top_shape()[0] != graph.n_local:
            raise Exception("ShardedGCNIILayer: the architecture must hold this rank's %d rows" % graph.n_local)
        width = architecture.top_shape()[1]
        self.W = architecture.create_var((width, width), "zero", regularize=regularization)
        self.graph, self.H0, self.a, self.l, self.k, self.symmetric = graph, H0, a, l, k, symmetric
        self.activation = linear if activation is None else activation
        self.beta_transformer = math.log1p if beta_transformer is None else beta_transformer
        self.dropout = dropout
        self._states = dict()
        return architecture.top_shape()

    def _step(self, which, H, H0):
        H0 = H0.to(torch.float32).contiguous()
        state = self._states.get(which)
        if state is None or tuple(state.H0.shape) != tuple(H0.shape):
            state = self._states[which] = self.graph.make_state(H0.clone())
        elif self.graph.row_order is not None:
            state.H0_user.copy_(H0)
            state.H0.copy_(H0.index_select(0, self.graph.row_order))
        else:
            state.H0.copy_(H0)
        return self.graph.propagate(state, self.a, 1, start=H.to(torch.float32).contiguous()).clone()

    def __forward__(self, architecture, features):
        b = self.beta_transformer(self.l / (self.k + 1))
        eye = torch.eye(self.W.shape[1], device=self.W.device, dtype=self.W.dtype)
        transform = (1 - b) * eye + b * self.W
        H0 = self.H0.value
        if torch.is_grad_enabled() and (features.requires_grad or H0.requires_grad):
            if not self.symmetric:
                raise Exception("ShardedGCNIILayer: gradients need a symmetric adjacency (the backward reuses the forward product)")
            tradeoff = _BlockMixStep.apply(features, H0, self)
        else:
            tradeoff = self._step("forward", features.detach(), H0.detach())
        return architecture.dropout(self.activation(sparse.dense(tradeoff, transform) if tradeoff.is_cuda else torch.matmul(tradeoff, transform)),
                                    self.dropout)


class SummedGradients:
    """Optimizer wrapper for models that hold one vertex block each: before every step the gradients of the (replicated)
    parameters are summed over the ranks, so every rank applies the same update -- what one process holding all rows would
    compute, given a task that scales its local loss by the GLOBAL item count (BlockNodeClassification).  Terms of the objective
    that do not depend on the rows -- the weight decay -- are held by every rank alike: train() reads ``replicas`` and lets each
    rank contribute 1/replicas of them, so the default ``regularization`` gives the one-process update.
    Use as ``architecture.train(..., optimizer=lambda params: SummedGradients(torch.optim.Adam(params, ...), comm))``."""

    def __init__(self, optimizer, comm):
        self.optimizer, self.comm = optimizer, comm
        self.replicas = comm.size

    def zero_grad(self, set_to_none=True):
        self.optimizer.zero_grad(set_to_none=set_to_none)

    def step(self):
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                self.comm.all_reduce(p.grad)
        self.optimizer.step()

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    @property
    def state(self):
        return self.optimizer.state


class BlockNodeClassification:
    """NodeClassification (graph_predictor.py:10-31) over the labelled nodes of ONE vertex block: ``nodes`` are LOCAL row ids.
    loss() has the GLOBAL mean cross entropy as its value on every rank (so early stopping decides alike everywhere) and this
    rank's share of it as its gradient (SummedGradients adds the shares up); evaluate() is the global accuracy."""

    def __init__(self, nodes, labels, comm):
        from .tasks import NodeClassification
        self.local = NodeClassification(nodes, labels) if len(nodes) else None
        self.comm, self.count = comm, len(nodes)
        t = torch.tensor([float(self.count)], dtype=torch.float64)
        total = comm.all_reduce(t.to(self._device()) if self._device().type == "cuda" else t)
        self.total = float(total.item())
        self.nodes, self.labels = nodes, labels

    def _device(self):
        from .params import default_device
        return default_device()

    def predict(self, features):
        return self.local.predict(features) if self.local is not None else torch.zeros(0, dtype=torch.int64, device=features.device)

    def loss(self, features):
        share = self.local.loss(features) * (self.count / self.total) if self.local is not None else features.sum() * 0.0
        everyone = self.comm.all_reduce(share.detach().clone().reshape(1))
        return share + (everyone.reshape(()) - share.detach())

    def evaluate(self, features):
        right = float(self.local.evaluate(features)) * self.count if self.local is not None else 0.0
        t = torch.tensor([right], dtype=torch.float64, device=features.device)
        return float(self.comm.all_reduce(t).item()) / self.total
