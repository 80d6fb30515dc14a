"""Graph architectures over the HIP propagation path: GNN (adjacency owner), PPRIteration,
APPNP, GCNLayer, GCN.

Mirrors reference gnntf/core/gnn/gnn.py:29-50, gnntf/core/gnn/architectures/filter.py:6-35
and gnntf/core/gnn/architectures/gcn.py:77-113.  Where the reference calls
tf.sparse.sparse_dense_matmul on a freshly re-normalised tf.sparse tensor, these layers call
the fused gfx950 kernels through gnntf.sparse; the eval-mode normalised adjacency is
computed once per (normalized, add_eye) and cached, because it is constant.
"""
from __future__ import annotations

import math

import torch

from . import ordering, sparse
from .blocks import Concatenate, Dense, Dropout, affine, is_relu, linear, relu
from .params import default_device
from .protocol import Layer
from .training import Trainable


class Structural(Layer):
    """gnn.py:5-26: trainable per-node embeddings (``bipartite`` rows in one variable, the rest in another), optionally
    l2-normalised, prepended to the incoming features; with zero-width features the embeddings ARE the features."""

    def __build__(self, architecture, dims: int = 16, l2_contraint: bool = False, bipartite: int = 0, **kwargs):
        rows, width = architecture.top_shape()
        self.l2_contraint = l2_contraint
        self.embeddings = architecture.create_var((bipartite, dims), **kwargs)
        self.embeddings2 = architecture.create_var((rows - bipartite, dims), **kwargs)
        return rows, dims + width

    def __forward__(self, architecture, features):
        table = self.embeddings2 if self.embeddings.shape[0] == 0 else torch.cat([self.embeddings, self.embeddings2], dim=0)
        if self.l2_contraint:
            table = torch.nn.functional.normalize(table, dim=1, eps=1e-12)
        return table if features.shape[1] == 0 else torch.cat([table, features], dim=1)


class GNN(Trainable):
    """gnn.py:29-50."""

    def __init__(self, graph, features, preprocessor: Layer = None, reorder=None):
        """``reorder`` (opt-in, not in the reference): store the graph and the feature rows with the vertices relabelled; every
        [N, .] tensor inside the model then lives in that order, and the model's OUTPUT is put back into the caller's order, so
        tasks, labels and node ids are unaffected.  Results agree with the unordered model to float32 rounding.
        ``"degree"``: stable order of descending entry count (5-20 % faster propagation at C <= 128: rows sharing a wave, their
        H0/out rows and the hub rows become neighbours in memory).
        ``"locality"``: community by community (gnntf.ordering.locality_order: label propagation) with the library told so
        (row windows, one window of the numbering per XCD at a time) -- for graphs that HAVE communities, as the reference's
        citation datasets do: -16 ... -31 % per propagation at C = 7 ... 256 on a planted-partition x power-law graph of 10M
        vertices.  On a graph without communities (R-MAT) the order finds none (``locality_share`` fails
        ordering.found_communities) and the model keeps the default order (``reorder_used`` is then None); so do graphs of
        a few windows, which fit the caches in any order.
        Measurements: profiles/NOTES.md round 5."""
        super().__init__(features)
        self._order = self._newid = None
        self.reorder_used, self.locality_share = None, None
        if isinstance(graph, sparse.DeviceGraph):
            if reorder is not None:
                raise Exception("GNN: reorder needs the COO adjacency, not a ready DeviceGraph")
            self.graph = graph
        else:
            coo = sparse.as_coo(graph).to(default_device())
            if reorder not in (None, "degree", "locality"):
                raise Exception("Invalid reorder option")
            n = coo.dense_shape[0]
            order = None
            if reorder == "degree":
                order = torch.argsort(torch.bincount(coo.indices[:, 0], minlength=n), descending=True, stable=True)    # new id -> old id
            elif reorder == "locality":
                if coo.dense_shape[0] != coo.dense_shape[1]:
                    raise Exception("GNN: reorder=\"locality\" needs a square adjacency")
                order = ordering.locality_order(coo.indices, n)
            if order is not None:
                newid = torch.empty_like(order)
                newid[order] = torch.arange(n, device=order.device)
                if reorder == "locality":
                    # did the order find communities?  On a graph without them (R-MAT) it is a loss against the default: keep that
                    self.locality_share = ordering.share_within(coo.indices, newid, ordering.LOCALITY_WINDOW)
                    baseline = ordering.degree_order_share(coo.indices, n, ordering.LOCALITY_WINDOW)
                    if not ordering.found_communities(self.locality_share, n, ordering.LOCALITY_WINDOW, baseline):
                        order, reorder = None, None
            self.reorder_used = reorder
            if order is not None:
                if isinstance(self.features, sparse.SparseRows):
                    raise Exception("GNN: reorder needs dense input features")
                self._order, self._newid = order, newid
                coo = sparse.SparseCOO(newid[coo.indices], coo.values, coo.dense_shape)
                self.features = self.features.index_select(0, order)
            self.graph = sparse.DeviceGraph(coo, device=default_device())
            if reorder == "locality":
                self.graph.set_row_window(ordering.LOCALITY_WINDOW)
        self._adjacency_cache = dict()
        if preprocessor is not None:
            self.add(preprocessor)

    def __call__(self, features):
        if self._order is None:
            return super().__call__(features)
        if features is not self.features:                  # rows given in the caller's order
            features = features.index_select(0, self._order)
        return super().__call__(features).index_select(0, self._newid)

    def get_adjacency(self, graph_dropout=0.5, normalized="symmetric", add_eye="none"):
        """Edge dropout (training mode only) -> optional +I -> D^-1/2 A D^-1/2 by column sums
        (gnn.py:36-50), as one device call.  Returns an Adjacency for gnntf.spmm."""
        if normalized not in ("symmetric", "bipartite", "none"):
            raise Exception("Invalid matrix normalization")
        if graph_dropout != 0 and self.is_training():
            seed, stream = self._next_mask_stream()
            if normalized == "symmetric" and add_eye == "none":          # the values are produced inside the SpMM kernels
                return sparse.dropped_adjacency(self.graph, graph_dropout, seed, stream)
            return sparse.normalize(self.graph, normalized, add_eye, graph_dropout, seed, stream)
        key = (normalized, add_eye)
        if key not in self._adjacency_cache:
            self._adjacency_cache[key] = sparse.normalize(self.graph, normalized, add_eye)
        return self._adjacency_cache[key]


def _propagation_run(architecture: "GNN", H0_value, a, iterations, graph_dropout, with_relu=False):
    """``iterations`` PPR steps from H0 through the fused loop (what PPRLoop and a run of plain PPRIteration layers execute).
    ``with_relu``: relu after every step (the reference's ``activation`` argument, filter.py:22,28,35), in the kernels' epilogue.
    Returns (result, run) with run(k) = the value after the first k iterations (for the intermediate layers' lazy ``.value``)."""
    training = graph_dropout != 0 and architecture.is_training()
    cheap = True                                                # asking make_adj for an iteration's adjacency again costs nothing
    if training:
        seed, first = architecture._next_mask_stream(iterations)
        graph, p = architecture.graph, graph_dropout
        if sparse.can_fuse_dropout(graph, p):
            # the degree scales of all K iterations in one pass over the structure; kept (K x N floats) for the backward
            scales = sparse.dropped_degree_scales(graph, p, seed, first, iterations)
            make_adj = lambda k, bwd=False: sparse.dropped_adjacency(graph, p, seed, first + k, D=scales[k])
        elif graph.nnz * iterations * 4 <= (64 << 20):
            # small graph (launch-latency regime): keep the K materialised adjacencies for the backward (one permute launch
            # there instead of three launches to regenerate); large graphs regenerate to save K nnz-sized arrays
            kept = dict()

            def make_adj(k, bwd=False):
                if k not in kept:
                    kept[k] = sparse.normalize(graph, "symmetric", "none", p, seed, first + k)
                return kept.pop(k) if bwd else kept[k]
        else:
            cheap = False                                       # every call materialises an nnz-sized value array
            make_adj = lambda k, bwd=False: sparse.normalize(graph, "symmetric", "none", p, seed, first + k, transposed_only=bwd)
    else:
        adj = architecture.get_adjacency(graph_dropout)
        make_adj = lambda k, bwd=False: adj
    if not torch.is_grad_enabled() and not training:
        run = lambda k: sparse.appnp_propagate(make_adj(0, False), H0_value, a, k, relu=with_relu)
    else:
        run = lambda k: sparse.ppr_loop(make_adj, H0_value, a, k, relu=with_relu)
    return run(iterations), run, (make_adj if cheap else None)


class PPRIteration(Layer):
    """One APPNP power-iteration step (filter.py:6-22): activation(dropout((A.H)(1-a) + H0 a)).

    A RUN of such layers as user code builds it (reference demos/custom_layers.py:8-13: ``for _ in range(10):
    gnn.add(PPRIteration(H0, 0.1))``) executes as one fused loop when the layers are plain -- the same H0 layer, one float restart
    probability, the identity or relu as the activation of ALL of them (relu runs in the kernels' epilogue, gnx_appnp_propagate_act),
    identity restart transform, no feature dropout (in eval mode, where it does not act, any), one graph_dropout -- and the run starts from
    H0's own value: the same arithmetic and the same sequence of edge-dropout masks as layer by layer (bitwise on graphs below
    2^20 vertices; above, narrow widths run on the relabelled copy: float32 rounding), at the cost of the PPRLoop layer.  The last
    layer of the run holds the result; the ``.value`` of an intermediate layer is computed when somebody reads it.
    ``architecture.fuse_runs = False`` switches this off."""

    def __build__(self, architecture: GNN, H0: Layer, restart_probability: float = 0.1, activation=linear,
                  dropout: float = 0, graph_dropout: float = 0.5, restart_transform=linear):
        self.restart_probability = restart_probability
        self.H0 = H0
        self.dropout = dropout
        self.graph_dropout = graph_dropout
        self.activation = activation
        self.restart_transform = restart_transform
        return architecture.top_shape()  # preserves the feature shape

    def __forward__(self, architecture: GNN, features):
        self.G = architecture.get_adjacency(self.graph_dropout)
        a = self.restart_transform(self.restart_probability)
        mixed = sparse.ppr_step(self.G, features, self.H0.value, a)
        return self.activation(architecture.dropout(mixed, self.dropout))

    # ``.value`` (layered.py:79-81) may be pending after a fused run: computed on first read
    @property
    def value(self):
        pending = self.__dict__.get("_pending_value")
        if pending is not None:
            self.__dict__["_value"], self.__dict__["_pending_value"] = pending(), None
        return self.__dict__.get("_value")

    @value.setter
    def value(self, v):
        self.__dict__["_value"], self.__dict__["_pending_value"] = v, None

    def _plain(self, training=True):
        """Whether this layer can be one step of a fused run.  Feature dropout (filter.py:22) only acts in training mode
        (layered.py:44-45): an eval-mode run fuses whatever the layers' ``dropout`` says."""
        a = self.restart_probability
        return (type(self) is PPRIteration and isinstance(a, (int, float)) and not isinstance(a, bool)
                and (self.activation is linear or is_relu(self.activation))
                and self.restart_transform is linear and (self.dropout == 0 or not training) and self.output_regularize == 0)

    def __run__(self, architecture: GNN, features, stack, at):
        training = not isinstance(architecture, GNN) or architecture.is_training()
        if not self._plain(training) or not isinstance(architecture, GNN) or features is not getattr(self.H0, "value", None) \
                or not isinstance(features, torch.Tensor) or not features.is_cuda:
            return None
        run = [self]
        for layer in stack[at + 1:]:
            if not (isinstance(layer, PPRIteration) and layer._plain(training) and layer.H0 is self.H0 and layer is not self
                    and layer.restart_probability == self.restart_probability and layer.graph_dropout == self.graph_dropout
                    and (layer.activation is self.activation or (is_relu(layer.activation) and is_relu(self.activation)))
                    and all(layer is not seen for seen in run)):
                break
            run.append(layer)
        if len(run) < 2:
            return None
        out, upto, make_adj = _propagation_run(architecture, features, float(self.restart_probability), len(run), self.graph_dropout,
                                               with_relu=is_relu(self.activation))
        for k, layer in enumerate(run[:-1]):
            layer.__dict__["_value"], layer.__dict__["_pending_value"] = None, (lambda k=k: upto(k + 1))
        run[-1].value = out
        for k, layer in enumerate(run):                         # filter.py:18 leaves the iteration's adjacency in self.G
            layer.G = make_adj(k, False) if make_adj is not None else None
        return len(run), out


class PPRLoop(Layer):
    """``iterations`` PPRIteration layers (filter.py:34-35) collapsed into one layer and one autograd node:
    same arithmetic and the same sequence of edge-dropout masks as the layer-by-layer form, but no
    per-iteration ``.value`` tensors and no stored activations for the backward (they are not needed: the
    step is linear in H, and dropped adjacencies are regenerated from the counter RNG).  What ``APPNP(..., fused=True)`` builds (an
    explicit opt-in: the default keeps the reference's layer list and fuses at execution, PPRIteration.__run__); only the default
    identity activation / zero feature dropout can be collapsed."""

    def __build__(self, architecture: GNN, H0: Layer, restart_probability: float = 0.1, iterations: int = 10,
                  graph_dropout: float = 0.5):
        self.restart_probability = restart_probability
        self.H0 = H0
        self.iterations = iterations
        self.graph_dropout = graph_dropout
        return architecture.top_shape()

    def __forward__(self, architecture: GNN, features):
        return _propagation_run(architecture, self.H0.value, self.restart_probability, self.iterations, self.graph_dropout)[0]


class APPNP(GNN):
    """filter.py:25-35 -- https://arxiv.org/pdf/1810.05997.pdf"""

    def __init__(self, G, features, num_classes: int, a: float = 0.1, latent_dims=[64], iterations=10,
                 dropout=0.6, graph_dropout=0.5, activation=linear, fused=False, **kwargs):
        """Builds filter.py:30-35's layer list as it stands there: Dropout(0.5), one Dense per latent width, the output Dense (= H0),
        then ``iterations`` PPRIteration layers -- ``layers()`` has the reference's length, order and types, every iteration its own
        ``.value`` / ``.G``.  The container EXECUTES the run of plain PPRIteration layers as one fused loop (PPRIteration.__run__:
        same arithmetic, same sequence of edge-dropout masks, one autograd node in training mode; settled rows skipped, the relabelled
        copy at narrow widths, line-friendly padded widths), so the list costs what the collapsed form costs.
        ``fused=True`` (not in the reference) collapses the iterations into ONE PPRLoop layer explicitly: no per-iteration layer
        objects at all; needs a float restart probability and the identity activation."""
        super().__init__(G, features, **kwargs)
        self.add(Dropout(0.5))
        for latent_dim in latent_dims:
            self.add(Dense(latent_dim, activation=relu, dropout=dropout))
        H0 = self.add(Dense(num_classes, regularize=False))
        if fused:
            if a is None or isinstance(a, torch.Tensor) or activation is not linear:
                raise Exception("APPNP(fused=True) needs a float restart probability and the identity activation")
            self.add(PPRLoop(H0, a, iterations, graph_dropout=graph_dropout))
            return
        for _ in range(iterations):
            self.add(PPRIteration(H0, self.create_var() if a is None else a, graph_dropout=graph_dropout, activation=activation))


class GCNLayer(Layer):
    """gcn.py:77-89: dropout(activation((A.X).W + b)) -- aggregation first, at the input width, as the
    reference computes it.  ``transform_first=True`` opts into the algebraically equal A.(X.W) when the layer
    narrows the features (outputs < inputs): the SpMM then runs at the output width with bias + relu in its
    epilogue (half the gather traffic for 128 -> 64); float32 rounding differs by a few ulp."""

    def __build__(self, gcn, outputs: int, activation=relu, bias: bool = True, dropout: float = 0, graph_dropout: float = 0,
                  transform_first: bool = False):
        self.W = gcn.create_var((gcn.top_shape()[1], outputs))
        self.b = gcn.create_var((1, outputs), "zero") if bias else 0
        self.activation = activation
        self.dropout = dropout
        self.graph_dropout = graph_dropout
        self.transform_first = bool(transform_first) and outputs < gcn.top_shape()[1]
        return (gcn.top_shape()[0], outputs)

    def __forward__(self, gcn, features):
        adjacency = gcn.get_adjacency(self.graph_dropout)
        if self.transform_first:
            bias = self.b if isinstance(self.b, torch.Tensor) else None
            projected = affine(features, self.W, 0)
            if self.activation is relu or self.activation is linear:
                out = sparse.spmm_bias_act(adjacency, projected, bias, relu=self.activation is relu)
            else:
                out = self.activation(sparse.spmm_bias_act(adjacency, projected, bias))
            return gcn.dropout(out, self.dropout)
        aggregated_features = sparse.spmm(adjacency, features)
        return gcn.dropout(affine(aggregated_features, self.W, self.b, self.activation), self.dropout)


class GCNSpectralPreservingLayer(GCNLayer):
    """gcn.py:92-105: the GCN layer with its bias taken back out after the activation and the surviving half of the dropout
    doubled -- 2 * dropout(act((A.X).W + b) - b).  Same kernels as GCNLayer (the SpMM, then the matrix-core transform with the
    bias and a relu fused); the correction is an elementwise tail."""

    def __forward__(self, gcn, features):
        activated = affine(sparse.spmm(gcn.get_adjacency(self.graph_dropout), features), self.W, self.b, self.activation)
        return 2 * gcn.dropout(activated - self.b, self.dropout)


class GCN(GNN):
    """gcn.py:108-113 (the last layer keeps the default relu, as in the reference)."""

    def __init__(self, G, features, num_classes, latent_dims=[64], layer_type=GCNLayer, transform_first=False, **kwargs):
        super().__init__(G, features, **kwargs)
        extra = dict(transform_first=True) if transform_first else dict()
        for latent_dim in latent_dims:
            self.add(layer_type(latent_dim, graph_dropout=0.5, dropout=0.5, **extra))
        self.add(layer_type(num_classes, **extra))


class GCNIILayer(Layer):
    """gcn.py:7-27: dropout(act(((1-a) A.H + a H0) . ((1-b) I + b W))), b = beta_transformer(l / (k+1)).
    ONE launch per layer for C in {16, 32, 64} -- the mixed rows stay in LDS and meet (1-b) I + b W on the matrix cores
    (gnx_gcnii_step); in training the same launch also writes the mixed rows once (dW needs them) instead of a second launch
    reading them back."""

    def __build__(self, architecture, H0: Layer, a: float, l: float, k: int = 0, activation=linear,
                  beta_transformer=math.log1p, dropout: float = 0.5, graph_dropout: float = 0.5, regularization=True):
        width = architecture.top_shape()[1]
        self.W = architecture.create_var((width, width), "zero", regularize=regularization)
        self.a, self.l, self.k = a, l, k
        self.activation = activation
        self.dropout = dropout
        self.graph_dropout = graph_dropout
        self.H0 = H0
        self.beta_transformer = beta_transformer
        return architecture.top_shape()

    def __forward__(self, gcn, features):
        b = self.beta_transformer(self.l / (self.k + 1))
        eye = torch.eye(self.W.shape[1], device=self.W.device, dtype=self.W.dtype)
        transform = (1 - b) * eye + b * self.W
        adjacency = gcn.get_adjacency(self.graph_dropout)
        if features.is_cuda and adjacency.diag is None:
            fused_act = self.activation is relu or self.activation is linear
            out = sparse.gcnii_step(adjacency, features, self.H0.value, self.a, transform, relu=self.activation is relu)
            return gcn.dropout(out if fused_act else self.activation(out), self.dropout)
        tradeoff = sparse.ppr_step(adjacency, features, self.H0.value, self.a)
        return gcn.dropout(self.activation(torch.matmul(tradeoff, transform)), self.dropout)


class GCNIISpectralPreservingLayer(GCNIILayer):
    """gcn.py:30-51: GCNIILayer with a bias added before the activation and removed after it, and the dropout's survivors
    doubled: 2 * dropout(act(T.M + bias) - bias), T = (1-a) A.H + a H0, M = (1-b) I + b W.  The propagation + mix is the fused
    SpMM kernel; the transform carries the bias and the relu in the matrix-core kernel's epilogue."""

    def __build__(self, architecture, H0, a, l, k=0, **kwargs):
        shape = super().__build__(architecture, H0, a, l, k, **kwargs)
        self.bias = architecture.create_var((1, shape[1]), "zero")
        return shape

    def __forward__(self, gcn, features):
        b = self.beta_transformer(self.l / (self.k + 1))
        eye = torch.eye(self.W.shape[1], device=self.W.device, dtype=self.W.dtype)
        tradeoff = sparse.ppr_step(gcn.get_adjacency(self.graph_dropout), features, self.H0.value, self.a)
        activated = affine(tradeoff, (1 - b) * eye + b * self.W, self.bias, self.activation)
        return 2 * gcn.dropout(activated - self.bias, self.dropout)


class GCNII(GNN):
    """gcn.py:54-74 -- http://proceedings.mlr.press/v119/chen20v/chen20v.pdf"""

    def __init__(self, graph, features, num_classes, a: float = 0.1, l: float = 0.5, latent_dims=[64], iterations=64,
                 dropout=0.6, convolution_regularization=True, layer_type=GCNIILayer, **kwargs):
        super().__init__(graph, features, **kwargs)
        self.add(Dropout(dropout))
        for latent_dim in latent_dims:
            self.add(Dense(latent_dim, dropout=0, activation=relu))
        H0 = self.top_layer()
        for iteration in range(iterations):
            self.add(layer_type(H0, a, l, iteration, activation=relu, dropout=dropout, graph_dropout=0,
                                regularization=convolution_regularization))
        self.add(Dense(num_classes, dropout=0, regularize=False))


def leaky_relu(x):
    return torch.nn.functional.leaky_relu(x, 0.2)          # tf.nn.leaky_relu's default slope


class NGCFLayer(Layer):
    """gcn.py:116-135: with A = the BIPARTITE-normalised adjacency (rows scaled by 1 / column sum, gnn.py:43-45),
    l2_normalize(dropout(act((X * (A.X)).W1 + b1) + act((A.X).W2 + b2))).  The adjacency is taken ONCE, at build time
    (gcn.py:127) -- so a ``node_dropout`` is one fixed mask, drawn while the fresh architecture is still in training mode."""

    def __build__(self, gcn, outputs: int, activation=leaky_relu, bias: bool = True, dropout: float = 0, node_dropout: float = 0,
                  regularize: float = 1):
        rows, width = gcn.top_shape()
        spread = 1. / rows ** 0.5
        self.W1 = gcn.create_var((width, outputs), regularize=regularize, normalization=spread)
        self.W2 = gcn.create_var((width, outputs), regularize=regularize, normalization=spread)
        self.b1 = gcn.create_var((1, outputs), normalization=spread) if bias else 0
        self.b2 = gcn.create_var((1, outputs), normalization=spread) if bias else 0
        self.activation, self.dropout, self.node_dropout = activation, dropout, node_dropout
        self.adjacency = gcn.get_adjacency(self.node_dropout, add_eye="none", normalized="bipartite")
        return rows, outputs

    def __forward__(self, gcn, features):
        aggregated = sparse.spmm(self.adjacency, features)                       # the propagation kernel
        interaction = self.activation(affine(features * aggregated, self.W1, self.b1))
        message = self.activation(affine(aggregated, self.W2, self.b2))
        return torch.nn.functional.normalize(gcn.dropout(interaction + message, self.dropout), dim=1, eps=1e-12)


class NGCF(GNN):
    """gcn.py:138-154 -- https://dl.acm.org/doi/pdf/10.1145/3468264.3468552.  The closing Concatenate stacks the layers'
    outputs along axis 0 exactly like the reference's (layers.py:98-101), so rows 0..N-1 of the output are the FIRST layer's
    embeddings -- which is what the link tasks index."""

    def __init__(self, graph, features, num_classes: int, latent_dims=None, dropout=0.1, **kwargs):
        super().__init__(graph, features, **kwargs)
        widths = [num_classes] * 2 if latent_dims is None else list(latent_dims)
        stack = [self.add(NGCFLayer(width, regularize=0.0, dropout=dropout, output_regularize=1)) for width in widths + [num_classes]]
        self.add(Concatenate(stack))


class MLP(Trainable):
    """The graph-free baseline (reference gnntf/core/nn/architectures/mlp.py:6-12): it falls out of the
    generic layers; no propagation kernel is involved."""

    def __init__(self, features, num_classes: int, latent_dims=[64], dropout: float = 0.5):
        super().__init__(features)
        self.add(Dropout(dropout))
        for latent_dim in latent_dims:
            self.add(Dense(latent_dim, dropout=dropout, activation=relu))
        self.add(Dense(num_classes, regularize=False))
