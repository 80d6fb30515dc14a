"""Sparse containers and the differentiable propagation ops over libgnx.so.

Stands where the reference uses tf.sparse.SparseTensor and tf.sparse.sparse_dense_matmul
(reference gnntf/core/gnn/graph_manipulation.py:31, gnntf/core/gnn/architectures/filter.py:19,
gnntf/core/gnn/architectures/gcn.py:88).
"""
from __future__ import annotations

import math
from ctypes import byref, c_float, c_int64, c_void_p

import numpy as np
import torch

from . import _native as nat


class SparseCOO:
    """What graph2adj returns: an UNSORTED COO that may hold duplicates, exactly like the
    tf.sparse.SparseTensor of the reference (graph_manipulation.py:31).
    ``indices`` int64 [nnz, 2], ``values`` float32 [nnz], ``dense_shape`` (rows, cols)."""

    def __init__(self, indices, values, dense_shape):
        if isinstance(indices, torch.Tensor):
            self.indices = indices.to(torch.int64).reshape(-1, 2)
            self.values = torch.as_tensor(values, dtype=torch.float32, device=self.indices.device).reshape(-1)
        else:
            self.indices = torch.from_numpy(np.ascontiguousarray(np.asarray(indices, dtype=np.int64).reshape(-1, 2)))
            self.values = torch.from_numpy(np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1)))
        if self.indices.shape[0] != self.values.shape[0]:
            raise Exception("SparseCOO: indices and values disagree on the number of entries")
        self.dense_shape = (int(dense_shape[0]), int(dense_shape[1]))

    @property
    def shape(self):
        return self.dense_shape

    def to(self, device):
        return SparseCOO(self.indices.to(device), self.values.to(device), self.dense_shape)


def as_coo(graph) -> SparseCOO:
    """Accepts a SparseCOO, a torch sparse COO tensor, a scipy sparse matrix or an
    (indices, values, shape) triple."""
    if isinstance(graph, SparseCOO):
        return graph
    if isinstance(graph, torch.Tensor) and graph.is_sparse:
        return SparseCOO(graph._indices().t().contiguous(), graph._values(), graph.shape)
    if hasattr(graph, "tocoo"):
        m = graph.tocoo()
        return SparseCOO(np.stack([m.row, m.col], axis=1), m.data, m.shape)
    if isinstance(graph, (tuple, list)) and len(graph) == 3:
        return SparseCOO(*graph)
    raise Exception("Unsupported graph container: " + str(type(graph)))


class DeviceGraph:
    """Owner of a gnx_graph_t (device CSR built from the COO)."""

    def __init__(self, coo: SparseCOO = None, device=None, csr=None):
        self._h = c_void_p()
        lib = nat.lib()
        if csr is not None:
            rowptr, colidx, vals, shape = csr
            nat.require_cuda(rowptr, colidx, vals)
            if (rowptr.dtype, colidx.dtype, vals.dtype) != (torch.int64, torch.int32, torch.float32):
                raise Exception("DeviceGraph: a CSR needs int64 rowptr, int32 colidx and float32 values")
            if not (rowptr.device == colidx.device == vals.device):
                raise Exception("DeviceGraph: the CSR arrays live on different devices")
            if rowptr.numel() != shape[0] + 1 or colidx.numel() != vals.numel():
                raise Exception("DeviceGraph: CSR array lengths do not match the shape")
            self.device = rowptr.device
            self._keep = (rowptr.contiguous(), colidx.contiguous(), vals.contiguous())
            with nat.on_device(self.device):
                nat.check(lib.gnx_graph_create_csr(shape[0], shape[1], self._keep[1].numel(), nat.ptr(self._keep[0]),
                                                   nat.ptr(self._keep[1]), nat.ptr(self._keep[2]), nat.current_stream(),
                                                   byref(self._h)))
            self._keep = None
        else:
            device = torch.device(device if device is not None else "cuda")
            if device.type != "cuda":
                raise Exception("gnntf: graphs live on the GPU only; there is no CPU fallback")
            self.device = device
            idx = coo.indices.to(device).contiguous()
            val = coo.values.to(device).contiguous()
            self.device = idx.device                        # with its index ("cuda" -> "cuda:0")
            with nat.on_device(device):
                nat.check(lib.gnx_graph_create_coo(coo.dense_shape[0], coo.dense_shape[1], idx.shape[0], nat.ptr(idx),
                                                   nat.ptr(val), nat.current_stream(), byref(self._h)))
        n_rows, n_cols, nnz_e, nnz_c = c_int64(), c_int64(), c_int64(), c_int64()
        nat.check(lib.gnx_graph_info(self._h, byref(n_rows), byref(n_cols), byref(nnz_e), byref(nnz_c)))
        self.n_rows, self.n_cols = n_rows.value, n_cols.value
        self.nnz_entries, self.nnz = nnz_e.value, nnz_c.value

    @property
    def handle(self):
        return self._h

    def csr_arrays(self, with_rows=False):
        """Copies of (rowptr int64, colidx int32, raw values float32[, rowidx int32])."""
        rowptr = torch.empty(self.n_rows + 1, dtype=torch.int64, device=self.device)
        colidx = torch.empty(self.nnz, dtype=torch.int32, device=self.device)
        vals = torch.empty(self.nnz, dtype=torch.float32, device=self.device)
        rows = torch.empty(self.nnz, dtype=torch.int32, device=self.device) if with_rows else None
        with nat.on_device(self.device):
            nat.check(nat.lib().gnx_graph_export(self._h, nat.ptr(rowptr), nat.ptr(colidx), nat.ptr(vals), nat.ptr(rows),
                                                 nat.current_stream()))
        return (rowptr, colidx, vals, rows) if with_rows else (rowptr, colidx, vals)

    def last_kernel(self) -> str:
        return (nat.lib().gnx_graph_last_kernel(self._h) or b"").decode()

    def reserve(self, C, transposed=False, k_loop=False):
        """Builds NOW what the launches otherwise build on first use, sized for feature rows of up to ``C`` floats (the long-row
        slab; with ``transposed`` the transposed structure a backward needs; with ``k_loop`` the relabelled copy appnp_propagate
        runs narrow widths on): those lazy builds allocate and synchronise, which a stream under hipGraph capture must not see
        (a launch that would have to grow something there raises instead).  Call before capturing (gnx_graph_reserve)."""
        flags = (nat.RESERVE_TRANSPOSED if transposed else 0) | (nat.RESERVE_K_LOOP if k_loop else 0)
        with nat.on_device(self.device):
            nat.check(nat.lib().gnx_graph_reserve(self._h, int(C), flags, nat.current_stream()))

    def set_row_window(self, window_rows):
        """Declares that the vertex numbering of THIS graph carries locality (a community / breadth-first order): launches take the
        rows in windows of ``window_rows`` consecutive ids (degree-binned inside a window) and narrow widths stay off the
        degree-relabelled copy; 0 = the default global degree bins (gnx_graph_set_row_window).  Same sums, other launch order."""
        with nat.on_device(self.device):
            nat.check(nat.lib().gnx_graph_set_row_window(self._h, int(window_rows), nat.current_stream()))
        self.row_window = int(window_rows)

    def set_dropout_counter(self, counter):
        """``counter``: a one-element int64 device tensor added to every dropout stream id used with this graph (read by the
        kernels when they run), or None.  Lets a captured training step draw fresh masks on every replay."""
        if counter is not None and (counter.dtype != torch.int64 or counter.numel() != 1 or counter.device != self.device):
            raise Exception("set_dropout_counter: needs a one-element int64 tensor on the graph's device")
        self._counter = counter                                   # keep it alive while the handle points at it
        nat.check(nat.lib().gnx_graph_set_dropout_counter(self._h, nat.ptr(counter)))

    def __del__(self):
        try:
            if self._h:
                nat.lib().gnx_graph_destroy(self._h)
                self._h = c_void_p()
        except Exception:
            pass


class Adjacency:
    """What GNN.get_adjacency returns: a device graph + one set of (normalised, possibly
    dropped-out) values + the diagonal weight of an added identity.  Usable with
    ``gnntf.spmm(adj, H)`` wherever the reference calls tf.sparse.sparse_dense_matmul."""

    def __init__(self, graph: DeviceGraph, vals: torch.Tensor = None, diag: torch.Tensor = None, vals_t: torch.Tensor = None):
        self.graph = graph
        self.vals = vals          # values in coalesced-CSR order (None: the handle's raw values)
        self.diag = diag
        self.vals_t = vals_t      # the same values in the order of the transposed structure (backward), or None

    def transposed_values(self):
        """Values in transposed order, permuted once and kept (a constant adjacency is reused by every backward)."""
        if self.vals_t is None:
            out = torch.empty(self.graph.nnz, dtype=torch.float32, device=self.graph.device)
            with nat.on_device(self.graph.device):
                nat.check(nat.lib().gnx_graph_permute_values_t(self.graph.handle, nat.ptr(self.vals), nat.ptr(out), nat.current_stream()))
            self.vals_t = out
        return self.vals_t

    @property
    def shape(self):
        return (self.graph.n_rows, self.graph.n_cols)


class DroppedAdjacency(Adjacency):
    """The dropped + symmetrically re-normalised adjacency of one training iteration (layered.py:47-50 + gnn.py:41-42) WITHOUT
    its nnz-sized value array: only the N degree scales are computed up front; the SpMM kernels produce every entry's weight
    (D[row] * dropout(raw)) * D[col] from the counter RNG while they gather (gnx_spmm_dropped) -- forward and transposed, bit
    for bit the values gnx_graph_normalize would have written.  ``.vals`` materialises them on demand (custom layers).
    PRECONDITION: finite features and finite degree scales.  A dropped entry is SKIPPED here (its row is not gathered), whereas
    the reference keeps it as an explicit zero (tf.nn.dropout on G.values, layered.py:50), so 0 * inf or 0 * NaN in the
    gathered row -- or a NaN scale from a negative column sum -- makes a NaN there and not here.  For exact NaN propagation
    use the materialised form (``normalize(graph, "symmetric", "none", p, seed, stream)`` + ``spmm``), which multiplies every
    stored entry."""

    def __init__(self, graph: DeviceGraph, p, seed, stream_id, D=None):
        super().__init__(graph, None, None, None)
        self.p, self.seed, self.stream_id = float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFFFFFFFFFF
        self.D = D if D is not None else dropped_degree_scales(graph, self.p, self.seed, self.stream_id, 1)[0]
        self._vals = None

    @property
    def vals(self):
        if self._vals is None:
            self._vals = normalize(self.graph, "symmetric", "none", self.p, self.seed, self.stream_id).vals
        return self._vals

    @vals.setter
    def vals(self, value):
        self._vals = value

    def transposed_values(self):
        if self.vals_t is None:
            self.vals_t = normalize(self.graph, "symmetric", "none", self.p, self.seed, self.stream_id, transposed_only=True).vals_t
        return self.vals_t


def dropped_degree_scales(graph: DeviceGraph, p, seed, first_stream, n_streams) -> torch.Tensor:
    """D = divide_no_nan(1, sqrt(column sums of the dropped values)) (gnn.py:41) for ``n_streams`` consecutive dropout streams,
    [n_streams, n]: ONE pass over the structure for all of them (gnx_graph_colsum_streams)."""
    D = torch.empty((n_streams, graph.n_cols), dtype=torch.float32, device=graph.device)
    with nat.on_device(graph.device):
        nat.check(nat.lib().gnx_graph_colsum_streams(graph.handle, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                     int(first_stream) & 0xFFFFFFFFFFFFFFFF, int(n_streams), nat.ptr(D), nat.current_stream()))
        nat.check(nat.lib().gnx_degree_scale(nat.ptr(D), D.numel(), nat.NORM["symmetric"], 0, nat.current_stream()))
    return D


def can_fuse_dropout(graph: DeviceGraph, p) -> bool:
    return graph.nnz_entries == graph.nnz and graph.n_rows == graph.n_cols and p > 0


def dropped_adjacency(graph: DeviceGraph, p, seed, stream_id, D=None) -> Adjacency:
    """A training iteration's adjacency: the fused form when the graph allows it (no duplicate COO entries, square),
    else the materialised one.  ``D``: its degree scales if already known (dropped_degree_scales)."""
    if can_fuse_dropout(graph, p):
        return DroppedAdjacency(graph, p, seed, stream_id, D=D)
    return normalize(graph, "symmetric", "none", p, seed, stream_id)


def normalize(graph: DeviceGraph, normalized="symmetric", add_eye="none", dropout=0.0, seed=0, stream_id=0,
              transposed_only=False) -> Adjacency:
    """GNN.get_adjacency on the device (reference gnn.py:36-50).  ``transposed_only``: write the values in the
    order of the transposed structure only (all a backward pass needs)."""
    if normalized not in nat.NORM:
        raise Exception("Invalid matrix normalization")
    if add_eye not in nat.EYE:
        raise Exception("Invalid add_eye option")
    vals = torch.empty(graph.nnz, dtype=torch.float32, device=graph.device)
    diag = torch.empty(graph.n_rows, dtype=torch.float32, device=graph.device) if add_eye != "none" else None
    fn = nat.lib().gnx_graph_normalize_t if transposed_only else nat.lib().gnx_graph_normalize
    with nat.on_device(graph.device):
        nat.check(fn(graph.handle, nat.NORM[normalized], nat.EYE[add_eye], float(dropout), int(seed) & 0xFFFFFFFFFFFFFFFF,
                     int(stream_id) & 0xFFFFFFFFFFFFFFFF, nat.ptr(vals), nat.ptr(diag), nat.current_stream()))
    return Adjacency(graph, None, diag, vals_t=vals) if transposed_only else Adjacency(graph, vals, diag)


def _as_f32_rows(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.float32:
        x = x.float()
    if x.dim() != 2:
        raise Exception("propagation expects a 2-D feature matrix")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    return x


def _same_device(g, *tensors):
    """Raw pointers cross the C ABI: every operand must live on the graph's device."""
    for t in tensors:
        if t is not None and t.device != g.device:
            raise Exception(f"spmm: operand on {t.device}, the graph lives on {g.device}")


def _launch(adj: Adjacency, X, H0, beta, alpha, act, transposed=False, out=None, out_rows=None):
    g = adj.graph
    nat.require_cuda(X, H0)
    _same_device(g, X, H0, out, out_rows, adj.diag)
    X = _as_f32_rows(X)
    rows_in = g.n_rows if transposed else g.n_cols
    rows_out = g.n_cols if transposed else g.n_rows
    if X.shape[0] != rows_in:
        raise Exception(f"spmm: features have {X.shape[0]} rows, adjacency expects {rows_in}")
    C = X.shape[1]
    if out is None:
        out = torch.empty((rows_out, C), dtype=torch.float32, device=X.device)
    elif (tuple(out.shape) != (rows_out, C) or out.dtype != torch.float32 or out.stride(1) != 1 or not out.is_cuda):
        raise Exception("spmm: bad output buffer")
    ldh0 = 0
    if H0 is not None:
        H0 = _as_f32_rows(H0)
        if tuple(H0.shape) == (1, C) and rows_out != 1:
            H0, ldh0 = H0.contiguous(), 0                       # one row for every output row (bias)
        elif tuple(H0.shape) != (rows_out, C):
            raise Exception("spmm: H0 shape mismatch")
        else:
            ldh0 = H0.stride(0)
    if isinstance(adj, DroppedAdjacency) and out_rows is None:       # weights produced inside the kernel
        with nat.on_device(X.device):
            nat.check(nat.lib().gnx_spmm_dropped(g.handle, nat.ptr(adj.D), adj.p, adj.seed, adj.stream_id, 1 if transposed else 0,
                                                 nat.ptr(X), X.stride(0), C, nat.ptr(H0), ldh0, float(beta), float(alpha), int(act),
                                                 nat.ptr(out), out.stride(0), nat.current_stream()))
        return out
    if transposed:
        fn, values = nat.lib().gnx_spmm_tv, adj.transposed_values()
    else:
        if adj.vals is None and adj.vals_t is not None:
            raise Exception("spmm: this adjacency only holds transposed-order values")
        fn, values = nat.lib().gnx_spmm, adj.vals
    with nat.on_device(X.device):
        if out_rows is not None:                                # result row i -> out[out_rows[i]]
            if transposed or out_rows.dtype != torch.int32 or out_rows.numel() != rows_out or not out_rows.is_cuda:
                raise Exception("spmm: bad output row map")
            nat.check(nat.lib().gnx_spmm_scatter(g.handle, nat.ptr(values), nat.ptr(adj.diag), nat.ptr(X), X.stride(0), C, nat.ptr(H0),
                                                 ldh0, float(beta), float(alpha), int(act), nat.ptr(out_rows), nat.ptr(out),
                                                 out.stride(0), nat.current_stream()))
        else:
            nat.check(fn(g.handle, nat.ptr(values), nat.ptr(adj.diag), nat.ptr(X), X.stride(0), C, nat.ptr(H0),
                         ldh0, float(beta), float(alpha), int(act), nat.ptr(out), out.stride(0),
                         nat.current_stream()))
    return out


def _launch_chained(adj: "DroppedAdjacency", X, H0, beta, alpha, prescaled, D_next, skip_empty=False):
    """One forward training iteration inside a loop (gnx_spmm_dropped_chained): X carries its column scale when ``prescaled``,
    the result carries ``D_next`` (the next iteration's column scale) unless that is None.  ``skip_empty``: rows without entries
    are left untouched (every iteration but the last: nobody gathers them; the library ignores it on graphs where somebody does)."""
    g = adj.graph
    nat.require_cuda(X, H0)
    _same_device(g, X, H0, adj.D, D_next)
    if X.shape[0] != g.n_cols or tuple(H0.shape) != (g.n_rows, X.shape[1]) or not X.is_contiguous() or not H0.is_contiguous():
        raise Exception("chained propagation: bad operand shapes")
    out = torch.empty((g.n_rows, X.shape[1]), dtype=torch.float32, device=X.device)
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_spmm_dropped_chained(g.handle, nat.ptr(adj.D), adj.p, adj.seed, adj.stream_id, 1 if prescaled else 0,
                                                     nat.ptr(D_next), nat.ptr(X), X.stride(0), X.shape[1], nat.ptr(H0), H0.stride(0),
                                                     float(beta), float(alpha), nat.ACT_NONE | (nat.ACT_SKIP_EMPTY if skip_empty else 0),
                                                     nat.ptr(out), out.stride(0), nat.current_stream()))
    return out


def _launch_back(adj: "DroppedAdjacency", X, prescaled, D_next, S_in, s_alpha, s_beta, S_out, y_beta, Y_out, skip_empty=False):
    """One backward training iteration inside a loop (gnx_spmm_dropped_back): acc = A_k^T X over the transposed structure, weights
    made in the kernel; S_out = s_beta acc + s_alpha S_in (S_in may be S_out), Y_out = y_beta acc * D_next (skipped when None)."""
    g = adj.graph
    nat.require_cuda(X, S_in, S_out)
    _same_device(g, X, S_in, S_out, Y_out, adj.D, D_next)
    C = X.shape[1]
    if any(t is not None and (tuple(t.shape) != (g.n_rows, C) or not t.is_contiguous() or t.dtype != torch.float32) for t in (X, S_in, S_out, Y_out)):
        raise Exception("chained backward: bad operand shapes")
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_spmm_dropped_back(g.handle, nat.ptr(adj.D), adj.p, adj.seed, adj.stream_id, 1 if prescaled else 0, nat.ptr(D_next),
                                                  nat.ptr(X), C, C, nat.ptr(S_in), C, float(s_alpha), float(s_beta), nat.ptr(S_out), C,
                                                  float(y_beta), nat.ptr(Y_out), C, nat.ACT_SKIP_EMPTY if skip_empty else nat.ACT_NONE,
                                                  nat.current_stream()))


def _backward_chained(adjs, g, a):
    """dH0 of K chained training iterations for the upstream gradient ``g``: g_k = (1-a) A_k^T g_{k+1}, dH0 = g_0 + a (g_1 + ... +
    g_K), as K calls of gnx_spmm_dropped_back -- every call adds its g_k to the running sum in its epilogue and hands the next call
    its operand pre-scaled by that call's column scale, so no gradient of an iteration is kept, no per-entry scale is gathered
    and no separate summation pass exists."""
    K = len(adjs)
    g = _as_f32_rows(g).contiguous()
    S = torch.empty_like(g)
    X = g
    for k in range(K - 1, -1, -1):
        first, last = k == K - 1, k == 0
        Y = None if last else torch.empty_like(g)
        # rows without entries: their g_k is 0 -- after the first call their sum is final and their Y row is never gathered
        _launch_back(adjs[k], X, not first, None if last else adjs[k - 1].D, g if first else S, a if first else 1.0,
                     (1.0 - a) if last else a * (1.0 - a), S, 1.0 - a, Y, skip_empty=not first)
        X = Y
    return S


def launch_rows(adj: Adjacency, X, H0, beta, alpha, rows, out, act=nat.ACT_NONE):
    """The fused step over a graph that holds a SUBSET of the output rows (the interior or the boundary rows of
    a vertex block): result row r is written to out[rows[r]] and mixes in H0[rows[r]] (gnx_spmm_rows)."""
    g = adj.graph
    nat.require_cuda(X, H0, rows, out)
    _same_device(g, X, H0, rows, out)
    X = _as_f32_rows(X)
    if X.shape[0] != g.n_cols:
        raise Exception(f"spmm: features have {X.shape[0]} rows, adjacency expects {g.n_cols}")
    C = X.shape[1]
    if rows.dtype != torch.int32 or rows.numel() != g.n_rows or not rows.is_contiguous():
        raise Exception("spmm: bad row map")
    if out.dtype != torch.float32 or out.dim() != 2 or out.shape[1] != C or out.stride(1) != 1:
        raise Exception("spmm: bad output buffer")
    ldh0 = 0
    if H0 is not None:
        H0 = _as_f32_rows(H0)
        if tuple(H0.shape) != tuple(out.shape):
            raise Exception("spmm: H0 shape mismatch")
        ldh0 = H0.stride(0)
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_spmm_rows(g.handle, nat.ptr(adj.vals), nat.ptr(X), X.stride(0), C, nat.ptr(H0), ldh0, float(beta),
                                          float(alpha), int(act), nat.ptr(rows), nat.ptr(out), out.stride(0),
                                          nat.current_stream()))
    return out


class _SpMM(torch.autograd.Function):
    """out = A . X ; backward dX = A^T . g (what tf.GradientTape derives for filter.py:19)."""

    @staticmethod
    def forward(ctx, X, adj):
        ctx.adj = adj
        return _launch(adj, X, None, 1.0, 0.0, nat.ACT_NONE)

    @staticmethod
    def backward(ctx, g):
        return _launch(ctx.adj, g.contiguous(), None, 1.0, 0.0, nat.ACT_NONE, transposed=True), None


class _PPRStep(torch.autograd.Function):
    """out = (A . H)*(1-a) + H0*a in one kernel (filter.py:19-21);
    backward dH = (1-a) A^T g, dH0 = a g."""

    @staticmethod
    def forward(ctx, H, H0, adj, a):
        ctx.adj, ctx.a = adj, a
        return _launch(adj, H, H0, 1.0 - a, a, nat.ACT_NONE)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gH = _launch(ctx.adj, g, None, 1.0 - ctx.a, 0.0, nat.ACT_NONE, transposed=True) if ctx.needs_input_grad[0] else None
        gH0 = g * ctx.a if ctx.needs_input_grad[1] else None
        return gH, gH0, None, None


PAD_WIDTHS = True      # tools flip this for A/B runs


PAD_MIN_ROWS = 1 << 16      # smaller graphs are launch-bound: the pad / slice launches would cost more than the loads save


def lines_per_row(C: int) -> float:
    """Average number of 128-byte lines a row of C floats touches when rows are stored back to back (row r starts at byte 4 C r)."""
    size, step = 4 * C, math.gcd(4 * C, 128)
    offsets = range(0, 128, step)
    return sum((off + size + 127) // 128 for off in offsets) / len(offsets)


def friendly_width(C: int, n_rows: int = PAD_MIN_ROWS) -> int:
    """The row width (floats) the K-iteration loops run at.  A gather moves whole 128-byte lines and the kernels load 16 bytes
    per lane when rows are 16-byte aligned: rows of 7 ... 31 floats are padded to the next power of two (a row then never
    straddles a line it does not fill: C = 9 ... 15 run 26 % faster as 16, 20 ... 24 as 32), wider ones to the next multiple of
    4 (C = 41 or 47 -- odd class counts -- would otherwise fall back to 4-byte loads) -- and on to the first multiple of 4 up to
    the next multiple of 32 at which a row touches no more lines than its size needs (round 6: C = 56, 224-byte rows, half of which
    span three lines: 4.50 ms per iteration on the config-4 graph against 3.75 as 64; 44 -> 48, 52 / 56 / 60 -> 64; 40 and 48
    stay, their rows never span a third line).  The pad columns are zero and stay zero."""
    if not PAD_WIDTHS or C <= 6 or n_rows < PAD_MIN_ROWS:     # up to 6 floats the pad / un-pad copies cost what the wider loads save
        return C
    if C <= 32:
        return 1 << (C - 1).bit_length()
    Cp = (C + 3) // 4 * 4
    best = Cp
    for wider in range(Cp + 4, (Cp + 31) // 32 * 32 + 1, 4):
        if lines_per_row(wider) < lines_per_row(best) - 1e-9:
            best = wider
    return best


def _padded(H: torch.Tensor, Cp: int) -> torch.Tensor:
    if H.shape[1] == Cp:
        return H
    out = torch.zeros((H.shape[0], Cp), dtype=torch.float32, device=H.device)
    out[:, :H.shape[1]] = H
    return out


class _PPRLoop(torch.autograd.Function):
    """K PPRIteration steps as ONE autograd node.  The step is linear in H, so the backward needs no
    stored activations: g_k = (1-a) A_k^T g_{k+1}, dH0 = g_0 + a * sum_k g_{k+1}.  In training mode
    every iteration has its own dropped + re-normalised adjacency A_k (filter.py:18 calls get_adjacency
    each time); the counter RNG lets the backward REGENERATE A_k from (seed, stream id) instead of
    keeping K value arrays alive.
    ``relu`` (the reference's per-iteration activation, filter.py:22): H_k = relu(Z_k) in every launch's epilogue; the backward then
    needs the sign of every Z_k, so the K outputs ARE kept (what the layer-by-layer form keeps anyway) and every gradient is
    masked by H_k > 0 before it goes through A_k^T: gz_k = g_k * (H_k > 0), g_{k-1} = (1-a) A_k^T gz_k, dH0 = g_0 + a sum_k gz_k."""

    @staticmethod
    def forward(ctx, H0, make_adj, a, K, relu=False):
        ctx.make_adj, ctx.a, ctx.K, ctx.relu = make_adj, a, K, relu
        act = nat.ACT_RELU if relu else nat.ACT_NONE
        H0 = _as_f32_rows(H0).contiguous()
        ctx.C = C = H0.shape[1]
        H0 = _padded(H0, friendly_width(C, H0.shape[0]))
        H = H0
        first = make_adj(0, False) if K > 0 else None
        kept = []
        if K > 1 and isinstance(first, DroppedAdjacency) and not relu:
            # weights made in the kernels (only the K degree-scale vectors exist): the next iteration's column scale rides out with
            # the rows, so from k = 1 on no per-entry scale gather is left (gnx_spmm_dropped_chained)
            adjs = [first] + [make_adj(k, False) for k in range(1, K)]
            chained = all(isinstance(adj, DroppedAdjacency) and adj.graph is first.graph for adj in adjs)
            ctx.chained = chained and first.graph.n_rows == first.graph.n_cols
            for k, adj in enumerate(adjs):
                if chained:
                    # (rows without entries are a * H0 in the result and gathered by nobody: only the last iteration writes them)
                    H = _launch_chained(adj, H, H0, 1.0 - a, a, prescaled=k > 0, D_next=adjs[k + 1].D if k + 1 < K else None,
                                        skip_empty=k + 1 < K)
                else:
                    H = _launch(adj, H, H0, 1.0 - a, a, nat.ACT_NONE)
        else:
            for k in range(K):           # one adjacency alive at a time (a materialised one is an nnz-sized array)
                H = _launch(first if k == 0 else make_adj(k, False), H, H0, 1.0 - a, a, act)
                if relu:
                    kept.append(H)
        if relu:
            ctx.save_for_backward(*kept)
        return H if H.shape[1] == C else H[:, :C].contiguous()

    @staticmethod
    def backward(ctx, g):
        # dH0 = g_0 + a (g_1 + ... + g_K): the gradients of the iterations are KEPT (as many as a tenth of the card's memory
        # holds, at most 15) and added up by one pass (gnx_linear_combination) instead of a read-modify-write of dH0 per iteration
        g = _padded(g.contiguous(), friendly_width(ctx.C, g.shape[0]))
        if getattr(ctx, "chained", False):
            adjs = [ctx.make_adj(k, True) for k in range(ctx.K)]
            if all(isinstance(adj, DroppedAdjacency) for adj in adjs):
                gH0 = _backward_chained(adjs, g, ctx.a)
                return (gH0 if gH0.shape[1] == ctx.C else gH0[:, :ctx.C].contiguous()), None, None, None, None
        outs = ctx.saved_tensors if ctx.relu else None
        room = int(0.1 * torch.cuda.get_device_properties(g.device).total_memory) // max(g.numel() * 4, 1)
        limit = max(2, min(LINCOMB_TERMS - 1, room))
        pending, total = [], None
        for k in range(ctx.K - 1, -1, -1):
            if outs is not None:
                g = _relu_mask(g, outs[k])
            pending.append((g, ctx.a))
            if len(pending) >= limit:
                total, pending = linear_combination(([(total, 1.0)] if total is not None else []) + pending), []
            g = _launch(ctx.make_adj(k, True), g, None, 1.0 - ctx.a, 0.0, nat.ACT_NONE, transposed=True)
        pending.append((g, 1.0))
        gH0 = linear_combination(([(total, 1.0)] if total is not None else []) + pending)
        return (gH0 if gH0.shape[1] == ctx.C else gH0[:, :ctx.C].contiguous()), None, None, None, None


LINCOMB_TERMS = 16


def linear_combination(terms) -> torch.Tensor:
    """sum_j coef_j * tensor_j for up to 16 equally shaped float32 device tensors, ``terms`` = [(tensor, coef), ...], in ONE pass
    (gnx_linear_combination; terms are added in list order).  Returns a new tensor."""
    if not 1 <= len(terms) <= LINCOMB_TERMS:
        raise Exception("linear_combination: 1 to %d terms" % LINCOMB_TERMS)
    # the kernel reads 16 bytes per lane: a contiguous VIEW with a storage offset (a row or column slice of a padded buffer)
    # need not be 16-byte aligned -- such a term is copied to a fresh allocation first
    tensors = [t if t.is_contiguous() and t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format) for t, _ in terms]
    first = tensors[0]
    nat.require_cuda(*tensors)
    for t in tensors:
        if t.shape != first.shape or t.dtype != torch.float32 or t.device != first.device:
            raise Exception("linear_combination: the terms must be float32 tensors of one shape on one device")
    out = torch.empty_like(first)
    k = len(tensors)
    ptrs = (c_void_p * k)(*[t.data_ptr() for t in tensors])
    coefs = (c_float * k)(*[float(c) for _, c in terms])
    with nat.on_device(first.device):
        nat.check(nat.lib().gnx_linear_combination(k, ptrs, coefs, first.numel(), nat.ptr(out), nat.current_stream()))
    return out


def ppr_loop(make_adj, H0: torch.Tensor, a: float, iterations: int, relu: bool = False) -> torch.Tensor:
    """``iterations`` fused PPR steps starting from H0; ``make_adj(k, for_backward)`` returns the Adjacency
    of iteration k (called again, with the same k and for_backward=True, during the backward, where only the
    transposed-order values are needed).  ``relu``: relu after every step (filter.py:22)."""
    return _PPRLoop.apply(H0, make_adj, float(a), int(iterations), bool(relu))


class _SpMMBiasAct(torch.autograd.Function):
    """out = act(A . Y + bias) in one kernel (bias broadcast through the H0 operand, relu in the epilogue);
    backward: g' = g * (out > 0), dY = A^T g', dbias = column sums of g'."""

    @staticmethod
    def forward(ctx, Y, bias, adj, relu):
        out = _launch(adj, Y, bias, 1.0, 1.0, nat.ACT_RELU if relu else nat.ACT_NONE)
        ctx.adj, ctx.relu, ctx.has_bias = adj, relu, bias is not None
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _relu_mask(g, out) if ctx.relu else g
        g = g.contiguous()
        gY = _launch(ctx.adj, g, None, 1.0, 0.0, nat.ACT_NONE, transposed=True) if ctx.needs_input_grad[0] else None
        gb = g.sum(dim=0, keepdim=True) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gY, gb, None, None


def spmm_bias_act(adj: Adjacency, Y: torch.Tensor, bias=None, relu=False) -> torch.Tensor:
    """act(A . Y + bias) fused; ``bias`` is [1, C] or None."""
    return _SpMMBiasAct.apply(Y, bias, adj, bool(relu))


def spmm(adj: Adjacency, X: torch.Tensor) -> torch.Tensor:
    """Drop-in for tf.sparse.sparse_dense_matmul(adj, X); differentiable w.r.t. X."""
    return _SpMM.apply(X, adj)


def ppr_step(adj: Adjacency, H: torch.Tensor, H0: torch.Tensor, a) -> torch.Tensor:
    """One fused PPRIteration step.  ``a`` may be a float (fused kernel) or a tensor
    (a trainable teleport probability: un-fused so autograd reaches it)."""
    if isinstance(a, torch.Tensor):
        return spmm(adj, H) * (1 - a) + H0 * a
    return _PPRStep.apply(H, H0, adj, float(a))


def appnp_propagate(adj: Adjacency, H0: torch.Tensor, a: float = 0.1, iterations: int = 10, relu: bool = False) -> torch.Tensor:
    """The eval-mode K-iteration loop as ONE library call with two ping-pong buffers
    (no autograd, no per-layer .value caching) -- the measured hot path.  ``relu``: the reference's per-iteration activation
    (filter.py:22,28,35) in every iteration's epilogue (gnx_appnp_propagate_act)."""
    g = adj.graph
    nat.require_cuda(H0)
    H0 = _as_f32_rows(H0).contiguous()
    if g.n_rows != g.n_cols or H0.shape[0] != g.n_rows:
        raise Exception("appnp_propagate: needs a square graph matching H0")
    C = H0.shape[1]
    H0 = _padded(H0, friendly_width(C, H0.shape[0]))
    out = torch.empty_like(H0)
    work = torch.empty_like(H0) if iterations > 1 else None
    with nat.on_device(H0.device):
        nat.check(nat.lib().gnx_appnp_propagate_act(g.handle, nat.ptr(adj.vals), nat.ptr(adj.diag), nat.ptr(H0), float(a),
                                                    int(iterations), H0.shape[1], nat.ACT_RELU if relu else nat.ACT_NONE,
                                                    nat.ptr(out), nat.ptr(work), nat.current_stream()))
    return out if out.shape[1] == C else out[:, :C].contiguous()


def gather_rows(X: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[r] = X[idx[r]] through the library's halo-packing kernel."""
    nat.require_cuda(X, idx)
    X = _as_f32_rows(X)
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty((idx.numel(), X.shape[1]), dtype=torch.float32, device=X.device)
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_gather_rows(nat.ptr(X), X.stride(0), nat.ptr(idx), idx.numel(), X.shape[1], nat.ptr(out),
                                            out.stride(0), nat.current_stream()))
    return out


# ---- the dense ends of the path: matrix-core kernels of libgnx.so (csrc/gnx_dense.hip) ----------------------------------
def _dense_launch(X, W, bias, relu):
    nat.require_cuda(X, W, bias)
    X, W = _as_f32_rows(X), _as_f32_rows(W)
    if X.shape[1] != W.shape[0]:
        raise Exception(f"dense: features have {X.shape[1]} columns, the weights expect {W.shape[0]}")
    out = torch.empty((X.shape[0], W.shape[1]), dtype=torch.float32, device=X.device)
    b = None if bias is None else bias.to(torch.float32).reshape(-1).contiguous()
    if b is not None and b.numel() != W.shape[1]:
        raise Exception("dense: bias width mismatch")
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_dense(nat.ptr(X), X.stride(0), X.shape[0], X.shape[1], nat.ptr(W), W.stride(0), W.shape[1], nat.ptr(b),
                                      nat.ACT_RELU if relu else nat.ACT_NONE, nat.ptr(out), out.stride(0), nat.current_stream()))
    return out


def _relu_mask(g, out):
    """g * (out > 0) in one pass (relu's backward; ``out`` is a relu output, never NaN)."""
    return torch.ops.aten.threshold_backward(g.contiguous(), out, 0.0)


def _dense_wgrad(X, G):
    """dW = X^T . G on the matrix cores (gnx_dense_wgrad): row slabs, partial sums added in a fixed order."""
    X, G = _as_f32_rows(X), _as_f32_rows(G)
    n, F = X.shape
    O = G.shape[1]
    slabs = max(1, min(2048, (n + 255) // 256, (1 << 28) // max(F * O, 1)))        # scratch capped at 1 GiB
    work = torch.empty(slabs * F * O, dtype=torch.float32, device=X.device)
    dW = torch.empty((F, O), dtype=torch.float32, device=X.device)
    with nat.on_device(X.device):
        nat.check(nat.lib().gnx_dense_wgrad(nat.ptr(X), X.stride(0), nat.ptr(G), G.stride(0), n, F, O, nat.ptr(dW), nat.ptr(work),
                                            work.numel(), nat.current_stream()))
    return dW


class _DenseAct(torch.autograd.Function):
    """out = act(X . W + b) on the matrix cores (layers.py:135-136; the transform of gcn.py:89).
    backward: g' = g * (out > 0); dX = g' . W^T through the same kernel; dW = X^T . g' through gnx_dense_wgrad (MFMA over row
    slabs); db = column sums of g' (a torch reduction)."""

    @staticmethod
    def forward(ctx, X, W, bias, relu):
        out = _dense_launch(X, W, bias, relu)
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.save_for_backward(X, W, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        X, W, out = ctx.saved_tensors
        g = _relu_mask(g, out) if ctx.relu else g
        g = g.contiguous()
        gX = _dense_launch(g, W.t().contiguous(), None, False) if ctx.needs_input_grad[0] else None
        gW = _dense_wgrad(X, g) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=0).reshape(ctx.bias_shape) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gX, gW, gb, None


def dense(X: torch.Tensor, W: torch.Tensor, bias=None, relu=False) -> torch.Tensor:
    """act(X . W + bias) for device tensors, through gnx_dense (float32 MFMA).  ``bias`` [1, O] / [O] or None."""
    return _DenseAct.apply(X, W, bias, bool(relu))


def _gcnii_launch(adj: Adjacency, H, H0, a, M, relu, keep_mixed):
    """gnx_gcnii_step; returns (out, T or None).  T = the mixed rows (A.H)(1-a) + H0 a, written by the same launch when kept."""
    g = adj.graph
    nat.require_cuda(H, H0, M)
    H, H0, M = _as_f32_rows(H).contiguous(), _as_f32_rows(H0).contiguous(), _as_f32_rows(M)
    C = H.shape[1]
    if g.n_rows != g.n_cols or H.shape[0] != g.n_rows or tuple(H0.shape) != tuple(H.shape) or tuple(M.shape) != (C, C):
        raise Exception("gcnii_step: shape mismatch")
    if adj.diag is not None:
        raise Exception("gcnii_step: add_eye adjacencies are not supported by the fused step")
    out = torch.empty_like(H)
    mixed = torch.empty_like(H) if keep_mixed or C not in (16, 32, 64) else None
    with nat.on_device(H.device):
        nat.check(nat.lib().gnx_gcnii_step(g.handle, nat.ptr(adj.vals), nat.ptr(H), nat.ptr(H0), float(a), C, nat.ptr(M), M.stride(0),
                                           nat.ACT_RELU if relu else nat.ACT_NONE, nat.ptr(out), nat.ptr(mixed), nat.current_stream()))
    return out, mixed


class _GCNIIStep(torch.autograd.Function):
    """out = act(T . M), T = (A . H)(1-a) + H0 a, as ONE launch that also leaves T in memory for the backward (gcn.py:22-27 under
    tf.GradientTape): with g' = g * (out > 0):  dM = T^T g' (gnx_dense_wgrad), dT = g' M^T (gnx_dense), dH = (1-a) A^T dT, dH0 = a dT."""

    @staticmethod
    def forward(ctx, H, H0, M, adj, a, relu):
        out, T = _gcnii_launch(adj, H, H0, a, M, relu, keep_mixed=True)
        ctx.adj, ctx.a, ctx.relu = adj, a, relu
        ctx.save_for_backward(T, M, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        T, M, out = ctx.saved_tensors
        g = (_relu_mask(g, out) if ctx.relu else g).contiguous()
        gM = _dense_wgrad(T, g) if ctx.needs_input_grad[2] else None
        gH = gH0 = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gT = _dense_launch(g, M.t().contiguous(), None, False)
            if ctx.needs_input_grad[0]:
                gH = _launch(ctx.adj, gT, None, 1.0 - ctx.a, 0.0, nat.ACT_NONE, transposed=True)
            if ctx.needs_input_grad[1]:
                gH0 = gT * ctx.a
        return gH, gH0, gM, None, None, None


def gcnii_step(adj: Adjacency, H: torch.Tensor, H0: torch.Tensor, a: float, M: torch.Tensor, relu=True) -> torch.Tensor:
    """act(((A . H)(1-a) + H0 a) . M), M = (1-b) I + b W (gcn.py:22-27) -- ONE fused launch for C in {16, 32, 64}: the mixed
    rows stay in LDS and meet M on the matrix cores (gnx_gcnii_step).  Without autograd they never reach HBM; when gradients are
    needed the same launch also writes them (dM = T^T g needs them), once, and the transform does not read them back.  Other
    widths run the fused SpMM+mix and then the matrix-core transform."""
    if torch.is_grad_enabled() and (H.requires_grad or H0.requires_grad or M.requires_grad):
        if isinstance(adj, DroppedAdjacency):                       # weights made inside the SpMM: the generic composition knows how
            return dense(ppr_step(adj, H, H0, a), M, None, relu)
        return _GCNIIStep.apply(H, H0, M, adj, float(a), bool(relu))
    return _gcnii_launch(adj, H, H0, a, M, relu, keep_mixed=False)[0]


class DeviceIndex:
    """Node ids / labels / edges of a task, checked ONCE on the host (range) and kept on the device: a task evaluates the same
    index lists every epoch, and on small graphs an upload + a device-side check per call would dominate the epoch.
    The kernels themselves never dereference an out-of-range id (NaN / -1 instead); device tensors handed in directly skip the
    host check."""

    def __init__(self, values, device, upper=None, what="node id"):
        if isinstance(values, torch.Tensor):
            self.tensor = values.to(device=device, dtype=torch.int64).contiguous()
        else:
            host = np.ascontiguousarray(np.asarray(values, dtype=np.int64))
            if upper is not None and host.size and (int(host.min()) < 0 or int(host.max()) >= upper):
                raise Exception(f"{what} out of range [0, {upper})")
            self.tensor = torch.from_numpy(host).to(device)
        self.upper = upper


def _as_index(x, device, upper, what):
    if isinstance(x, DeviceIndex):
        if x.tensor.device != device or (x.upper is not None and upper is not None and x.upper != upper):
            raise Exception("DeviceIndex built for another device / size")
        return x.tensor
    return DeviceIndex(x, device, upper, what).tensor


class _NodeCE(torch.autograd.Function):
    """mean_i CE(log_softmax(logits[nodes_i]), labels_i) fused over the listed nodes (graph_predictor.py:19-25)."""

    @staticmethod
    def forward(ctx, logits, nodes, labels):
        logits = _as_f32_rows(logits)
        m = nodes.numel()
        per_node = torch.empty(m + 256, dtype=torch.float32, device=logits.device)       # + scratch of the two-level mean
        mean = torch.empty(1, dtype=torch.float32, device=logits.device)
        with nat.on_device(logits.device):
            nat.check(nat.lib().gnx_node_ce(nat.ptr(logits), logits.stride(0), logits.shape[0], logits.shape[1], nat.ptr(nodes),
                                            nat.ptr(labels), m, nat.ptr(per_node), nat.ptr(mean), nat.current_stream()))
        ctx.save_for_backward(logits, nodes, labels)
        return mean.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, nodes, labels = ctx.saved_tensors
        grad = torch.zeros((logits.shape[0], logits.shape[1]), dtype=torch.float32, device=logits.device)
        g = g.to(torch.float32).reshape(1).contiguous()
        with nat.on_device(logits.device):
            nat.check(nat.lib().gnx_node_ce_backward(nat.ptr(logits), logits.stride(0), logits.shape[0], logits.shape[1], nat.ptr(nodes),
                                                     nat.ptr(labels), nodes.numel(), nat.ptr(g), nat.ptr(grad), grad.stride(0),
                                                     nat.current_stream()))
        return grad, None, None


def node_ce(logits: torch.Tensor, nodes, labels) -> torch.Tensor:
    """The NodeClassification loss on the device in two small launches (gather + log-softmax + CE, then the mean).
    ``nodes`` / ``labels``: host sequences (range-checked here), DeviceIndex objects (checked when built) or device tensors."""
    nat.require_cuda(logits)
    nodes = _as_index(nodes, logits.device, logits.shape[0], "node id")
    labels = _as_index(labels, logits.device, logits.shape[1], "label")
    if nodes.numel() != labels.numel() or nodes.numel() == 0:
        raise Exception("node_ce: nodes and labels must be equally long and non-empty")
    return _NodeCE.apply(logits, nodes, labels)


def node_argmax(logits: torch.Tensor, nodes=None) -> torch.Tensor:
    """argmax over the rows of ``nodes`` (all rows when None) in one launch; ties go to the lowest class."""
    nat.require_cuda(logits)
    logits = _as_f32_rows(logits.detach())
    idx = None if nodes is None else _as_index(nodes, logits.device, logits.shape[0], "node id")
    m = logits.shape[0] if idx is None else idx.numel()
    out = torch.empty(m, dtype=torch.int64, device=logits.device)
    with nat.on_device(logits.device):
        nat.check(nat.lib().gnx_node_argmax(nat.ptr(logits), logits.stride(0), logits.shape[0], logits.shape[1], nat.ptr(idx), m,
                                            nat.ptr(out), nat.current_stream()))
    return out


# ---- sparse input features: the first Dense of the pre-MLP as an SpMM over the rows of W ---------------------------------
class SparseRows:
    """A feature matrix that is mostly zeros (Cora: 1433 columns, 1.3 % non-zero), held as a device CSR.  It flows through
    the layer stack in place of the dense tensor until the first Dense consumes it: Dropout on it drops stored entries
    (tf.nn.dropout leaves zeros zero, layers.py:180-181), Dense on it is X . W computed as an SpMM whose "dense operand" is
    W -- 4F bytes of X per row become 8 bytes per stored entry, and W (F x 64 floats) stays cache resident."""

    def __init__(self, graph: DeviceGraph, dropout=0.0, seed=0, stream_id=0):
        self.graph, self.p, self.seed, self.stream_id = graph, float(dropout), int(seed), int(stream_id)
        self.shape = (graph.n_rows, graph.n_cols)

    @classmethod
    def from_dense(cls, X: torch.Tensor):
        idx = torch.nonzero(X)
        return cls(DeviceGraph(SparseCOO(idx, X[idx[:, 0], idx[:, 1]], tuple(X.shape)), device=X.device))

    def with_dropout(self, p, seed, stream_id):
        if self.p != 0:
            raise Exception("SparseRows: dropout applied twice before a Dense layer")
        return SparseRows(self.graph, p, seed, stream_id)

    def adjacency(self) -> Adjacency:
        """The stored values (dropped per entry and rescaled while a dropout is pending) as an Adjacency over X."""
        if self.p == 0:
            return Adjacency(self.graph, None)
        return normalize(self.graph, "none", "none", self.p, self.seed, self.stream_id)


class _SparseDense(torch.autograd.Function):
    """out = act(X . W + b), X sparse: forward = fused SpMM (bias through the H0 operand, relu in the epilogue);
    backward dW = X^T . g' = the transposed SpMM; X is an input and gets no gradient."""

    @staticmethod
    def forward(ctx, W, bias, adj, relu):
        out = _launch(adj, W, bias, 1.0, 1.0, nat.ACT_RELU if relu else nat.ACT_NONE)
        ctx.adj, ctx.relu, ctx.has_bias = adj, relu, bias is not None
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = (_relu_mask(g, out) if ctx.relu else g).contiguous()
        gW = _launch(ctx.adj, g, None, 1.0, 0.0, nat.ACT_NONE, transposed=True) if ctx.needs_input_grad[0] else None
        gb = g.sum(dim=0, keepdim=True) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gW, gb, None, None


def sparse_dense(rows: SparseRows, W: torch.Tensor, bias=None, relu=False) -> torch.Tensor:
    """act(X . W + bias) for sparse X (``bias`` [1, O] or None)."""
    return _SparseDense.apply(W, bias, rows.adjacency(), bool(relu))


# ---- link head: the logit of every listed edge in one launch -------------------------------------------------------------
class _EdgeScores(torch.autograd.Function):
    """logit_i = sum_c F[u_i, c] F[v_i, c] (r[c] or 1)  (graph_predictor.py:122-126); backward scatters into dF with atomics;
    the DistMult weights' gradient (C numbers) is a reduction over the edges, done with torch on the gathered rows."""

    @staticmethod
    def forward(ctx, F, edges, r):
        F = _as_f32_rows(F)
        rr = None if r is None else r.to(torch.float32).reshape(-1).contiguous()
        out = torch.empty(edges.shape[0], dtype=torch.float32, device=F.device)
        with nat.on_device(F.device):
            nat.check(nat.lib().gnx_edge_scores(nat.ptr(F), F.stride(0), F.shape[0], F.shape[1], nat.ptr(edges), edges.shape[0], nat.ptr(rr),
                                                nat.ptr(out), nat.current_stream()))
        ctx.save_for_backward(F, edges, rr)
        ctx.r_shape = None if r is None else tuple(r.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        F, edges, rr = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        gF = gr = None
        if ctx.needs_input_grad[0]:
            gF = torch.zeros((F.shape[0], F.shape[1]), dtype=torch.float32, device=F.device)
            with nat.on_device(F.device):
                nat.check(nat.lib().gnx_edge_scores_backward(nat.ptr(F), F.stride(0), F.shape[0], F.shape[1], nat.ptr(edges), edges.shape[0], nat.ptr(rr),
                                                             nat.ptr(g), nat.ptr(gF), gF.stride(0), nat.current_stream()))
        if rr is not None and ctx.needs_input_grad[2]:
            gr = (g[:, None] * F[edges[:, 0]] * F[edges[:, 1]]).sum(0).reshape(ctx.r_shape)
        return gF, None, gr


def edge_scores(F: torch.Tensor, edges, r=None) -> torch.Tensor:
    """Logits of the listed edges ([m, 2] node ids): <F[u], F[v]> or, with DistMult weights r [C, 1], <F[u] * F[v], r>."""
    nat.require_cuda(F, r)
    e = _as_index(edges, F.device, F.shape[0], "edge endpoint").reshape(-1, 2)
    return _EdgeScores.apply(F, e, r)
