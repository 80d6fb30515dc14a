"""The device side of the vertex-block path: what gnntf.sharded asks of a backend, as calls into libgnx.so (the product backend).
Tests on CPU ranks (gloo) supply their own checker backend with the same methods: nothing here is a CPU implementation."""
from __future__ import annotations

import torch

from . import _native as nat
from . import sparse


class NativeBackend:
    """libgnx.so through ctypes; every tensor lives on this rank's GPU."""

    def graph_from_coo(self, idx, vals, shape):
        return sparse.DeviceGraph(sparse.SparseCOO(idx, vals, shape), device=idx.device)

    def graph_from_csr(self, rowptr, colidx, vals, shape):
        return sparse.DeviceGraph(csr=(rowptr, colidx, vals, shape))

    def csr_arrays(self, graph, with_rows=False):
        return graph.csr_arrays(with_rows=with_rows)

    def colsum(self, graph):
        out = torch.empty(graph.n_cols, dtype=torch.float32, device=graph.device)
        with nat.on_device(graph.device):
            nat.check(nat.lib().gnx_graph_colsum(graph.handle, 0.0, 0, 0, nat.ptr(out), nat.current_stream()))
        return out

    def degree_scale(self, deg, normalized="symmetric"):
        with nat.on_device(deg.device):
            nat.check(nat.lib().gnx_degree_scale(nat.ptr(deg), deg.numel(), nat.NORM[normalized], 0, nat.current_stream()))
        return deg

    def scale_values(self, graph, row_scale, col_scale):
        out = torch.empty(graph.nnz, dtype=torch.float32, device=graph.device)
        row_scale = row_scale.contiguous() if row_scale is not None else None
        with nat.on_device(graph.device):
            nat.check(nat.lib().gnx_graph_scale_values(graph.handle, 0.0, 0, 0, nat.ptr(row_scale), nat.ptr(col_scale),
                                                       nat.ptr(out), nat.current_stream()))
        return out

    def spmm_mix(self, graph, vals, X, H0, beta, alpha, out, out_rows=None, rows=None, skip_empty=False):
        """out[i] = beta * (A X)[i] + alpha * H0[i];  ``rows``: the graph holds a subset of the output rows --
        result row r belongs to out[rows[r]] / H0[rows[r]];  ``out_rows``: scatter of the result only;
        ``skip_empty``: rows without entries already hold alpha * H0 and are left alone (GNX_ACT_SKIP_EMPTY)."""
        adj = sparse.Adjacency(graph, vals)
        act = nat.ACT_NONE | (nat.ACT_SKIP_EMPTY if skip_empty else 0)
        if rows is not None:
            sparse.launch_rows(adj, X, H0, beta, alpha, rows, out, act=act)
        else:
            sparse._launch(adj, X, H0, beta, alpha, act, out=out, out_rows=out_rows)

    def spmm_plain(self, graph, X, out):
        sparse._launch(sparse.Adjacency(graph, None), X, None, 1.0, 0.0, nat.ACT_NONE, out=out)

    def gather_rows(self, X, idx):
        return sparse.gather_rows(X, idx)

    # ---- the exchange plan of a block: gnx_halo_plan_* (layout + packing of both halves of the send buffer) -------------
    def halo_plan(self, rank, n_local, recv_pull, recv_push, send_pull, send_push, pull_src, push_graph):
        return NativeHaloPlan(rank, n_local, recv_pull, recv_push, send_pull, send_push, pull_src, push_graph)

    def halo_pack(self, plan, part, buf, send):
        """The chosen half (or both) of the send buffer from the local rows of ``buf`` (gnx_halo_pack)."""
        if plan.n_send == 0:
            return
        with nat.on_device(buf.device):
            nat.check(nat.lib().gnx_halo_pack(plan.handle, PARTS[part], nat.ptr(buf), buf.stride(0), buf.shape[1], nat.ptr(send),
                                              send.stride(0), nat.current_stream()))

    # ---- training with edge dropout on a vertex block (raw values; weights made inside the kernels) ----------
    def set_block(self, graph, row0_global, row0_buf, col_gid):
        """Dropout draws of this block are keyed by GLOBAL (row, col) from now on (gnx_graph_set_block)."""
        with nat.on_device(graph.device):
            nat.check(nat.lib().gnx_graph_set_block(graph.handle, int(row0_global), int(row0_buf), nat.ptr(col_gid),
                                                    nat.current_stream()))

    def colsum_streams(self, graph, p, seed, first_stream, n_streams):
        """[n_streams, n_cols]: this block's PARTIAL column sums of the dropped raw values, one row per dropout stream."""
        out = torch.empty((n_streams, graph.n_cols), dtype=torch.float32, device=graph.device)
        with nat.on_device(graph.device):
            nat.check(nat.lib().gnx_graph_colsum_streams(graph.handle, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                         int(first_stream) & 0xFFFFFFFFFFFFFFFF, int(n_streams), nat.ptr(out),
                                                         nat.current_stream()))
        return out

    def spmm_dropped(self, graph, D, p, seed, stream_id, transposed, X, H0, beta, alpha, out):
        """out = beta * (A_k X) + alpha * H0 (or A_k^T X) with A_k = the dropped + re-normalised block of dropout stream
        ``stream_id``; D: the degree scales of every column of the block for that stream."""
        adj = sparse.DroppedAdjacency(graph, p, seed, stream_id, D=D)
        sparse._launch(adj, X, H0, beta, alpha, nat.ACT_NONE, transposed=bool(transposed), out=out)

    def spmm_dropped_chained(self, graph, D, p, seed, stream_id, prescaled, D_next, X, H0, beta, alpha, out):
        """The forward spmm_dropped inside a loop: X carries its column scale when ``prescaled``; the result rows carry
        ``D_next`` (the next iteration's scale of every buffer column; row r's own entry is used) unless it is None."""
        with nat.on_device(graph.device):
            nat.check(nat.lib().gnx_spmm_dropped_chained(graph.handle, nat.ptr(D), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                         int(stream_id) & 0xFFFFFFFFFFFFFFFF, 1 if prescaled else 0, nat.ptr(D_next),
                                                         nat.ptr(X), X.stride(0), X.shape[1], nat.ptr(H0), H0.stride(0), float(beta),
                                                         float(alpha), nat.ACT_NONE, nat.ptr(out), out.stride(0), nat.current_stream()))

    def spmm_t_mix(self, graph, X, H0, beta, alpha, out):
        """out = beta * (A^T X) + alpha * H0 over the graph's own (raw) values."""
        adj = getattr(graph, "_plain_adjacency", None)
        if adj is None:
            adj = graph._plain_adjacency = sparse.Adjacency(graph, None)         # keeps the transposed-order values
        sparse._launch(adj, X, H0, beta, alpha, nat.ACT_NONE, transposed=True, out=out)


PARTS = {"all": nat.HALO_ALL, "pull": nat.HALO_PULL, "push": nat.HALO_PUSH}


class NativeHaloPlan:
    """gnx_halo_plan_t of one vertex block: the library computes the layout of the feature and send buffers and packs the
    outgoing rows; this object only keeps the handle and what it borrows alive."""

    def __init__(self, rank, n_local, recv_pull, recv_push, send_pull, send_push, pull_src, push_graph):
        from ctypes import byref, c_int64, c_void_p
        P = len(recv_pull)
        arr = lambda counts: (c_int64 * P)(*[int(c) for c in counts])
        self._keep = (pull_src, push_graph)                       # borrowed by the plan
        self.handle = c_void_p()
        nat.check(nat.lib().gnx_halo_plan_create(P, int(rank), int(n_local), arr(recv_pull), arr(recv_push), arr(send_pull), arr(send_push),
                                                 nat.ptr(pull_src) if pull_src is not None and pull_src.numel() else None,
                                                 push_graph.handle if push_graph is not None else None, byref(self.handle)))
        n_buf, local0, n_send, n_send_pull = c_int64(), c_int64(), c_int64(), c_int64()
        recv0, pull0, push0 = (c_int64 * P)(), (c_int64 * P)(), (c_int64 * P)()
        nat.check(nat.lib().gnx_halo_plan_layout(self.handle, byref(n_buf), byref(local0), byref(n_send), byref(n_send_pull), recv0, pull0, push0))
        self.n_buf, self.local_row0, self.n_send, self.n_send_pull = n_buf.value, local0.value, n_send.value, n_send_pull.value
        self.recv_row0, self.send_pull_row0, self.send_push_row0 = list(recv0), list(pull0), list(push0)

    def __del__(self):
        try:
            if self.handle:
                nat.lib().gnx_halo_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
