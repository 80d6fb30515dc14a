"""ctypes binding of libgnx.so (include/gnx.h) -- the only way the package reaches the GPU.

There is deliberately NO fallback: if the library is missing or a call fails, an Exception
is raised.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 600          # include/gnx.h GNX_ABI_VERSION: the header this binding was written against
# GNX_LIBRARY: another build of the same library (the tuning build of tools/, `make TUNING=1` -> lib/tune/libgnx.so)
LIB_PATH = os.environ.get("GNX_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libgnx.so")

NORM = {"none": 0, "symmetric": 1, "bipartite": 2}
EYE = {"none": 0, "before": 1, "after": 2}
ACT_NONE, ACT_RELU, ACT_SKIP_EMPTY = 0, 1, 256
HALO_ALL, HALO_PULL, HALO_PUSH = 0, 1, 2
RESERVE_TRANSPOSED, RESERVE_K_LOOP = 1, 2

# name -> (restype, argtypes); must list every symbol include/gnx.h declares
SIGNATURES = {
    "gnx_last_error": (c_char_p, []),
    "gnx_version": (c_int, []),
    "gnx_graph_create_coo": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "gnx_graph_create_csr": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "gnx_graph_destroy": (c_int, [c_void_p]),
    "gnx_graph_info": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "gnx_graph_csr": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "gnx_graph_export": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnx_graph_normalize": (c_int, [c_void_p, c_int, c_int, c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "gnx_graph_normalize_t": (c_int, [c_void_p, c_int, c_int, c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "gnx_graph_reserve": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "gnx_graph_set_row_window": (c_int, [c_void_p, c_int64, c_void_p]),
    "gnx_graph_set_dropout_counter": (c_int, [c_void_p, c_void_p]),
    "gnx_graph_set_block": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gnx_graph_colsum": (c_int, [c_void_p, c_float, c_uint64, c_uint64, c_void_p, c_void_p]),
    "gnx_graph_colsum_streams": (c_int, [c_void_p, c_float, c_uint64, c_uint64, c_int, c_void_p, c_void_p]),
    "gnx_degree_scale": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p]),
    "gnx_graph_scale_values": (c_int, [c_void_p, c_float, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnx_spmm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float,
                         c_int, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_t": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float,
                           c_int, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float,
                                 c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_dropped": (c_int, [c_void_p, c_void_p, c_float, c_uint64, c_uint64, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                 c_float, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_dropped_chained": (c_int, [c_void_p, c_void_p, c_float, c_uint64, c_uint64, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                         c_int64, c_float, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_dropped_back": (c_int, [c_void_p, c_void_p, c_float, c_uint64, c_uint64, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                      c_int64, c_float, c_float, c_void_p, c_int64, c_float, c_void_p, c_int64, c_int, c_void_p]),
    "gnx_spmm_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float, c_int,
                              c_void_p, c_void_p, c_int64, c_void_p]),
    "gnx_spmm_tv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_float, c_float,
                            c_int, c_void_p, c_int64, c_void_p]),
    "gnx_graph_permute_values_t": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnx_ppr_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p, c_void_p]),
    "gnx_appnp_propagate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int64, c_void_p, c_void_p,
                                    c_void_p]),
    "gnx_appnp_propagate_act": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int64, c_int, c_void_p, c_void_p,
                                        c_void_p]),
    "gnx_halo_plan_create": (c_int, [c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "gnx_halo_plan_destroy": (c_int, [c_void_p]),
    "gnx_halo_plan_layout": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_void_p, c_void_p,
                                     c_void_p]),
    "gnx_halo_pack": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gnx_halo_exchange": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gnx_halo_bind_rccl": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "gnx_gather_rows32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gnx_gather_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gnx_gcnii_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p, c_int64, c_int, c_void_p,
                               c_void_p, c_void_p]),
    "gnx_dense": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64,
                          c_void_p]),
    "gnx_dense_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "gnx_node_ce": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gnx_node_ce_backward": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                     c_void_p]),
    "gnx_node_argmax": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "gnx_edge_scores": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gnx_edge_scores_backward": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_void_p]),
    "gnx_linear_combination": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "gnx_stream_read": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "gnx_stream_copy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "gnx_probe_block_xcd": (c_int, [c_int64, c_void_p, c_void_p]),
    "gnx_graph_last_kernel": (c_char_p, [c_void_p]),
}

_lib = None


def lib():
    """Loads libgnx.so once; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Exception(f"gnntf: the HIP library {LIB_PATH} is missing -- build it with "
                            f"`make -C gnn-tf_amd/csrc` (or __graft_entry__.build()); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        # the argument lists of several entries changed under the same names at 0.3 (include/gnx.h, GNX_ABI_VERSION): a library
        # of another minor version would be called with shifted arguments -- refused, not tried
        version = int(handle.gnx_version())
        if version // 100 != ABI_VERSION // 100:
            raise Exception(f"gnntf: {LIB_PATH} implements ABI {version}, this package binds ABI {ABI_VERSION}: rebuild the library "
                            f"(make -C gnn-tf_amd/csrc)")
        _lib = handle
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().gnx_last_error()
        raise Exception((msg.decode() if msg else "") or f"libgnx error {rc}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream():
    """The raw hipStream_t of torch's current stream on the current device (the cheap accessor: this sits on every launch)."""
    import torch
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


class on_device:
    """``with on_device(t.device):`` -- makes the tensor's device current for the launch; free when it already is
    (the usual case: one process per GPU)."""

    __slots__ = ("index", "prev")

    def __init__(self, device):
        self.index = device.index

    def __enter__(self):
        import torch
        self.prev = torch.cuda.current_device()
        if self.index is None:
            self.index = self.prev
        if self.prev != self.index:
            torch.cuda.set_device(self.index)

    def __exit__(self, *exc):
        if self.prev != self.index:
            import torch
            torch.cuda.set_device(self.prev)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise Exception("gnntf: the propagation path runs on the GPU only (tensor on %s); there is no CPU "
                            "fallback" % t.device)
