"""The C-ABI library loads without a GPU and exports every symbol include/gnx.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnx_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("gnx_graph_create_coo", "gnx_graph_normalize", "gnx_spmm", "gnx_spmm_t", "gnx_ppr_step",
                 "gnx_appnp_propagate", "gnx_gather_rows", "gnx_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from gnntf import _native
    assert os.path.exists(_native.LIB_PATH), "libgnx.so missing: run __graft_entry__.build()"
    handle = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in gnx.h but not exported"
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_version_and_error_string():
    from gnntf import _native
    lib = _native.lib()
    assert lib.gnx_version() >= 100
    assert isinstance(lib.gnx_last_error(), bytes)


def test_argument_errors_do_not_need_a_gpu():
    """NULL-handle / bad-argument paths return an error code and a message (no compute)."""
    from gnntf import _native
    lib = _native.lib()
    assert lib.gnx_graph_info(None, None, None, None, None) == -1
    assert b"NULL handle" in lib.gnx_last_error()
    out = ctypes.c_void_p()
    assert lib.gnx_graph_create_coo(-1, 4, 0, None, None, None, ctypes.byref(out)) == -1
    assert b"negative" in lib.gnx_last_error()
    assert lib.gnx_degree_scale(None, 0, 7, 0, None) == -1
    assert lib.gnx_last_error() == b"Invalid matrix normalization"
    with pytest.raises(Exception, match="Invalid matrix normalization"):
        _native.check(-1)


def test_no_fallback_on_cpu():
    """The product path must fail loudly, never route through a CPU implementation."""
    import torch
    import gnntf
    coo = gnntf.SparseCOO([[0, 1], [1, 0]], [1.0, 1.0], (2, 2))
    with pytest.raises(Exception, match="GPU only"):
        gnntf.DeviceGraph(coo, device="cpu")
    with pytest.raises(Exception, match="no CPU fallback"):
        gnntf.gather_rows(torch.zeros(4, 4), torch.tensor([0, 1]))
    src = "".join(open(os.path.join(ROOT, "gnn-tf_amd", "gnntf", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "gnn-tf_amd", "gnntf")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src
