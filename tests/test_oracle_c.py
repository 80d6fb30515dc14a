"""The C port of the oracle (oracle/propagate_ref.c, used as bench.py's cpu_baseline) agrees
with the numpy restatement."""
import ctypes

import numpy as np
import pytest

import graphs
from oracle import gnntf_oracle as orc


@pytest.fixture(scope="module")
def cport():
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_oracle())
    lib.oracle_appnp_propagate.restype = ctypes.c_int
    lib.oracle_appnp_propagate.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int64,
                                                                                     ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("K,C", [(1, 7), (10, 16), (3, 64)])
def test_c_port_matches_numpy_oracle(cport, K, C):
    coo, vals, shape = graphs.rmat_symmetric_coo(500, 4000, seed=K)
    rowptr, colidx, cvals = orc.coo_to_csr_coalesced(coo, vals, shape)
    H0 = np.random.default_rng(C).uniform(-1, 1, size=(500, C)).astype(np.float32)
    out = np.empty_like(H0)
    work = np.empty_like(H0)
    rc = cport.oracle_appnp_propagate(500, rowptr.ctypes.data, colidx.ctypes.data, cvals.ctypes.data, H0.ctypes.data,
                                      0.1, K, C, out.ctypes.data, work.ctypes.data)
    assert rc == 0
    want = orc.appnp_propagate(coo, vals, shape, H0, a=0.1, iterations=K)
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)
    assert cport.oracle_num_threads() >= 1


def test_c_port_sample_iteration_parallel_normalisation(cport):
    """bench.py's cpu_baseline entry: one iteration on a row prefix, normalisation spread over the threads (atomic column
    sums: same values up to float summation order) or skipped (values taken as A_hat's)."""
    coo, vals, shape = graphs.rmat_symmetric_coo(800, 7000, seed=9)
    vals = (vals * np.random.default_rng(0).uniform(0.5, 2.0, size=len(vals))).astype(np.float32)
    rowptr, colidx, cvals = orc.coo_to_csr_coalesced(coo, vals, shape)
    C, rows = 12, 300
    H = np.random.default_rng(1).uniform(-1, 1, size=(800, C)).astype(np.float32)
    H0 = np.random.default_rng(2).uniform(-1, 1, size=(800, C)).astype(np.float32)
    fn = cport.oracle_sample_iteration_par
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    out = np.empty((rows, C), dtype=np.float32)
    assert fn(800, rows, rowptr.ctypes.data, colidx.ctypes.data, cvals.ctypes.data, H.ctypes.data, H0.ctypes.data, 0.1, C, out.ctypes.data, 1) == 0
    ai, av = orc.get_adjacency(coo, vals, shape)
    want = orc.ppr_iteration(ai, av, shape, H, H0, 0.1)
    np.testing.assert_allclose(out, want[:rows], rtol=1e-5, atol=1e-6)
    assert fn(800, rows, rowptr.ctypes.data, colidx.ctypes.data, cvals.ctypes.data, H.ctypes.data, H0.ctypes.data, 0.1, C, out.ctypes.data, 0) == 0
    want_raw = orc.ppr_iteration(coo, vals, shape, H, H0, 0.1)
    np.testing.assert_allclose(out, want_raw[:rows], rtol=1e-5, atol=1e-6)


def test_scipy_and_torch_restatements_match_numpy_oracle():
    """bench.py's other two CPU baselines (oracle/cpu_baselines.py): the one-thread scipy iteration with its own
    renormalisation (whole matrix: exact; row prefix with the true column sums: the same rows) and the torch.sparse.mm
    iteration over pre-normalised values."""
    from oracle import cpu_baselines as cb
    coo, vals, shape = graphs.rmat_symmetric_coo(700, 6000, seed=4)
    vals = (vals * np.random.default_rng(3).uniform(0.5, 2.0, size=len(vals))).astype(np.float32)
    sym = {}
    for (i, j), v in zip(coo.tolist(), vals.tolist()):               # keep the values symmetric (the generator's pattern is)
        sym[(min(i, j), max(i, j))] = v
    vals = np.array([sym[(min(i, j), max(i, j))] for i, j in coo.tolist()], dtype=np.float32)
    rowptr, colidx, cvals = orc.coo_to_csr_coalesced(coo, vals, shape)
    C = 9
    H = np.random.default_rng(1).uniform(-1, 1, size=(700, C)).astype(np.float32)
    H0 = np.random.default_rng(2).uniform(-1, 1, size=(700, C)).astype(np.float32)
    ai, av = orc.get_adjacency(coo, vals, shape)
    want = orc.ppr_iteration(ai, av, shape, H, H0, 0.1)
    out, t_norm, t_spmm = cb.scipy_iteration(rowptr, colidx, cvals, H, H0, 0.1)
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-6)
    assert t_norm > 0 and t_spmm > 0
    colsum = np.bincount(colidx, weights=cvals, minlength=700).astype(np.float32)
    part, _, _ = cb.scipy_iteration(rowptr, colidx, cvals, H, H0, 0.1, rows=250, colsum=colsum)
    np.testing.assert_allclose(part, want[:250], rtol=1e-5, atol=1e-6)
    _, _, nvals = orc.coo_to_csr_coalesced(ai, av, shape)
    got, seconds, threads = cb.torch_sparse_iteration(rowptr, colidx, nvals, H, H0, 0.1, rows=300, threads=2)
    np.testing.assert_allclose(got, want[:300], rtol=1e-5, atol=1e-6)
    assert seconds > 0 and threads == 2
    assert np.array_equal(cb.divide_no_nan(np.float32(1), np.array([0, 4], dtype=np.float32)), np.array([0, 0.25], dtype=np.float32))
