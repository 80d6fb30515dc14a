"""Seeded synthetic graphs shared by the tests, the golden-fixture generator and bench.py."""
import numpy as np


def undirected_edges(n, m, rng):
    """m distinct undirected edges without self loops, as an int64 [m, 2] array (u < v)."""
    seen = set()
    out = []
    while len(out) < m:
        u, v = int(rng.integers(n)), int(rng.integers(n))
        if u == v:
            continue
        key = (min(u, v), max(u, v))
        if key in seen:
            continue
        seen.add(key)
        out.append(key)
    return np.asarray(out, dtype=np.int64)


def cora_shaped(seed=0, n=2708, m=5278, f=1433, density=0.0127):
    """Cora-sized stand-in: DGL lists both directions of every edge and
    graph2adj appends the reverse again, so the COO holds every entry twice (21112 entries,
    10556 unique); features are binary, ~1.27 % dense, row-normalised."""
    rng = np.random.default_rng(seed)
    und = undirected_edges(n, m, rng)
    both = np.concatenate([und, und[:, ::-1]])           # the DiGraph's edge list
    both = both[rng.permutation(len(both))]
    coo = np.concatenate([both, both[:, ::-1]])          # graph2adj(directed=False) appends the reverse
    vals = np.ones(len(coo), dtype=np.float32)
    X = (rng.random((n, f)) < density).astype(np.float32)
    X[np.arange(n), rng.integers(f, size=n)] = 1.0       # no empty rows
    X = X / X.sum(axis=1, keepdims=True)
    return coo, vals, (n, n), X.astype(np.float32)


def rmat_edges(scale, m, rng, a=0.57, b=0.19, c=0.19):
    """m directed R-MAT pairs over 2**scale vertices (SURVEY.md section 8(d) parameters)."""
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(m)
        src = src * 2 + (r >= a + b)
        dst = dst * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c))
    return src, dst


def rmat_symmetric_coo(n, m_directed, seed, scale=None):
    """Relabelled, self-loop-free, symmetrised, coalesced R-MAT graph on n vertices, with a
    random vertex permutation so locality is not an artefact of the generator."""
    rng = np.random.default_rng(seed)
    scale = scale or max(1, int(np.ceil(np.log2(n))))
    s, d = rmat_edges(scale, m_directed, rng)
    s, d = s % n, d % n
    keep = s != d
    s, d = s[keep], d[keep]
    perm = rng.permutation(n)
    s, d = perm[s], perm[d]
    key = np.unique(np.concatenate([s * n + d, d * n + s]))
    coo = np.stack([key // n, key % n], axis=1).astype(np.int64)
    return coo, np.ones(len(coo), dtype=np.float32), (n, n)


def random_coo(n_rows, n_cols, nnz, seed, weighted=True, dup_frac=0.2):
    """Unsorted COO with a share of duplicated entries."""
    rng = np.random.default_rng(seed)
    base = max(1, int(nnz * (1 - dup_frac)))
    idx = np.stack([rng.integers(n_rows, size=base), rng.integers(n_cols, size=base)], axis=1)
    extra = idx[rng.integers(base, size=nnz - base)] if nnz > base else idx[:0]
    idx = np.concatenate([idx, extra])
    idx = idx[rng.permutation(len(idx))].astype(np.int64)
    vals = (rng.random(len(idx)).astype(np.float32) + 0.25) if weighted else np.ones(len(idx), dtype=np.float32)
    return idx, vals, (n_rows, n_cols)
