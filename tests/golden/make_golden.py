"""Generates the committed golden fixtures from the CPU oracle (oracle/gnntf_oracle.py).

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference ships no vectors and cannot run here (no TensorFlow), so
these are outputs of the restatement, not of gnntf itself.  They freeze the oracle against
regressions and carry identical inputs/expected outputs to the GPU box.  All inputs are
rounded to float16-representable values so they can be stored compactly and exactly.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from oracle import gnntf_oracle as orc  # noqa: E402
import graphs  # noqa: E402


def f16_exact(x):
    return np.asarray(x, dtype=np.float16)


def small_init(rng, fan_in, fan_out):
    """variables.py:32-34 'small': U(+-1/sqrt(fan_out))."""
    bound = 1.0 / np.sqrt(fan_out)
    return f16_exact(rng.uniform(-bound, bound, size=(fan_in, fan_out)))


def cora_appnp():
    coo, vals, shape, X = graphs.cora_shaped(seed=0)
    rng = np.random.default_rng(1)
    n, f = X.shape
    xr, xc = np.nonzero(X)
    W1, b1 = small_init(rng, f, 64), f16_exact(rng.uniform(-0.1, 0.1, size=(1, 64)))
    W2, b2 = small_init(rng, 64, 7), f16_exact(rng.uniform(-0.1, 0.1, size=(1, 7)))
    weights = [(W1.astype(np.float32), b1.astype(np.float32)), (W2.astype(np.float32), b2.astype(np.float32))]
    logits32, H0 = orc.appnp_forward_eval(coo, vals, shape, X, weights, a=0.1, iterations=10, dtype=np.float32)
    logits64, _ = orc.appnp_forward_eval(coo, vals, shape, X, weights, a=0.1, iterations=10, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "cora_shaped_appnp.npz"),
                        coo=coo.astype(np.int32), n=n, f=f, x_rows=xr.astype(np.int32), x_cols=xc.astype(np.int16),
                        W1=W1, b1=b1, W2=W2, b2=b2, a=0.1, iterations=10,
                        H0=H0, logits32=logits32, logits64=logits64, argmax=np.argmax(logits64, axis=1).astype(np.int8))


def arxiv_mini_gcn():
    n, f, hidden, classes = 2048, 128, 64, 40
    coo, vals, shape = graphs.rmat_symmetric_coo(n, 14000, seed=1)
    rng = np.random.default_rng(2)
    X = f16_exact(rng.standard_normal((n, f)))
    W1, b1 = small_init(rng, f, hidden), f16_exact(rng.uniform(-0.1, 0.1, size=(1, hidden)))
    W2, b2 = small_init(rng, hidden, classes), f16_exact(rng.uniform(-0.1, 0.1, size=(1, classes)))
    weights = [(W1.astype(np.float32), b1.astype(np.float32)), (W2.astype(np.float32), b2.astype(np.float32))]
    out32 = orc.gcn_forward_eval(coo, vals, shape, X.astype(np.float32), weights, dtype=np.float32)
    out64 = orc.gcn_forward_eval(coo, vals, shape, X.astype(np.float64), weights, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "arxiv_mini_gcn.npz"), coo=coo.astype(np.int32), n=n, X=X, W1=W1, b1=b1, W2=W2,
                        b2=b2, out32=out32, out64=out64.astype(np.float32))


def dropout_masks():
    """Training mode: edge-dropout masks of the counter RNG + the re-normalised values, on a
    small weighted graph with duplicate entries, for two streams."""
    coo, vals, shape = graphs.random_coo(96, 96, 700, seed=3, weighted=True, dup_frac=0.3)
    vals = f16_exact(vals).astype(np.float32)
    out = dict(coo=coo.astype(np.int32), vals=vals, n=96, p=0.5, seed=1234)
    for stream in (0, 7):
        keep = orc.keep_mask(coo, 0.5, 1234, stream)
        ai, av = orc.get_adjacency(coo, vals, shape, graph_dropout=0.5, training=True, seed=1234, stream=stream)
        out[f"keep_{stream}"] = keep
        out[f"adj_vals_{stream}"] = av
    np.savez_compressed(os.path.join(HERE, "dropout_masks.npz"), **out)


if __name__ == "__main__":
    cora_appnp()
    arxiv_mini_gcn()
    dropout_masks()
    for name in sorted(os.listdir(HERE)):
        if name.endswith(".npz"):
            print(name, os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB")
