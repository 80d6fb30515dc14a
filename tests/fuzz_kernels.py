#!/usr/bin/env python3
"""Seeded fuzz of the kernels against float64 numpy / torch.

    python tests/fuzz_kernels.py [cases] [seed]              the long run (minutes; not part of the test suite)
    python tests/fuzz_kernels.py --dump CASE SEED OUT.npz    write one case's drawn inputs as a fixture (no GPU needed)

``draw_case`` makes every random draw of a case on the host (numpy only: replaying a seed needs no GPU), ``check_case`` runs the
kernels on it.  tests/test_gpu_fuzz.py runs a fixed 200-case slice under ``-m gpu`` and pins the one miss the long runs ever
produced (seed 303, case 1081; tests/golden/fuzz_seed303_case1081.npz)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd"), os.path.join(ROOT, "tests")]
import numpy as np

KINDS = ("spmm", "dropped", "kloop", "gcnii", "dense", "wgrad", "head", "edge")
DROP_SEED = 5                                    # seed of the counter RNG in the edge-dropout cases (streams = case number + k)


def draw_case(rng, case):
    """All random draws of case number ``case``.  The order of the draws is the seed's contract: a recorded (seed, case) pair names
    the same inputs as long as no draw is added -- round 5 added ``window`` (the last draw of a graph case), so pairs recorded in
    rounds 2-4 name other inputs now; the one miss those rounds produced is kept as a FIXTURE for that reason
    (tests/golden/fuzz_seed303_case1081.npz, dumped before the change)."""
    kind = case % 8
    s = dict(case=case, kind=kind)
    if kind in (0, 1, 2, 3):
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 6000, 33000]))        # 33000: the 128-entry long-row regime (2^15 ... 2^20 rows)
        sq = kind != 0 or rng.random() < 0.5
        n_cols = n if sq else int(rng.integers(1, 2000))
        nnz = int(rng.integers(0, 40 * n + 1))
        idx = np.stack([rng.integers(n, size=nnz), rng.integers(n_cols, size=nnz)], 1).astype(np.int64)
        if nnz > 1200 and rng.random() < 0.6:
            idx[: nnz // 2, 0] = int(rng.integers(n))                          # a hub row: the long-row kernels
        if nnz > 4000:
            for k in range(3):                                                   # rows of 130 ... 500 entries: long in one regime, short in the other
                m = int(rng.integers(130, 500))
                idx[nnz - (k + 1) * 500: nnz - (k + 1) * 500 + m, 0] = int(rng.integers(n))
        if kind in (1, 2, 3):
            idx = np.unique(idx, axis=0); nnz = len(idx)                          # fused dropout / K loop / GCNII: no duplicates
        vals = (rng.random(nnz) + 0.25).astype(np.float32)
        C = int(rng.choice([1, 3, 4, 8, 16, 17, 32, 40, 64, 100, 128, 200, 256, 320])) if kind != 3 else int(rng.choice([16, 32, 64, 48]))
        X = rng.standard_normal((n_cols, C)).astype(np.float32)
        H0 = rng.standard_normal((n, C)).astype(np.float32)
        s.update(n=n, sq=sq, n_cols=n_cols, nnz=nnz, idx=idx, vals=vals, C=C, X=X, H0=H0)
        if kind == 0:
            s["relu"] = rng.random() < 0.3
        elif kind == 1 and nnz:
            s["p"] = float(rng.choice([0.1, 0.5, 0.9]))
            s["K"] = int(rng.integers(2, 7))
        elif kind == 2 and nnz:
            s["K"] = int(rng.integers(1, 8))
        elif kind == 3 and nnz:
            s["M"] = (0.5 * np.eye(C) + rng.standard_normal((C, C)) * 0.2).astype(np.float32)
        s["window"] = int(rng.choice([0, 0, 64, 1000]))       # gnx_graph_set_row_window: another launch order, the same sums
    elif kind in (4, 5) and rng.random() < 0.5:
        # tall inputs in the shapes of the persistent kernels (k_dense_wreg / k_dense_ring / k_wgrad_acc), as aligned column slices of
        # wider matrices with a ragged number of rows; float64 on the device
        s["tall"] = True
        s["n"] = int(rng.integers(16384, 70000))
        s["F"] = int(rng.choice([32, 64, 100, 128, 192, 256, 260, 512])); s["O"] = int(rng.choice([4, 8, 16, 32, 40, 64, 100, 128, 132, 256]))
        s["padx"], s["pado"] = 4 * int(rng.integers(0, 9)), 4 * int(rng.integers(0, 9))
        s["gen_seed"] = int(rng.integers(1 << 30))
        if kind == 4:
            s["relu"], s["use_b"] = rng.random() < 0.5, rng.random() < 0.8
    elif kind in (4, 5):
        s["tall"] = False
        n = int(rng.choice([1, 15, 16, 17, 127, 129, 1000, 5000, 20000])); F = int(rng.integers(1, 700)); O = int(rng.integers(1, 300))
        s["X"], s["W"], s["b"] = (rng.standard_normal(shape).astype(np.float32) for shape in ((n, F), (F, O), (1, O)))
        s.update(n=n, F=F, O=O)
        if kind == 5:
            s["G"] = rng.standard_normal((n, O)).astype(np.float32)
    elif kind == 6:
        n, C, m = int(rng.integers(1, 5000)), int(rng.integers(1, 200)), int(rng.integers(1, 9000))
        s["L"] = (rng.standard_normal((n, C)) * 4).astype(np.float32)
        s["nodes"], s["labels"] = rng.integers(0, n, size=m), rng.integers(0, C, size=m)
    else:
        n, C, m = int(rng.integers(2, 5000)), int(rng.integers(1, 200)), int(rng.integers(1, 9000))
        s["F"] = rng.standard_normal((n, C)).astype(np.float32)
        s["e"] = rng.integers(0, n, size=(m, 2))
    return s


def replay(seed, case):
    """The inputs of (seed, case): the draws of every earlier case are made and thrown away."""
    rng = np.random.default_rng(seed)
    for c in range(case):
        draw_case(rng, c)
    return draw_case(rng, case)


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def rel_rows(x, y, floor_share=1e-2):
    """Per row: largest |x - y| relative to the row's largest |y|, but never to less than ``floor_share`` (1 %) of the matrix's
    largest element: a row whose terms cancel to ~1e-3 of their size (seed 303, case 1081: C = 1, |row| = 1.1e-3, terms ~ 1) carries
    the float32 noise of its terms.  The floor is for the BACKWARD comparison, where that cancellation was found; the forward
    comparison passes floor_share = 0 (the criterion before round 5: relative to the row's own largest element, floor 1e-3)."""
    return ((x - y).abs() / y.abs().max(dim=1, keepdim=True).values.clamp_min(max(1e-3, floor_share * float(y.abs().max())))).max(dim=1).values


PASSED_ONLY_WITH_THE_FLOOR = [0]        # backward comparisons of this process that the 1 % floor let pass (run() reports the count)


def training_loops(s):
    """The dropped-edge training loops of one ``kind == 1`` case, every way they can be computed.  Returns a dict of device
    tensors: forward chained / step by step, backward chained / step by step, plus the K adjacencies and degree scales."""
    import torch
    import gnntf
    from gnntf.sparse import _launch
    n, idx, vals, p, K, case = s["n"], s["idx"], s["vals"], s["p"], s["K"], s["case"]
    a_ = 0.1
    g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, s["n_cols"])), device="cuda:0")
    if s.get("window"):
        g.set_row_window(s["window"])
    D = gnntf.sparse.dropped_degree_scales(g, p, DROP_SEED, case, K)
    adjs = [gnntf.sparse.dropped_adjacency(g, p, DROP_SEED, case + k, D=D[k]) for k in range(K)]
    H0, up = dev(s["H0"]), dev(s["X"])
    with torch.no_grad():
        f_got = gnntf.sparse.ppr_loop(lambda k, bwd=False: adjs[k], H0, a_, K)
        f_want = H0
        for k in range(K):
            f_want = _launch(adjs[k], f_want, H0, 1.0 - a_, a_, 0)
    b_got = gnntf.sparse._backward_chained(adjs, up, a_)
    gk, b_want = up, up * a_
    for k in range(K - 1, -1, -1):
        gk = _launch(adjs[k], gk, None, 1.0 - a_, 0.0, 0, transposed=True)
        b_want = b_want + gk * (a_ if k >= 1 else 1.0)
    return dict(g=g, D=D, adjs=adjs, a=a_, f_got=f_got, f_want=f_want, b_got=b_got, b_want=b_want)


def backward_float64(s, a_=0.1):
    """dH0 of the K dropped iterations in float64 through the oracle's materialised dropped adjacencies (A_k^T products with
    scipy), and per row the sum of the absolute values of every term that enters it (the scale float32 rounding acts on)."""
    import scipy.sparse as sp
    from oracle import gnntf_oracle as orc
    n, idx, vals, p, K, case = s["n"], s["idx"], s["vals"], s["p"], s["K"], s["case"]
    ref_g = s["X"].astype(np.float64)
    mag_g = np.abs(ref_g)
    ref, mag = a_ * ref_g, a_ * mag_g
    for k in range(K - 1, -1, -1):
        ai, av = orc.get_adjacency(idx, vals, (n, n), graph_dropout=p, training=True, seed=DROP_SEED, stream=case + k, dtype=np.float64)
        A = sp.csr_matrix((av, (ai[:, 0], ai[:, 1])), shape=(n, n))
        ref_g = (1.0 - a_) * (A.T @ ref_g)
        mag_g = (1.0 - a_) * (abs(A).T @ mag_g)
        ref = ref + ref_g * (a_ if k >= 1 else 1.0)
        mag = mag + mag_g * (a_ if k >= 1 else 1.0)
    return ref, mag


def check_case(s):
    """Runs the kernels on one drawn case; returns the name of the statistic it counts for (None: an empty graph, nothing run)."""
    import torch
    import gnntf
    from gnntf.sparse import _launch, _dense_wgrad
    from oracle import gnntf_oracle as orc
    case, kind = s["case"], s["kind"]
    if kind in (0, 1, 2, 3):
        n, n_cols, nnz, idx, vals, C, X, H0, sq = (s[k] for k in ("n", "n_cols", "nnz", "idx", "vals", "C", "X", "H0", "sq"))
        g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, n_cols)), device="cuda:0")
        if s.get("window"):
            g.set_row_window(s["window"])
        longest = int(np.bincount(idx[:, 0], minlength=n).max()) if nnz else 1
        atol = 2e-4 + 1e-5 * np.sqrt(longest) + 1e-7 * longest                 # float32 sums over a hub row's entries cancel (seed 21, case 2280: 336K terms, |sum| = 110, off by 0.018)
        if kind == 0:
            relu = s["relu"]
            got = _launch(gnntf.Adjacency(g), dev(X), dev(H0), 0.8, 0.2, 1 if relu else 0).cpu().numpy()
            want = orc.sparse_dense_matmul(idx, vals.astype(np.float64), (n, n_cols), X.astype(np.float64)) * 0.8 + 0.2 * H0
            want = np.maximum(want, 0) if relu else want
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=f"spmm case {case}")
            if sq and nnz:
                gt = _launch(gnntf.Adjacency(g, dev(g.csr_arrays()[2].cpu().numpy())), dev(H0), None, 1.0, 0.0, 0, transposed=True).cpu().numpy()
                wt = orc.sparse_dense_matmul(idx[:, ::-1], vals.astype(np.float64), (n_cols, n), H0.astype(np.float64))
                np.testing.assert_allclose(gt, wt, rtol=1e-4, atol=2e-4 + 1e-5 * np.sqrt(int(np.bincount(idx[:, 1], minlength=n_cols).max())), err_msg=f"spmm_t case {case}")
            return "spmm"
        if kind == 1 and nnz:
            p, K = s["p"], s["K"]
            fused = gnntf.sparse.dropped_adjacency(g, p, DROP_SEED, case)
            two = gnntf.normalize(g, "symmetric", "none", dropout=p, seed=DROP_SEED, stream_id=case)
            for tr in (False, True):
                a_ = _launch(fused, dev(X), dev(H0), 0.9, 0.1, 0, transposed=tr)
                b_ = _launch(two, dev(X), dev(H0), 0.9, 0.1, 0, transposed=tr)
                assert torch.equal(a_, b_), f"dropped case {case} transposed={tr}: {float((a_ - b_).abs().max())}"
            # the training loops: column sums of K streams in one call (bitwise the one-stream sums), the chained forward and the
            # chained backward (running gradient sum + pre-scaled operand in the epilogue) against K un-chained launches
            r = training_loops(s)
            for k in range(K):
                assert torch.equal(r["D"][k], gnntf.sparse.dropped_degree_scales(r["g"], p, DROP_SEED, case + k, 1)[0]), f"scales case {case} stream {k}"
            tol = 2e-5 * (1.0 + np.sqrt(longest) / 10.0)
            e_f = float(rel_rows(r["f_got"], r["f_want"], floor_share=0.0).max())
            assert e_f < tol, f"chained forward case {case}: {e_f}"
            e_b = rel_rows(r["b_got"], r["b_want"])
            if float(e_b.max()) < tol <= float(rel_rows(r["b_got"], r["b_want"], floor_share=0.0).max()):
                PASSED_ONLY_WITH_THE_FLOOR[0] += 1
            if float(e_b.max()) >= tol:        # which of the two is off?  float64 through the materialised dropped adjacencies decides
                ref_t = dev(backward_float64(s)[0].astype(np.float32))
                e_chained, e_steps = float(rel_rows(r["b_got"], ref_t).max()), float(rel_rows(r["b_want"], ref_t).max())
                worst = int(e_b.argmax())
                raise AssertionError(f"chained backward case {case}: chained vs steps {float(e_b.max()):.3e} (tol {tol:.3e}); vs float64: chained "
                                     f"{e_chained:.3e}, steps {e_steps:.3e}; n={n} nnz={nnz} C={C} p={p} K={K} longest={longest} worst row {worst} "
                                     f"deg {int(np.bincount(idx[:, 0], minlength=n)[worst])} max|want| {float(r['b_want'][worst].abs().max()):.3e} "
                                     f"max D {float(r['D'].max()):.3e}")
            return "dropped"
        if kind == 2 and nnz:
            adj = gnntf.normalize(g, "symmetric")
            K = s["K"]
            H = dev(H0)
            for _ in range(K):
                H = gnntf.ppr_step(adj, H, dev(H0), 0.15)
            got = gnntf.appnp_propagate(adj, dev(H0), 0.15, K)
            if gnntf.sparse.friendly_width(C, n) == C:
                assert torch.equal(got, H), f"kloop case {case}"
            else:        # odd widths run the loop at a padded row width: other kernel variants, other summation grouping on hub rows
                spread = float(np.sqrt(max(np.bincount(idx[:, 0]).max(), 1)))
                assert torch.allclose(got, H, rtol=1e-5, atol=2e-6 * spread), f"kloop case {case}: {float((got - H).abs().max())}"
            # the same loop with relu in every iteration's epilogue (filter.py:22,28,35) against K single steps with the relu flag
            H = dev(H0)
            for _ in range(K):
                H = _launch(adj, H, dev(H0), 0.85, 0.15, 1)
            got = gnntf.appnp_propagate(adj, dev(H0), 0.15, K, relu=True)
            if gnntf.sparse.friendly_width(C, n) == C:
                assert torch.equal(got, H), f"relu kloop case {case}"
            else:
                spread = float(np.sqrt(max(np.bincount(idx[:, 0]).max(), 1)))
                assert torch.allclose(got, H, rtol=1e-5, atol=2e-6 * spread), f"relu kloop case {case}: {float((got - H).abs().max())}"
            return "kloop"
        if kind == 3 and nnz:
            adj = gnntf.normalize(g, "symmetric")
            M = s["M"]
            with torch.no_grad():
                got = gnntf.gcnii_step(adj, dev(X), dev(H0), 0.1, dev(M), relu=True).cpu().numpy()
            ai, av = orc.get_adjacency(idx, vals, (n, n), dtype=np.float64)
            want = np.maximum(orc.ppr_iteration(ai, av, (n, n), X.astype(np.float64), H0.astype(np.float64), 0.1) @ M.astype(np.float64), 0)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=f"gcnii case {case}")
            return "gcnii"
        return None
    if kind in (4, 5) and s["tall"]:
        n, F, O, padx, pado = s["n"], s["F"], s["O"], s["padx"], s["pado"]
        gen = torch.Generator(device="cuda").manual_seed(s["gen_seed"])
        wide = torch.randn(n, F + 2 * padx, device="cuda", generator=gen)
        X = wide[:, padx:padx + F]
        if kind == 4:
            W = torch.randn(F, O, device="cuda", generator=gen); b = torch.randn(1, O, device="cuda", generator=gen)
            relu, use_b = s["relu"], s["use_b"]
            got = gnntf.dense(X, W, b if use_b else None, relu=relu)
            want = X.double() @ W.double() + (b.double() if use_b else 0.0)
            want = torch.relu(want) if relu else want
            assert torch.allclose(got.double(), want, rtol=1e-4, atol=1e-4 * float(np.sqrt(F))), f"tall dense case {case}: n={n} F={F} O={O} pad={padx}"
            return "dense"
        gw = torch.randn(n, O + 2 * pado, device="cuda", generator=gen)
        G = gw[:, pado:pado + O]
        got = _dense_wgrad(X, G)
        want = sum(X[i:i + 65536].double().t() @ G[i:i + 65536].double() for i in range(0, n, 65536))
        assert torch.allclose(got.double(), want, rtol=1e-4, atol=2e-4 * float(np.sqrt(n))), f"tall wgrad case {case}: n={n} F={F} O={O} pads={padx},{pado}"
        assert torch.equal(got, _dense_wgrad(X, G)), f"tall wgrad case {case} not repeatable"
        return "wgrad"
    if kind in (4, 5):
        n, F, X, W, b = s["n"], s["F"], s["X"], s["W"], s["b"]
        if kind == 4:
            got = gnntf.dense(dev(X), dev(W), dev(b), relu=True).cpu().numpy()
            np.testing.assert_allclose(got, np.maximum(X.astype(np.float64) @ W + b, 0), rtol=1e-4, atol=1e-4 * np.sqrt(F), err_msg=f"dense case {case}")
            return "dense"
        G = s["G"]
        got = _dense_wgrad(dev(X), dev(G)).cpu().numpy()
        np.testing.assert_allclose(got, X.astype(np.float64).T @ G, rtol=1e-4, atol=2e-4 * np.sqrt(n), err_msg=f"wgrad case {case}")
        return "wgrad"
    if kind == 6:
        L, nodes, labels = s["L"], s["nodes"], s["labels"]
        got = float(gnntf.node_ce(dev(L), nodes, labels))
        want = orc.node_loss(L.astype(np.float64), nodes, labels)
        assert abs(got - want) <= 2e-5 * max(abs(want), 1), f"head case {case}: {got} {want}"
        assert np.array_equal(gnntf.node_argmax(dev(L), nodes).cpu().numpy(), L[nodes].argmax(1))
        return "head"
    F, e = s["F"], s["e"]
    np.testing.assert_allclose(gnntf.edge_scores(dev(F), e).cpu().numpy(), orc.link_logits(F.astype(np.float64), e), rtol=1e-4, atol=1e-4, err_msg=f"edge case {case}")
    return "edge"


def run(cases, seed, first=0, verbose=True):
    """Cases ``first`` ... ``cases - 1`` of ``seed`` (earlier ones are drawn and skipped).  Returns the per-kind counts."""
    import torch
    import gnntf
    rng = np.random.default_rng(seed)
    gnntf.set_default_device("cuda:0")
    t0 = time.time()
    stats = dict.fromkeys(KINDS, 0)
    for case in range(cases):
        s = draw_case(rng, case)
        if case < first:
            continue
        kind = check_case(s)
        if kind is not None:
            stats[kind] += 1
        if verbose and case % 50 == 49:
            print(f"{case + 1} cases, {time.time() - t0:.0f} s", stats, flush=True)
    torch.cuda.synchronize()
    if verbose or PASSED_ONLY_WITH_THE_FLOOR[0]:
        print(f"backward comparisons that pass only with the 1 % floor of rel_rows: {PASSED_ONLY_WITH_THE_FLOOR[0]} of {stats['dropped']}", flush=True)
    return stats


def dump(case, seed, out):
    s = replay(seed, case)
    arrays = {k: v for k, v in s.items() if isinstance(v, np.ndarray)}
    scalars = {k: np.asarray(v) for k, v in s.items() if not isinstance(v, np.ndarray)}
    np.savez_compressed(out, seed=np.asarray(seed), **arrays, **scalars)
    print({k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in s.items()})


def load(path):
    z = np.load(path)
    s = {k: z[k] for k in z.files}
    return {k: (v if v.ndim else v.item()) for k, v in s.items()}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--dump":
        dump(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        print("FUZZ OK", cases, "cases, seed", seed, run(cases, seed))
