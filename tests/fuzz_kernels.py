#!/usr/bin/env python3
"""Long seeded fuzz of the kernels against float64 numpy / torch (not part of the test suite: minutes, not seconds).
    python tests/fuzz_kernels.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import gnntf
from gnntf.sparse import _launch, _dense_wgrad
from oracle import gnntf_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
gnntf.set_default_device("cuda:0")
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
t0 = time.time()
stats = dict(spmm=0, dropped=0, kloop=0, gcnii=0, dense=0, wgrad=0, head=0, edge=0)
for case in range(cases):
    kind = case % 8
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
        g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, n_cols)), device="cuda:0")
        X = rng.standard_normal((n_cols, C)).astype(np.float32)
        H0 = rng.standard_normal((n, C)).astype(np.float32)
        longest = int(np.bincount(idx[:, 0], minlength=n).max()) if nnz else 1
        atol = 2e-4 + 1e-5 * np.sqrt(longest) + 1e-7 * longest                 # float32 sums over a hub row's entries cancel (seed 21, case 2280: 336K terms, |sum| = 110, off by 0.018)
        if kind == 0:
            relu = rng.random() < 0.3
            got = _launch(gnntf.Adjacency(g), dev(X), dev(H0), 0.8, 0.2, 1 if relu else 0).cpu().numpy()
            want = orc.sparse_dense_matmul(idx, vals.astype(np.float64), (n, n_cols), X.astype(np.float64)) * 0.8 + 0.2 * H0
            want = np.maximum(want, 0) if relu else want
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=f"spmm case {case}")
            if sq and nnz:
                gt = _launch(gnntf.Adjacency(g, dev(vals_sorted := g.csr_arrays()[2].cpu().numpy())), dev(H0), None, 1.0, 0.0, 0, transposed=True).cpu().numpy()
                wt = orc.sparse_dense_matmul(idx[:, ::-1], vals.astype(np.float64), (n_cols, n), H0.astype(np.float64))
                np.testing.assert_allclose(gt, wt, rtol=1e-4, atol=2e-4 + 1e-5 * np.sqrt(int(np.bincount(idx[:, 1], minlength=n_cols).max())), err_msg=f"spmm_t case {case}")
            stats["spmm"] += 1
        elif kind == 1 and nnz:
            p = float(rng.choice([0.1, 0.5, 0.9]))
            fused = gnntf.sparse.dropped_adjacency(g, p, 5, case)
            two = gnntf.normalize(g, "symmetric", "none", dropout=p, seed=5, stream_id=case)
            for tr in (False, True):
                a_ = _launch(fused, dev(X), dev(H0), 0.9, 0.1, 0, transposed=tr)
                b_ = _launch(two, dev(X), dev(H0), 0.9, 0.1, 0, transposed=tr)
                assert torch.equal(a_, b_), f"dropped case {case} transposed={tr}: {float((a_ - b_).abs().max())}"
            # the training loops: column sums of K streams in one call (bitwise the one-stream sums), the chained forward and the
            # chained backward (running gradient sum + pre-scaled operand in the epilogue) against K un-chained launches
            K, a_ = int(rng.integers(2, 7)), 0.1
            D = gnntf.sparse.dropped_degree_scales(g, p, 5, case, K)
            for k in range(K):
                assert torch.equal(D[k], gnntf.sparse.dropped_degree_scales(g, p, 5, case + k, 1)[0]), f"scales case {case} stream {k}"
            adjs = [gnntf.sparse.dropped_adjacency(g, p, 5, case + k, D=D[k]) for k in range(K)]
            tol = 2e-5 * (1.0 + np.sqrt(longest) / 10.0)
            # error relative to the row's largest element, but never to less than 1 % of the matrix's largest: a row whose terms
            # cancel to ~1e-3 of their size (seed 303, case 1081: C = 1, |row| = 1.1e-3, terms ~ 1) carries the float32 noise of its terms
            rel = lambda x, y: float(((x - y).abs() / y.abs().max(dim=1, keepdim=True).values.clamp_min(max(1e-3, 1e-2 * float(y.abs().max())))).max())
            with torch.no_grad():
                f_got = gnntf.sparse.ppr_loop(lambda k, bwd=False: adjs[k], dev(H0), a_, K)
                f_want = dev(H0)
                for k in range(K):
                    f_want = _launch(adjs[k], f_want, dev(H0), 1.0 - a_, a_, 0)
            assert rel(f_got, f_want) < tol, f"chained forward case {case}: {rel(f_got, f_want)}"
            up = dev(X)
            b_got = gnntf.sparse._backward_chained(adjs, up, a_)
            gk, b_want = up, up * a_
            for k in range(K - 1, -1, -1):
                gk = _launch(adjs[k], gk, None, 1.0 - a_, 0.0, 0, transposed=True)
                b_want = b_want + gk * (a_ if k >= 1 else 1.0)
            if rel(b_got, b_want) >= tol:      # which of the two is off?  float64 through the materialised dropped adjacencies decides
                import scipy.sparse as sp
                ref_g = X.astype(np.float64)
                ref = a_ * ref_g
                for k in range(K - 1, -1, -1):
                    ai, av = orc.get_adjacency(idx, vals, (n, n), graph_dropout=p, training=True, seed=5, stream=case + k, dtype=np.float64)
                    A = sp.csr_matrix((av, (ai[:, 0], ai[:, 1])), shape=(n, n))
                    ref_g = (1.0 - a_) * (A.T @ ref_g)
                    ref = ref + ref_g * (a_ if k >= 1 else 1.0)
                ref_t = dev(ref.astype(np.float32))
                e_chained, e_steps = rel(b_got, ref_t), rel(b_want, ref_t)
                worst = int(((b_got - b_want).abs() / b_want.abs().max(dim=1, keepdim=True).values.clamp_min(max(1e-3, 1e-2 * float(b_want.abs().max())))).max(dim=1).values.argmax())
                raise AssertionError(f"chained backward case {case}: chained vs steps {rel(b_got, b_want):.3e} (tol {tol:.3e}); vs float64: chained "
                                     f"{e_chained:.3e}, steps {e_steps:.3e}; n={n} nnz={nnz} C={C} p={p} K={K} longest={longest} worst row {worst} "
                                     f"deg {int(np.bincount(idx[:, 0], minlength=n)[worst])} max|want| {float(b_want[worst].abs().max()):.3e} "
                                     f"max D {float(D.max()):.3e}")
            stats["dropped"] += 1
        elif kind == 2 and nnz:
            adj = gnntf.normalize(g, "symmetric")
            K = int(rng.integers(1, 8))
            H = dev(H0)
            for _ in range(K):
                H = gnntf.ppr_step(adj, H, dev(H0), 0.15)
            got = gnntf.appnp_propagate(adj, dev(H0), 0.15, K)
            if gnntf.sparse.friendly_width(C, n) == C:
                assert torch.equal(got, H), f"kloop case {case}"
            else:        # odd widths run the loop at a padded row width: other kernel variants, other summation grouping on hub rows
                longest = float(np.sqrt(max(np.bincount(idx[:, 0]).max(), 1)))
                assert torch.allclose(got, H, rtol=1e-5, atol=2e-6 * longest), f"kloop case {case}: {float((got - H).abs().max())}"
            stats["kloop"] += 1
        elif kind == 3 and nnz:
            adj = gnntf.normalize(g, "symmetric")
            M = (0.5 * np.eye(C) + rng.standard_normal((C, C)) * 0.2).astype(np.float32)
            with torch.no_grad():
                got = gnntf.gcnii_step(adj, dev(X), dev(H0), 0.1, dev(M), relu=True).cpu().numpy()
            ai, av = orc.get_adjacency(idx, vals, (n, n), dtype=np.float64)
            want = np.maximum(orc.ppr_iteration(ai, av, (n, n), X.astype(np.float64), H0.astype(np.float64), 0.1) @ M.astype(np.float64), 0)
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=f"gcnii case {case}")
            stats["gcnii"] += 1
        del g
    elif kind in (4, 5) and rng.random() < 0.5:
        # tall inputs in the shapes of the persistent kernels (k_dense_wreg / k_dense_ring / k_wgrad_acc), as aligned column slices of
        # wider matrices with a ragged number of rows; float64 on the device
        n = int(rng.integers(16384, 70000))
        F = int(rng.choice([32, 64, 100, 128, 192, 256, 260, 512])); O = int(rng.choice([4, 8, 16, 32, 40, 64, 100, 128, 132, 256]))
        padx, pado = 4 * int(rng.integers(0, 9)), 4 * int(rng.integers(0, 9))
        gen = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
        wide = torch.randn(n, F + 2 * padx, device="cuda", generator=gen)
        X = wide[:, padx:padx + F]
        if kind == 4:
            W = torch.randn(F, O, device="cuda", generator=gen); b = torch.randn(1, O, device="cuda", generator=gen)
            relu, use_b = rng.random() < 0.5, rng.random() < 0.8
            got = gnntf.dense(X, W, b if use_b else None, relu=relu)
            want = X.double() @ W.double() + (b.double() if use_b else 0.0)
            want = torch.relu(want) if relu else want
            stats["dense"] += 1
            assert torch.allclose(got.double(), want, rtol=1e-4, atol=1e-4 * float(np.sqrt(F))), f"tall dense case {case}: n={n} F={F} O={O} pad={padx}"
        else:
            gw = torch.randn(n, O + 2 * pado, device="cuda", generator=gen)
            G = gw[:, pado:pado + O]
            got = _dense_wgrad(X, G)
            want = sum(X[i:i + 65536].double().t() @ G[i:i + 65536].double() for i in range(0, n, 65536))
            assert torch.allclose(got.double(), want, rtol=1e-4, atol=2e-4 * float(np.sqrt(n))), f"tall wgrad case {case}: n={n} F={F} O={O} pads={padx},{pado}"
            assert torch.equal(got, _dense_wgrad(X, G)), f"tall wgrad case {case} not repeatable"
            stats["wgrad"] += 1
    elif kind in (4, 5):
        n = int(rng.choice([1, 15, 16, 17, 127, 129, 1000, 5000, 20000])); F = int(rng.integers(1, 700)); O = int(rng.integers(1, 300))
        X, W, b = (rng.standard_normal(s).astype(np.float32) for s in ((n, F), (F, O), (1, O)))
        if kind == 4:
            got = gnntf.dense(dev(X), dev(W), dev(b), relu=True).cpu().numpy()
            np.testing.assert_allclose(got, np.maximum(X.astype(np.float64) @ W + b, 0), rtol=1e-4, atol=1e-4 * np.sqrt(F), err_msg=f"dense case {case}")
            stats["dense"] += 1
        else:
            G = rng.standard_normal((n, O)).astype(np.float32)
            got = _dense_wgrad(dev(X), dev(G)).cpu().numpy()
            np.testing.assert_allclose(got, X.astype(np.float64).T @ G, rtol=1e-4, atol=2e-4 * np.sqrt(n), err_msg=f"wgrad case {case}")
            stats["wgrad"] += 1
    elif kind == 6:
        n, C, m = int(rng.integers(1, 5000)), int(rng.integers(1, 200)), int(rng.integers(1, 9000))
        L = (rng.standard_normal((n, C)) * 4).astype(np.float32)
        nodes, labels = rng.integers(0, n, size=m), rng.integers(0, C, size=m)
        got = float(gnntf.node_ce(dev(L), nodes, labels))
        want = orc.node_loss(L.astype(np.float64), nodes, labels)
        assert abs(got - want) <= 2e-5 * max(abs(want), 1), f"head case {case}: {got} {want}"
        assert np.array_equal(gnntf.node_argmax(dev(L), nodes).cpu().numpy(), L[nodes].argmax(1))
        stats["head"] += 1
    else:
        n, C, m = int(rng.integers(2, 5000)), int(rng.integers(1, 200)), int(rng.integers(1, 9000))
        F = rng.standard_normal((n, C)).astype(np.float32)
        e = rng.integers(0, n, size=(m, 2))
        np.testing.assert_allclose(gnntf.edge_scores(dev(F), e).cpu().numpy(), orc.link_logits(F.astype(np.float64), e), rtol=1e-4, atol=1e-4, err_msg=f"edge case {case}")
        stats["edge"] += 1
    if case % 50 == 49:
        print(f"{case + 1} cases, {time.time() - t0:.0f} s", stats, flush=True)
torch.cuda.synchronize()
print("FUZZ OK", cases, "cases, seed", seed, stats)
