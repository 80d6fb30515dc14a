"""bench_secondary.py -- the secondary workloads of the N = 1 bench run, timed in the same process after the headline so that they
are driver-timed too: BASELINE.json configs[3] (the roofline run, RMAT 10M / 100M, C = 256) and the other widths on that graph,
the same propagation through the layer API, a training step and its two launches at the widths gnntf trains at, the matrix-core
ends of the path, and the small configs (2: Cora-shaped APPNP, 3: arxiv-shaped GCN).  Every part returns (detail, flat): ``detail``
goes to the detail file, ``flat`` = the scalars the stdout line's roofline object carries."""
import argparse
import os
import sys
import time

import bench_record as br
from bench_device import build_single, kept_entries, median_ms
from bench_pmc import SEGMENT_WIDTHS, TRAIN_WIDTHS, training_launches
from bench_record import WORKLOADS, workload_name


def gather_yardstick(device, n, widths, a=0.1, d=16):
    """The no-reuse gather ceiling, measured in this run: one fused SpMM+mix launch over a graph whose every row has ``d`` uniformly
    random neighbours (round 4's regular-graph sweep) at the SAME N, for every width of ``widths``.  N * C * 4 bytes is far beyond
    the caches and no row is gathered more often than any other, so B_alg / t of THIS launch is a DRAM-level rate: what the chip
    gathers whole random rows at (wide rows), or -- at narrow widths, where a gather moves a 128-byte line for a 32-byte row --
    what the line granularity leaves of it.  R-MAT's figures above it are the hub rows served on-die.  Returns {C: record}."""
    import torch
    import gnntf
    from gnntf.sparse import _launch
    rows = torch.arange(n, device=device).repeat_interleave(d)
    cols = torch.randint(0, n, (n * d,), device=device)
    idx = torch.stack([rows, cols], 1)
    del rows, cols
    g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=device), (n, n)), device=device)
    del idx
    torch.cuda.empty_cache()
    adj = gnntf.Adjacency(g)
    out = {}
    for C in widths:
        H, H0 = torch.rand(n, C, device=device), torch.rand(n, C, device=device)
        res = torch.empty_like(H)
        ms = median_ms(lambda: _launch(adj, H, H0, 1.0 - a, a, 0, out=res), reps=3, warm=1)
        out[C] = {"GBs": br.alg_bytes_per_iteration(n, g.nnz, C) / ms / 1e6, "launch_ms": ms, "rows": n, "entries": g.nnz, "d": d, "C": C,
                  "kernel": g.last_kernel()}
        del H, H0, res
    del g, adj
    torch.cuda.empty_cache()
    return out


def _propagate(lib, nat, g, adj, H0, a, K, C, res, work, act=0):
    nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(res), nat.ptr(work), nat.current_stream()))


def widths_on_config4(ctx, skip_config4):
    """K = 10 propagation at every width of SEGMENT_WIDTHS on the config-4 graph, each with its roofline record and the in-run
    no-reuse yardstick at the same width."""
    import torch
    g, adj, device, K, a = ctx.g, ctx.adj, ctx.device, ctx.K, ctx.a
    n4, e4, c4 = WORKLOADS["config4"]
    n, nnz = g.n_rows, g.nnz
    out, flat, widths, yards = {}, {}, [], {}
    if ctx.args.gather_yardstick == "on":
        yards = gather_yardstick(device, n4, [w for w in SEGMENT_WIDTHS if not (skip_config4 and w == c4)], a)
        out["config4_no_reuse_gather_yardstick"] = [yards[w] for w in sorted(yards, reverse=True)]
    for C in ([] if skip_config4 else [c4]) + [w for w in SEGMENT_WIDTHS if w != c4]:
        gen = torch.Generator(device=device).manual_seed(2)
        H0 = torch.rand(n, C, device=device, generator=gen) * 2 - 1
        res, work = torch.empty_like(H0), torch.empty_like(H0)
        ms = median_ms(lambda: _propagate(ctx.lib, ctx.nat, g, adj, H0, a, K, C, res, work), reps=3, warm=1)
        roof = br.roofline_record(n, nnz, C, ms * 1e-3 / K, K, workload_name(n4, e4, C), ctx.measured_peak)
        rec = {"C": C, "kernel": g.last_kernel(), "ms_per_step": ms, "edges_per_s": nnz * K / ms * 1e3, "roofline": roof}
        br.add_gather_ceiling(roof, yards.get(C))
        if C == c4:
            out["config4_roofline_run"] = dict(rec, workload=workload_name(n4, e4, C) + f"_appnp_K{K}", prep=ctx.prep)
            flat.update(br.flat_keys("config4", roof, ms_per_step=ms, edges_per_s=rec["edges_per_s"]))
            flat.update(config4_workload=workload_name(n4, e4, C) + f"_appnp_K{K}", config4_rows=n, config4_entries=nnz, config4_kernel=rec["kernel"])
        else:
            widths.append(rec)
            flat.update(br.triple(f"config4_graph_C{C}", roof))
            if roof.get("no_reuse_gather_frac") is not None:
                flat[f"config4_graph_C{C}_no_reuse_gather_frac"] = roof["no_reuse_gather_frac"]
        del H0, res, work
    out["config4_graph_other_widths"] = widths
    flat.setdefault("config4_rows", n)
    flat.setdefault("config4_entries", nnz)
    return out, flat


def via_layer_api(ctx, skip_config4):
    """The same propagation through the API the north star names: architecture.predict() of gnntf.APPNP (filter.py:25-35 ->
    trainable.py:26-29) on the config-4 graph.  APPNP builds filter.py:30-35's own list [Dropout, Dense(F -> C), K x PPRIteration];
    the container executes the K layers as one fused run, which must cost what gnx_appnp_propagate costs and return the same bits.
    Then the same stack as user code builds it (reference demos/custom_layers.py:8-13), fused and layer by layer, and -- C = 8 --
    the reference's activation argument (filter.py:28,35: relu in every PPRIteration), fused against layer by layer."""
    import torch
    import gnntf
    g, adj, device, K, a, lib, nat = ctx.g, ctx.adj, ctx.device, ctx.K, ctx.a, ctx.lib, ctx.nat
    n, nnz = g.n_rows, g.nnz
    c4 = WORKLOADS["config4"][2]
    via_api, flat = [], {}
    for C in ([] if skip_config4 else [c4]) + [8]:
        gnntf.set_seed(0)
        F = 64
        X = torch.randn(n, F, device=device)
        model = gnntf.APPNP(g, X, num_classes=C, latent_dims=[], iterations=K, a=a)
        model.reset()                                            # variables are zero until reset() (variables.py:62-66; train() calls it)
        model.training_mode(False)
        nodes = torch.randperm(n, device=device)[:100_000]
        task = gnntf.NodeClassification(nodes)

        def predict():
            model._fast_predict = None                           # trainable.py:22-24: what reset() clears; every call recomputes
            return model.predict(task)
        first = len(model.layers()) - K                          # index of the first PPRIteration layer
        with torch.no_grad():
            t_predict = median_ms(predict, reps=3, warm=1)
            H0 = model.layers()[first - 1].value
            t_loop = median_ms(lambda: model.run(H0, first=first), reps=3, warm=1)
            kernel = g.last_kernel()
            res, work = torch.empty_like(H0), torch.empty_like(H0)
            t_direct = median_ms(lambda: _propagate(lib, nat, g, adj, H0, a, K, C, res, work), reps=3, warm=1)
            same = bool(torch.equal(model.layers()[-1].value, res))
        via_api.append({"C": C, "layers": [type(l).__name__ for l in model.layers()], "n_layers": len(model.layers()), "predict_ms": t_predict,
                        "propagation_layers_ms": t_loop,
                        "gnx_appnp_propagate_ms": t_direct, "layers_over_direct": t_loop / t_direct, "bitwise_equal": same, "kernel": kernel,
                        "edges_per_s_layers": nnz * K / t_loop * 1e3,
                        "what": f"gnntf.APPNP(graph, X[N, {F}], num_classes={C}, latent_dims=[]) in eval mode, the reference's layer list: predict_ms = "
                                f"architecture.predict(NodeClassification(100k nodes)) with the memo cleared (Dense {F} -> {C} on the matrix cores + K = {K} "
                                f"propagation + gather/argmax); propagation_layers_ms = the K PPRIteration layers alone (architecture.run(H0, first=...)); "
                                f"gnx_appnp_propagate_ms = the C entry on the same H0"})
        flat.update({f"config4_C{C}_via_layers_ms": t_loop, f"config4_C{C}_c_entry_ms": t_direct, f"config4_C{C}_layers_bitwise_equal_c_entry": same})
        for layer in model.layers():
            layer.value = None
        del model, H0, res, work, task, nodes
        torch.cuda.empty_cache()
        # the same propagation as user code builds it (reference demos/custom_layers.py:8-13): a Dense and K hand-added
        # PPRIteration(H0, a) layers.  The container runs them as one fused loop (Layer.__run__); fuse_runs = False is the
        # layer-by-layer execution of the same stack (K launches, K intermediate values)
        for act_name in (["linear", "relu"] if C == 8 else ["linear"]):
            gnntf.set_seed(0)
            hand = gnntf.GNN(g, X)
            H0l = hand.add(gnntf.Dense(C, regularize=False))
            for _ in range(K):
                hand.add(gnntf.PPRIteration(H0l, a) if act_name == "linear" else gnntf.PPRIteration(H0l, a, activation=gnntf.relu))
            hand.reset()
            hand.training_mode(False)
            with torch.no_grad():
                t_hand = median_ms(lambda: hand(hand.features), reps=3, warm=1)
                t_dense = median_ms(lambda: H0l(hand, hand.features), reps=3, warm=1)
                fused_out = hand(hand.features)
                hand.fuse_runs = False
                t_hand_layers = median_ms(lambda: hand(hand.features), reps=3, warm=1)
                by_layer = hand(hand.features)
                for layer in hand.layers():
                    layer.value = None
            rec = {"layers": [type(l).__name__ for l in hand.layers()][:3] + ["..."], "activation": act_name, "forward_ms": t_hand,
                   "dense_alone_ms": t_dense, "propagation_ms": t_hand - t_dense, "layer_by_layer_forward_ms": t_hand_layers,
                   "bitwise_equal_to_layer_by_layer": bool(torch.equal(fused_out, by_layer)),
                   "max_abs_difference_to_layer_by_layer": float((fused_out - by_layer).abs().max()),
                   "what": f"GNN(graph, X) + Dense({C}) + {K} x PPRIteration(H0, {a}, activation={act_name}) added by hand, eval mode: forward_ms with "
                           f"the container fusing the run (propagation_ms = forward - the Dense alone: to be compared with "
                           f"propagation_layers_ms), layer_by_layer_forward_ms with fuse_runs = False"}
            via_api[-1]["hand_built_stack" + ("" if act_name == "linear" else "_relu")] = rec
            if act_name == "relu":
                # (10M vertices at C = 8: the fused loop runs on the relabelled copy, so it agrees with the layer-by-layer form to float32
                #  rounding, not bitwise -- gnx.h, gnx_appnp_propagate)
                flat.update(config4_C8_relu_fused_forward_ms=t_hand, config4_C8_relu_layer_by_layer_forward_ms=t_hand_layers,
                            config4_C8_relu_fused_max_abs_diff=rec["max_abs_difference_to_layer_by_layer"])
            del hand, H0l, fused_out, by_layer
        del X
        torch.cuda.empty_cache()
    return {"config4_via_layer_api": via_api}, flat


def community_graph(ctx):
    """A graph WITH communities (planted partition x power-law degrees, same N, ~ the same entries: the structure the reference's
    citation datasets have and R-MAT lacks): gnntf.APPNP in its default order against GNN(reorder="locality") -- label propagation
    order + row windows (gnx_graph_set_row_window) -- at the widths gnntf's APPNP propagates; same model, same weights, the
    outputs compared in the caller's order; prep = what the reordered model's construction costs beyond the plain one's.
    Detail file only (validated on the builder's own generator only)."""
    import torch
    import gnntf
    from gnntf.rmat import community_pairs
    device, K, a = ctx.device, ctx.K, ctx.a
    n4, e4, _ = WORKLOADS["config4"]
    u, v, _ = community_pairs(n4, e4 // 2, 1, device)
    pairs = torch.unique(torch.minimum(u, v) * n4 + torch.maximum(u, v))      # every undirected pair once (no duplicate entries: the
    u, v = torch.div(pairs, n4, rounding_mode="floor"), pairs % n4             # training launches then draw inside the SpMM)
    cidx = torch.cat([torch.stack([u, v], 1), torch.stack([v, u], 1)])
    del u, v, pairs
    ccoo = gnntf.SparseCOO(cidx, torch.ones(cidx.shape[0], device=device), (n4, n4))
    Xc = torch.randn(n4, 16, device=device)
    comm_rec = {"what": "planted-partition x power-law graph (gnntf.rmat.community_pairs: communities of 64 ... 65536 vertices, 20 % of the pairs "
                        "leave their community, vertices randomly relabelled), gnntf.APPNP(..., latent_dims=[]) in eval mode: the K PPRIteration "
                        "layers alone (architecture.run(H0, first=2)), default order against reorder=\"locality\"", "widths": []}
    for C in (40, 8):
        per = {"C": C}
        outs = {}
        for reorder in (None, "locality"):
            gnntf.set_seed(0)
            torch.manual_seed(0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cm = gnntf.APPNP(ccoo, Xc, num_classes=C, latent_dims=[], iterations=K, a=a, reorder=reorder)
            torch.cuda.synchronize(); t_build = time.perf_counter() - t0
            cm.reset()
            cm.training_mode(False)
            with torch.no_grad():
                outs[reorder] = cm(cm.features)
                H0c = cm.layers()[1].value
                ms = median_ms(lambda: cm.run(H0c, first=2), reps=3, warm=1)
            key = "locality" if reorder else "default"
            per[key + "_ms"], per[key + "_build_s"], per[key + "_kernel"] = ms, t_build, cm.graph.last_kernel()
            if reorder:
                per["reorder_used"], per["locality_share"], per["entries"] = cm.reorder_used, cm.locality_share, cm.graph.nnz
            for layer in cm.layers():
                layer.value = None
            del cm, H0c
            torch.cuda.empty_cache()
        per["max_abs_difference_of_the_outputs"] = float((outs[None] - outs["locality"]).abs().max())
        per["argmax_equal_share"] = float((outs[None].argmax(1) == outs["locality"].argmax(1)).float().mean())
        per["time_ratio"] = per["locality_ms"] / per["default_ms"]
        comm_rec["widths"].append(per)
        del outs
    del cidx, ccoo, Xc
    torch.cuda.empty_cache()
    return {"community_graph_locality_order": comm_rec}, {}


def training_step(ctx):
    """Training-mode step: K = 10 iterations, each with its own dropped + re-normalised adjacency, forward +
    backward through the fused loop node (masks regenerated in the backward), at every width of TRAIN_WIDTHS (64; 40 and 7 = the
    widths gnntf trains at: filter.py:33-35, trainable.py:70-78); and the two launches the step consists of, each timed alone with
    events and priced against its own byte model."""
    import torch
    import gnntf
    g, device, K, a = ctx.g, ctx.device, ctx.K, ctx.a
    n4, e4, _ = WORKLOADS["config4"]
    n, nnz = g.n_rows, g.nnz
    out, flat = {}, {}
    kept = kept_entries(g, 0.5, 1, 1, 1)[0]
    for C in TRAIN_WIDTHS:
        H0 = (torch.rand(n, C, device=device) * 2 - 1).requires_grad_()
        gout = torch.rand(n, C, device=device)
        two_pass = lambda k, bwd=False: gnntf.normalize(g, "symmetric", "none", dropout=0.5, seed=1, stream_id=k, transposed_only=bwd)

        def train_step(make):
            H0.grad = None
            if make is None:                                        # what PPRLoop does: all K degree-scale vectors in one pass, kept for the backward
                scales = gnntf.sparse.dropped_degree_scales(g, 0.5, 1, 0, K)
                make = lambda k, bwd=False: gnntf.sparse.dropped_adjacency(g, 0.5, 1, k, D=scales[k])
            gnntf.ppr_loop(make, H0, a, K).backward(gout)
        ms = median_ms(lambda: train_step(None), reps=3, warm=1)
        ms2 = median_ms(lambda: train_step(two_pass), reps=3, warm=1) if C == TRAIN_WIDTHS[0] else None
        del H0, gout
        forward, backward, keep = training_launches(g, C, a, K, device)
        with torch.no_grad():
            ms_f = median_ms(forward, reps=5, warm=2)
            kernel_f = g.last_kernel()
            ms_b = median_ms(backward, reps=5, warm=2)
            ms_d = median_ms(lambda: gnntf.sparse.dropped_degree_scales(g, 0.5, 1, 0, K), reps=3, warm=1)
        del forward, backward, keep
        wl = workload_name(n4, e4, C)
        roof_f = br.roofline_record(n, nnz, C, ms_f * 1e-3, K, "train_forward_" + wl, ctx.measured_peak,
                                    b_alg=br.alg_bytes_dropped_iteration(n, nnz, kept, C), b_min=br.min_bytes_dropped_iteration(n, nnz, C),
                                    what="one forward TRAINING iteration (gnx_spmm_dropped_chained, "
                                    "a middle one: weights from the counter RNG inside the SpMM, only kept entries gathered, rows without "
                                    "entries left to the last iteration) incl. its long-row kernels")
        roof_b = br.roofline_record(n, nnz, C, ms_b * 1e-3, K, "train_backward_" + wl, ctx.measured_peak,
                                    b_alg=br.alg_bytes_dropped_iteration(n, nnz, kept, C, backward=True),
                                    b_min=br.min_bytes_dropped_iteration(n, nnz, C, backward=True),
                                    what="one backward TRAINING iteration "
                                    "(gnx_spmm_dropped_back over the transposed structure: the running gradient sum updated and the next step's "
                                    "pre-scaled operand written in the epilogue) incl. its long-row kernels")
        flat.update(br.triple(f"train_C{C}_forward", roof_f), **br.triple(f"train_C{C}_backward", roof_b))
        flat[f"train_C{C}_step_ms"] = ms
        out[f"training_step_C{C}"] = {"ms": ms, "two_pass_ms": ms2, "edges_per_s": 2 * nnz * K / ms * 1e3,
                                      "forward_launch_ms": ms_f, "backward_launch_ms": ms_b, "degree_scales_all_streams_ms": ms_d,
                                      "launches_share_of_step": (K * (ms_f + ms_b) + ms_d) / ms, "kept_entries": kept, "kernel": kernel_f,
                                      "launched_at_width": gnntf.sparse.friendly_width(C, n),
                                      "roofline": roof_f, "roofline_backward": roof_b,
                                      "what": f"forward + backward of {K} PPR iterations with per-iteration edge dropout 0.5 + renormalisation, "
                                              f"config-4 graph, C={C}; ms: weights produced inside the SpMM (gnx_spmm_dropped), two_pass_ms: "
                                              f"materialised per iteration (gnx_graph_normalize + gnx_spmm); roofline / roofline_backward: one "
                                              f"forward / backward iteration's launch timed alone, byte model alg_bytes_dropped_iteration (col + raw "
                                              f"value of EVERY entry, a neighbour row per KEPT entry, H0 + out + scales per row)"}
        torch.cuda.empty_cache()
    flat["train_kept_entries"] = kept
    # the same step with the vertices in the model-level degree order (GNN(reorder="degree"), an opt-in: results agree to float32
    # rounding): the training kernels walk the caller's numbering, so hub rows that are neighbours in memory pay here (NOTES round 6)
    rowptr, colidx, _ = g.csr_arrays()
    deg = rowptr[1:] - rowptr[:-1]
    order = torch.argsort(deg, descending=True, stable=True)
    newid = torch.empty_like(order)
    newid[order] = torch.arange(n, device=device)
    rows = torch.repeat_interleave(torch.arange(n, device=device), deg)
    idx = torch.stack([newid[rows], newid[colidx.long()]], 1)
    del rows, rowptr, colidx, deg, order, newid
    gd = gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(idx.shape[0], device=device), (n, n)), device=device)
    del idx
    for C in (TRAIN_WIDTHS[-1], TRAIN_WIDTHS[1]):
        H0 = (torch.rand(n, C, device=device) * 2 - 1).requires_grad_()
        gout = torch.rand(n, C, device=device)

        def step():
            H0.grad = None
            scales = gnntf.sparse.dropped_degree_scales(gd, 0.5, 1, 0, K)
            gnntf.ppr_loop(lambda k, bwd=False: gnntf.sparse.dropped_adjacency(gd, 0.5, 1, k, D=scales[k]), H0, a, K).backward(gout)
        ms = median_ms(step, reps=3, warm=1)
        out[f"training_step_C{C}"]["degree_order_ms"] = ms
        flat[f"train_C{C}_step_degree_order_ms"] = ms
        del H0, gout
    del gd
    torch.cuda.empty_cache()
    return out, flat


def matrix_core_kernels(ctx):
    """The matrix-core ends of the path at the config-4 size.  Detail file only."""
    import torch
    import gnntf
    g, adj, device, a = ctx.g, ctx.adj, ctx.device, ctx.a
    n = g.n_rows
    mf = {}
    for C in (64, 128):
        H, H0 = torch.rand(n, C, device=device) * 2 - 1, torch.rand(n, C, device=device) * 2 - 1
        M = 0.6 * torch.eye(C, device=device) + 0.4 * torch.randn(C, C, device=device) / 8
        with torch.no_grad():
            t_fused = median_ms(lambda: gnntf.gcnii_step(adj, H, H0, a, M, relu=True), reps=5, warm=2)
            kernel = g.last_kernel()
            t_two = median_ms(lambda: gnntf.dense(gnntf.ppr_step(adj, H, H0, a), M, None, relu=True), reps=5, warm=2)
        # training: forward + backward of the layer (dM, dH, dH0); the fused launch also writes the mixed rows it would otherwise re-read
        Ht, H0t, Mt = H.clone().requires_grad_(), H0.clone().requires_grad_(), M.clone().requires_grad_()
        up = torch.rand(n, C, device=device)

        def train(fused):
            for t in (Ht, H0t, Mt):
                t.grad = None
            out = gnntf.gcnii_step(adj, Ht, H0t, a, Mt, relu=True) if fused else gnntf.dense(gnntf.ppr_step(adj, Ht, H0t, a), Mt, None, relu=True)
            out.backward(up)
        t_train = median_ms(lambda: train(True), reps=3, warm=1)
        t_train_two = median_ms(lambda: train(False), reps=3, warm=1)
        mf[f"gcnii_layer_C{C}"] = {"fused_ms": t_fused, "spmm_then_dense_ms": t_two, "kernel": kernel,
                                   "train_fwd_bwd_fused_ms": t_train, "train_fwd_bwd_two_launch_ms": t_train_two,
                                   "what": "relu(((1-a) A.H + a H0) . M) on the config-4 graph: one launch (mixed rows stay in LDS, MFMA epilogue) vs "
                                           "fused SpMM+mix followed by gnx_dense; train_*: forward + backward of the layer, the fused launch writing "
                                           "the mixed rows the backward needs"}
        del H, H0, Ht, H0t, Mt, up, M
        torch.cuda.empty_cache()
    X = torch.randn(n, 256, device=device)
    W, b = torch.randn(256, 64, device=device) / 16, torch.randn(1, 64, device=device)
    with torch.no_grad():
        t_dense = median_ms(lambda: gnntf.dense(X, W, b, relu=True), reps=5, warm=2)
        t_torch = median_ms(lambda: torch.relu(torch.addmm(b, X, W)), reps=5, warm=2)      # hipBLASLt GEMM + separate bias / relu passes
    mf["dense_10M_x_256_to_64_relu"] = {"ms": t_dense, "torch_addmm_relu_ms": t_torch, "TFLOPs": 2.0 * n * 256 * 64 / t_dense / 1e9,
                                        "GBs": (n * 256 * 4 + n * 64 * 4) / t_dense / 1e6, "mfma_peak_TFLOPs": 157.3,
                                        "what": "gnx_dense (k_dense_wreg: W in registers, X through an LDS-DMA ring), float32 v_mfma_f32_16x16x4_f32; "
                                                "X read once from HBM"}
    from gnntf.sparse import _dense_wgrad
    Gd = torch.randn(n, 64, device=device)
    t_wgrad = median_ms(lambda: _dense_wgrad(X, Gd), reps=5, warm=2)
    t_wgrad_torch = median_ms(lambda: X.t() @ Gd, reps=3, warm=1)
    mf["dense_wgrad_10M_x_256_x_64"] = {"ms": t_wgrad, "torch_matmul_ms": t_wgrad_torch, "TFLOPs": 2.0 * n * 256 * 64 / t_wgrad / 1e9,
                                        "GBs": (n * 256 * 4 + n * 64 * 4) / t_wgrad / 1e6,
                                        "what": "gnx_dense_wgrad (k_wgrad_acc: every wave keeps a whole 256 x 64 partial in registers), dW = X^T . G"}
    del X, Gd
    logits = torch.randn(n, 40, device=device)
    nodes = torch.randperm(n, device=device)[:1_000_000]
    labels = torch.randint(0, 40, (1_000_000,), device=device)
    t_head = median_ms(lambda: gnntf.node_ce(logits, nodes, labels), reps=5, warm=2)
    mf["node_ce_1M_nodes_C40"] = {"ms": t_head, "what": "gather + log-softmax + cross entropy + mean, two launches"}
    return {"matrix_core_kernels": mf}, {}


def small_configs(ctx):
    """config 3: arxiv-shaped 2-layer GCN forward (N = 169,343; 1,166,243 undirected pairs -> 2,332,486 stored entries; 128 -> 64 ->
    40).  config 2: Cora-shaped APPNP (N = 2708, F = 1433 at 1.3 % density, C = 7, K = 10): the launch-latency regime -- ms per
    training epoch of architecture.train(), eager and replayed from hipGraphs (train(capture=True)), and the eval forward."""
    import numpy as np
    import torch
    import gnntf
    device = ctx.device
    g3, adj3, _ = build_single(argparse.Namespace(nodes=169_343, entries=2_332_486), device)
    X = torch.randn(g3.n_rows, 128, device=device)
    model = gnntf.GCN(g3, X, num_classes=40)
    model.training_mode(False)
    with torch.no_grad():
        t_fwd = median_ms(lambda: model(model.features), reps=20, warm=5)
        X64 = torch.randn(g3.n_rows, 64, device=device)
        t128 = median_ms(lambda: gnntf.spmm(adj3, X), reps=20, warm=5)
        t64 = median_ms(lambda: gnntf.spmm(adj3, X64), reps=20, warm=5)
    sys.path.insert(0, os.path.join(br.ROOT, "tests"))
    import graphs as test_graphs
    coo, vals, shape, Xc = test_graphs.cora_shaped(seed=0)
    labels = np.random.default_rng(0).integers(0, 7, size=shape[0])
    tr, va = list(range(140)), list(range(140, 640))
    tasks = lambda: dict(train=gnntf.NodeClassification(tr, labels[tr]), valid=gnntf.NodeClassification(va, labels[va]))
    cora = {}
    for capture in (False, True):
        gnntf.set_seed(0)
        m2 = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), Xc, num_classes=7)
        m2.train(epochs=5, patience=5, capture=capture, **tasks())
        spans = []
        for epochs in (50, 150):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m2.train(epochs=epochs, patience=1000, capture=capture, **tasks())
            torch.cuda.synchronize(); spans.append(time.perf_counter() - t0)
        cora["captured_train_ms_per_epoch" if capture else "train_ms_per_epoch"] = (spans[1] - spans[0]) / 100 * 1e3
    with torch.no_grad():
        cora["eval_forward_ms"] = median_ms(lambda: m2(m2.features), reps=50, warm=5)
    out = {"config2_cora_shaped_appnp": cora,
           "config3_arxiv_shaped_gcn": {"nodes": g3.n_rows, "entries": g3.nnz, "forward_ms": t_fwd, "spmm128_ms": t128, "spmm64_ms": t64,
                                        "spmm128_edges_per_s": g3.nnz / t128 * 1e3, "spmm64_edges_per_s": g3.nnz / t64 * 1e3}}
    flat = {"config3_gcn_forward_ms": t_fwd, "config3_spmm128_edges_per_s": g3.nnz / t128 * 1e3, "config3_spmm64_edges_per_s": g3.nnz / t64 * 1e3,
            "config2_eval_forward_ms": cora["eval_forward_ms"], "config2_captured_train_ms_per_epoch": cora["captured_train_ms_per_epoch"]}
    return out, flat


# (part, needs the config-4 graph, seconds it takes on an MI355X: what --max-seconds must still hold for it to start)
PARTS = (("widths_on_config4", True, 12.0), ("via_layer_api", True, 8.0), ("training_step", True, 10.0), ("matrix_core_kernels", True, 8.0),
         ("community_graph", False, 20.0), ("small_configs", False, 8.0))


def secondary_workloads(args, device, measured_peak, skip_config4=False, deadline=None):
    """Runs the parts in order, each only if --max-seconds still has room for it.  Returns (detail, flat)."""
    import torch
    from gnntf import _native as nat
    n4, e4, _ = WORKLOADS["config4"]
    g, adj, prep = build_single(argparse.Namespace(nodes=n4, entries=e4), device)
    ctx = argparse.Namespace(args=args, device=device, measured_peak=measured_peak, g=g, adj=adj, prep=prep, K=args.iterations, a=args.alpha,
                             nat=nat, lib=nat.lib())
    table = {"widths_on_config4": lambda: widths_on_config4(ctx, skip_config4), "via_layer_api": lambda: via_layer_api(ctx, skip_config4),
             "training_step": lambda: training_step(ctx), "matrix_core_kernels": lambda: matrix_core_kernels(ctx),
             "community_graph": lambda: community_graph(ctx), "small_configs": lambda: small_configs(ctx)}
    detail, flat = {}, {}
    for name, on_graph, seconds in PARTS:
        if not on_graph and ctx.g is not None:                   # the parts with graphs of their own: the config-4 graph goes first
            ctx.g = ctx.adj = None
            del g, adj
            torch.cuda.empty_cache()
        if deadline is not None and not deadline.room(seconds, "secondary: " + name):
            continue
        t0 = time.time()
        br.note("secondary: " + name)
        try:
            d, f = table[name]()
        except Exception as error:                               # one part failing must not take the others' results with it
            br.note(f"secondary: {name} failed and is left out: {error!r}")
            if deadline is not None:
                deadline.dropped.append(f"secondary: {name} failed: {error!r}"[:200])
            torch.cuda.empty_cache()
            continue
        detail.update(d)
        flat.update(f)
        br.PHASES["secondary_" + name] = round(time.time() - t0, 2)
    return detail, flat
