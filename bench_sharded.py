"""bench_sharded.py -- the N > 1 side of bench.py: ONE global graph on pv contiguous vertex blocks (gnntf.sharded), a pairwise RCCL
exchange of pulled rows / pushed partial sums per iteration.  Which halo plan / pipelining is fastest depends on what the links
of THIS node sustain, which nothing on a one-GPU box can tell: before the timed region one step of every variant is timed (max
over ranks) inside a wall-clock budget and the timed steps run on the fastest.

select_variant is host logic over an ``ops`` object (tests/test_bench_sharded.py drives it with a fake clock); BlockOps is the
real thing.  Every decision is taken from numbers all ranks hold identically (all-reduced times and flags), so every rank takes
the same branch; every early exit is collective and non-zero (agree())."""
import time

import bench_record as br
from bench_record import note


def select_variant(covers, chunk_order, ops, budget, K):
    """Times halo variants (cover x column chunks x early_pull) inside ``budget`` seconds of ops.spent().

    The FIRST variant is timed unconditionally (one step); everything after it starts only if what it is predicted to cost still
    fits: a further plan = what the first plan took to build; a variant's set-up (state + the bare exchange and the bare kernels
    of one iteration, which are measured FIRST) = what the previous one's took; a full step = K x (exchange + kernels) of this
    variant, the no-overlap bound.  So the selection ends within budget + one step.  Losing plans are freed before the next one
    is built.  Returns (best variant or None, variants, skipped)."""
    variants, skipped = [], []
    best = None
    alone_s = 0.0                                  # what the last variant's set-up + bare measurements took

    def skip(reason, **which):
        skipped.append(dict(which, reason=reason))
        note(f"selection: skipped {which}: {reason}")

    for cover in covers:
        if not ops.has_plan(cover):
            spent, need = ops.spent(), ops.plan_seconds()
            if best is not None and spent + need > budget:
                skip(f"plan not built: {spent:.1f} s of the {budget:.0f} s selection budget spent, a plan takes {need:.1f} s", cover=cover)
                continue
            ops.keep_only(best["cover"] if best else None)          # a losing plan goes before the next one is built
            if not ops.build_plan(cover):
                variants.append(dict(cover=cover, chunks=None, early_pull=None, step_ms=None, error="plan could not be built on some rank"))
                continue
        for chunks in chunk_order:
            if best is not None and ops.spent() + alone_s > budget:
                skip("selection budget spent", cover=cover, chunks=chunks)
                continue
            t0 = ops.spent()
            state, problem = ops.make_state(cover, chunks)
            if state is None:
                variants.append(dict(cover=cover, chunks=chunks, early_pull=None, step_ms=None, error=problem))
                continue
            ex_ms, comp_ms = ops.time_alone(cover, state)
            alone_s = ops.spent() - t0
            predicted = K * (ex_ms + comp_ms) * 1e-3                # a step with nothing overlapped
            for early in ops.early_options(cover):
                if best is not None and ops.spent() + predicted > budget:
                    skip(f"a step is predicted at {predicted:.1f} s (K x (exchange + kernels)), {max(budget - ops.spent(), 0.0):.1f} s left",
                         cover=cover, chunks=chunks, early_pull=early)
                    continue
                ms = ops.run_step(cover, state, early)              # (the first call also opens this variant's connections)
                if ops.spent() + ms * 1e-3 <= budget:
                    ms = min(ms, ops.run_step(cover, state, early))
                v = dict(cover=cover, chunks=chunks, early_pull=early, step_ms=ms, exchange_ms_alone=ex_ms, compute_ms_alone=comp_ms)
                variants.append(v)
                note(f"variant {v}")
                if best is None or ms < best["step_ms"]:
                    best = v
            ops.release(state)
    ops.keep_only(best["cover"] if best else None)
    return best, variants, skipped


class BlockOps:
    """select_variant's operations on the real vertex blocks.  All collective."""

    def __init__(self, args, dist, device, idx, vals, bounds, comm, H0, chunk_options, a, K):
        import torch
        self.torch, self.dist, self.device, self.args = torch, dist, device, args
        self.idx, self.vals, self.bounds, self.comm, self.H0 = idx, vals, bounds, comm, H0
        self.chunk_options, self.a, self.K = chunk_options, a, K
        self.graphs, self.plan_s, self.plan_peak_bytes = {}, {}, None
        # ranks that share this rank's card (rehearsals: several gloo ranks on one GPU): they all draw on the same free memory
        self.sharers = max(1, -(-dist.get_world_size() // max(torch.cuda.device_count(), 1))) if dist.is_initialized() else 1
        self.t0 = time.perf_counter()

    def agree(self, ok, what, fatal=False):
        """True only if ``ok`` on EVERY rank; fatal: all ranks leave together, non-zero."""
        t = self.torch.tensor([1 if ok else 0], device=self.device, dtype=self.torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        if int(t.item()) == 0 and fatal:
            note("fatal on some rank: " + what)
            raise SystemExit(f"bench.py: {what} failed on at least one rank; all ranks stop")
        return int(t.item()) == 1

    def slowest(self, seconds):
        t = self.torch.tensor([seconds], device=self.device, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def start(self):
        self.t0 = time.perf_counter()

    def spent(self):
        """Seconds since the selection began on the SLOWEST rank: every rank sees the same number and takes the same branch."""
        return self.slowest(time.perf_counter() - self.t0)

    def has_plan(self, cover):
        return cover in self.graphs

    def plan_seconds(self):
        return max(self.plan_s.values()) if self.plan_s else 0.0

    def build_plan(self, cover, fatal=False):
        """Builds a plan (collective).  A further plan is built only if EVERY rank has room for it -- free device memory against
        what the first plan's construction peaked at, decided collectively BEFORE the constructor -- because a rank that fails
        inside the constructor leaves its peers blocked in the constructor's collectives: such a failure is not caught here, the
        rank dies non-zero and the launcher ends the others."""
        from gnntf import sharded
        torch = self.torch
        if self.plan_peak_bytes is not None:
            free = torch.cuda.mem_get_info(self.device)[0] + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            if not self.agree(free >= 1.5 * self.plan_peak_bytes * self.sharers, f"memory for the {cover} plan", fatal=fatal):
                note(f"the {cover} plan is not built: not enough free device memory on some rank")
                return False
        t0 = time.time()
        kind, _, weight = cover.partition("@")
        torch.cuda.reset_peak_memory_stats(self.device)
        before = torch.cuda.memory_allocated(self.device)
        self.graphs[cover] = sharded.ShardedGraph(self.idx, self.vals, self.bounds, comm=self.comm, cover=kind, chunks=self.chunk_options[-1],
                                                  push_weight=float(weight or 0.0), split_rows=not self.args.whole_rows, relabel=True,
                                                  tune_overlap=self.args.overlap_probe == "on")
        torch.cuda.synchronize()
        if self.plan_peak_bytes is None:
            self.plan_peak_bytes = torch.cuda.max_memory_allocated(self.device) - before
        self.plan_s[cover] = round(self.slowest(time.time() - t0), 2)
        br.PHASES["plan_" + cover] = self.plan_s[cover]
        return True

    def keep_only(self, cover):
        for c in list(self.graphs):
            if c != cover and cover is not None:
                del self.graphs[c]
        self.torch.cuda.empty_cache()

    def make_state(self, cover, chunks):
        state, problem = None, ""
        try:
            state = self.graphs[cover].make_state(self.H0, chunks=chunks)
        except Exception as error:
            problem = repr(error)[:200]
        if not self.agree(not problem, "state"):                    # a variant that cannot be set up on SOME rank is dropped on EVERY rank
            state = None
            self.torch.cuda.empty_cache()
            return None, problem or "setup failed on another rank"
        return state, ""

    def time_alone(self, cover, state):
        sg = self.graphs[cover]
        return sg.time_exchange(state, repeats=1) * 1e3, sg.time_compute(state, self.a, repeats=1) * 1e3

    def early_options(self, cover):
        sg = self.graphs[cover]
        if self.args.early_pull == "auto" and sg.n_send_push_max > 0 and sg.n_send_pull_max > 0:
            return [False, True]
        return [self.args.early_pull == "on"]

    def run_step(self, cover, state, early):
        """Slowest rank's time of one K-iteration step, barrier + synchronize on both sides."""
        torch = self.torch
        self.dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.graphs[cover].propagate(state, self.a, self.K, early_pull=early)
        torch.cuda.synchronize()
        return self.slowest(time.perf_counter() - t0) * 1e3

    def release(self, state):
        del state
        self.torch.cuda.empty_cache()


def cover_order(args, pv):
    """Plan labels in the order they are built and timed: "cover" (fewest rows on the link), "pull" (the classic halo), then the
    weighted covers "cover@w" (fewer / shorter partial sums for more rows; under emulated link time they never won: last)."""
    weighted = [f"cover@{float(w):g}" for w in args.push_weights.split(",") if w.strip() and float(w) > 0]
    covers = ["cover", "pull"] + weighted if args.cover == "auto" else [args.cover]
    return covers[:1] if pv == 1 else covers


def setup(args, device, dist, backend, world, rank, pv, pf, deadline):
    """Generates the graph, hands every rank its block, builds the first plan, selects the variant.  Returns a namespace with the
    timed ``step`` and everything the record needs."""
    import argparse
    import torch
    from gnntf import rmat
    K, C, a = args.iterations, args.feats, args.alpha
    # (gloo rehearsals: an 8 GB broadcast staged through the host is the slow part there; every rank generates the same list instead)
    note(f"generating the graph ({args.nodes} vertices, {args.entries} entries) and handing every rank its block")
    idx, vals, bounds, comm, (gv, gf, pv, pf), t_gen = rmat.rmat_block_entries(args.nodes, args.entries, seed=1, device=device, grid=(pv, pf),
                                                                               replicate=backend == "gloo" and world > 1)
    note(f"entries of this rank's block: {idx.shape[0]} ({t_gen:.1f} s)")
    br.PHASES["startup_and_process_group"] = round(time.time() - br.T_START - t_gen, 2)
    br.PHASES["generate_and_broadcast"] = round(t_gen, 2)
    C_local = C // pf                                                       # this rank's feature slice
    gen = torch.Generator(device=device).manual_seed(2 + rank)
    covers = cover_order(args, pv)
    chunk_options = [k for k in (1, 2, 4) if k <= max(C_local // 32, 1)] if args.chunks <= 0 else [args.chunks]
    # most promising first (two chunks overlap exchange and SpMM at the least extra launches), so that a selection cut short by
    # its wall-clock budget (--select-seconds) has timed the likely winners
    chunk_order = [k for k in (2, 4, 1) if k in chunk_options] or chunk_options
    ops = BlockOps(args, dist, device, idx, vals, bounds, comm, None, chunk_options, a, K)
    ops.build_plan(covers[0], fatal=True)
    sg = ops.graphs[covers[0]]
    n_local, nnz_local, nnz_global = sg.n_local, sg.nnz_local, sg.nnz_global
    note(f"vertex blocks built ({covers[0]}): {pv} x {pf} grid, {n_local} rows / {nnz_local} entries on rank 0, {ops.plan_s[covers[0]]} s")
    H0 = torch.rand(n_local, C_local, device=device, generator=gen) * 2 - 1
    ops.H0 = H0
    variants, skipped = [], []
    selecting = sg.world > 1 and (len(covers) > 1 or len(chunk_options) > 1 or args.early_pull == "auto")
    budget = None
    if selecting:
        # what the selection may take: --select-seconds, and never more than --max-seconds leaves after a reserve for the timed region
        left = ops.slowest(-deadline.left())                         # (MAX of the negated = the LEAST time left on any rank)
        budget = max(0.0, min(float(args.select_seconds), -left - 60.0))
        ops.start()
        best, variants, skipped = select_variant(covers, chunk_order, ops, budget, K)
        if best is None:
            ops.agree(False, "no halo variant could be set up", fatal=True)
        br.PHASES["variant_selection"] = round(ops.spent(), 2)
    else:
        best = dict(cover=covers[0], chunks=chunk_options[-1] if args.chunks <= 0 else args.chunks, early_pull=args.early_pull == "on")
    del idx, vals
    ops.idx = ops.vals = None
    ops.keep_only(best["cover"])
    sg = ops.graphs[best["cover"]]
    state, problem = ops.make_state(best["cover"], best["chunks"])
    if state is None:
        ops.agree(False, "setting up the chosen variant: " + problem, fatal=True)
    halo = sg.halo_stats()
    halo.update(chunks=best["chunks"], early_pull=best["early_pull"], variants_timed_before_the_run=variants,
                variants_skipped=skipped, select_seconds_budget=budget,
                overlap_probe=getattr(sg.comm, "overlap_probe", None), overlap_probe_status=getattr(sg.comm, "overlap_status", None),
                chosen=dict(best), plan=best["cover"], pull_rows_sent=sg.n_send_pull_max, push_rows_sent=sg.n_send_push_max)
    prep = dict(gen_s=round(t_gen, 2), prep_s=round(sum(ops.plan_s.values()), 2), plans_built=list(ops.plan_s), plan_s=ops.plan_s)
    return argparse.Namespace(sg=sg, state=state, H0=H0, halo=halo, prep=prep, ops=ops, best=best, pv=pv, pf=pf, C_local=C_local,
                              n_local=n_local, nnz_local=nnz_local, nnz_global=nnz_global,
                              step=lambda: sg.propagate(state, a, K, early_pull=best["early_pull"]))


def line_halo(halo):
    """The halo block of the stdout line: plan sizes, the chosen variant, the bare exchange / kernels; of the variant table only
    (cover, chunks, early_pull, step_ms) -- the full table is in the detail file."""
    if not halo:
        return None
    out = {k: v for k, v in halo.items() if not isinstance(v, (dict, list))}
    out["chosen"] = halo.get("chosen")
    table = halo.get("variants_timed_before_the_run") or []
    out["halo_variants"] = [[v["cover"], v["chunks"], v["early_pull"], br.sig(v["step_ms"], 4) if v["step_ms"] else None] for v in table][:18]
    out["n_variants_skipped"] = len(halo.get("variants_skipped") or [])
    return out


def feature_slices(args, device, dist, world, barrier, deadline, ops):
    """N > 1, second field (never the headline): the SAME graph replicated on every rank, each rank propagating C / N of the feature
    columns -- no exchange at all, graph memory and prep grow with N.  Tells how far the vertex blocks are from a link-free bound.
    It must never take the headline down with it: every rank reports whether its setup worked, and the timed part (which holds
    collectives) runs only if it did everywhere."""
    import torch
    from bench_device import build_single, timed_steps
    from gnntf import _native as nat
    K, C, a = args.iterations, args.feats, args.alpha
    problem = ""
    try:
        g2, adj2, _ = build_single(args, device)
        Cs = C // world
        H2 = torch.rand(g2.n_rows, Cs, device=device) * 2 - 1
        out2, work2 = torch.empty_like(H2), torch.empty_like(H2)
    except Exception as error:                      # e.g. not enough memory for the whole graph beside what is still held
        problem = repr(error)[:300]
    if not ops.agree(not problem, "feature slices"):
        note("feature slices: skipped (" + (problem or "setup failed on another rank") + ")")
        return {"grid": f"1_vertex_block_x_{world}_feature_slices", "value": None, "error": problem or "setup failed on another rank"}

    def step2():
        nat.check(nat.lib().gnx_appnp_propagate(g2.handle, nat.ptr(adj2.vals), None, nat.ptr(H2), a, K, Cs, nat.ptr(out2), nat.ptr(work2),
                                                nat.current_stream()))
    steps2 = max(2, args.steps // 4)
    e2, _ = timed_steps(step2, steps2, 1, barrier)
    t2 = ops.slowest(e2)
    note(f"feature slices: {t2 / steps2 * 1e3:.1f} ms per step")
    return {"grid": f"1_vertex_block_x_{world}_feature_slices", "value": g2.nnz * K * steps2 / t2, "unit": "edges/s",
            "ms_per_step": t2 / steps2 * 1e3, "columns_per_rank": Cs, "kernel": g2.last_kernel(),
            "note": "graph replicated on every rank (memory and prep x N), no data-path communication; beside the headline, never instead"}
