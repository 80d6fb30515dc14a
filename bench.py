#!/usr/bin/env python3
"""bench.py -- propagated edges/s of the APPNP K=10 propagation on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one K-iteration propagation H <- (1-a) A_hat H + a H0 (K = 10; reference filter.py:17-22,34-35) over the resident
synthetic graph, through the C ABI of libgnx.so.  The workload is the SAME global graph for every N (strong scaling):
BASELINE.json configs[4] -- RMAT 80M vertices / 1B stored entries, 128 float32 features -- on pv = N contiguous vertex blocks with
a pairwise RCCL exchange of pulled rows / pushed partial sums per iteration (bench_sharded.py, gnntf.sharded).  N = 1 is that graph
on one GPU (it fits: ~131 GB) and additionally times, in the same run, BASELINE.json configs[3] (the roofline run, RMAT 10M / 100M,
C = 256), other widths, config 3's arxiv-shaped GCN forward and training steps (bench_secondary.py).  ``--workload config4`` makes
the roofline run the primary line instead.

Rank 0 prints ONE JSON line of at most 12 KB (bench_record.fit_line): the contract keys, ``config``, ``roofline`` (headline keys +
flat config4_* / per-width / train_* scalars) and ``cpu_baseline``.  Everything else -- full records with their notes, the
secondaries, yardsticks, the variant table -- goes to bench_detail_n<N>.json next to this script (``config.detail_file``).
ms_per_step / value are the MEDIAN of the per-step event times; the wall clock of the bracketed region is
reported beside them (wall_ms_per_step).

Files: bench_record.py (byte models, roofline record, the line), bench_device.py (graph, timing), bench_pmc.py (in-run counter
passes), bench_secondary.py, bench_sharded.py (N > 1), bench_cpu.py (cpu_baseline).
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]

import bench_record as br                                                                              # noqa: E402
from bench_record import (HBM_PEAK_GBS, WORKLOADS, alg_bytes_dropped_iteration, alg_bytes_per_iteration, min_bytes_dropped_iteration,   # noqa: E402,F401
                          min_bytes_per_iteration, note, phase, roofline_record, workload_name)
from bench_device import build_single, kept_entries, median_ms, stream_copy_GBs, stream_read_GBs, timed_steps   # noqa: E402,F401
from bench_pmc import fabric_bytes_by_segment, fabric_bytes_per_launch, measure_traffic_in_run, pmc_segments_child   # noqa: E402,F401

PUSH_WEIGHTS_DEFAULT = "0.5"        # weighted covers the N > 1 selection times after "cover" and "pull" (under EMULATED
                                    # link time w = 0 wins at every rate; one intermediate weight is timed anyway, inside the selection's budget)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["custom"], default="config5")
    ap.add_argument("--nodes", type=int, default=None, help="vertices of the GLOBAL graph (implies --workload custom)")
    ap.add_argument("--entries", type=int, default=None, help="stored directed entries of the GLOBAL graph")
    ap.add_argument("--feats", type=int, default=None)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--grid", type=str, default="", help="PVxPF process grid (vertex blocks x feature slices); default Nx1")
    ap.add_argument("--cover", default="auto",
                    help="halo plan: cover (pull/push vertex cover), cover@W (weighted cover, push weight W), pull (plain halo), or auto = one step "
                         "of each is timed before the run and the fastest kept")
    ap.add_argument("--push-weights", type=str, default=PUSH_WEIGHTS_DEFAULT,
                    help="--cover auto: weighted covers timed after the plain one and the pull plan, comma separated; empty = none")
    ap.add_argument("--chunks", type=int, default=0, help="column chunks whose exchange and SpMM overlap (0 = auto: 1, 2 and 4 are tried)")
    ap.add_argument("--early-pull", choices=["auto", "on", "off"], default="auto",
                    help="send the pulled rows ahead of the pushed partial sums (two messages per peer); auto = tried both ways")
    ap.add_argument("--select-seconds", type=float, default=120.0,
                    help="N > 1: wall-clock budget of the variant selection before the timed region, enforced before every plan, set-up and step "
                         "(bench_sharded.select_variant); what does not fit is recorded as skipped")
    ap.add_argument("--max-seconds", type=float, default=480.0,
                    help="what the whole run should take: optional parts (the second field, secondaries, yardsticks, the CPU baseline's length) are "
                         "dropped or shortened when behind, and recorded in config.dropped")
    ap.add_argument("--overlap-probe", choices=["on", "off"], default="on",
                    help="N > 1: probe where the exchange runs beside the compute stream (Comm.tune_overlap; GNX_TUNE_OVERLAP=0 also disables it)")
    ap.add_argument("--pmc-in-run", choices=["on", "off"], default="on",
                    help="N = 1: before anything else, run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (one step each, child "
                         "processes) so that roofline.traffic is measured in this run; off / failure: the committed profiles/pmc_traffic.json")
    ap.add_argument("--gather-yardstick", choices=["on", "off"], default="on",
                    help="N = 1: time the same kernel on a d-regular random graph of the same N and C (no reuse: B_alg is DRAM traffic there)")
    ap.add_argument("--pmc-child", choices=["", "segments"], default="", help=argparse.SUPPRESS)    # run by measure_traffic_in_run under rocprofv3
    ap.add_argument("--whole-rows", action="store_true", help="do not split interior / boundary rows")
    ap.add_argument("--force-sharded", action="store_true", help="run the vertex-partitioned path even with one rank (rehearsal)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads of the N = 1 line")
    ap.add_argument("--no-alt-grid", action="store_true", help="N > 1: skip the second field (the same graph on feature slices, no exchange)")
    args = ap.parse_args(argv)
    kind, _, weight = args.cover.partition("@")
    if kind not in ("auto", "cover", "pull") or (weight and (kind != "cover" or float(weight) < 0)):
        ap.error("--cover: auto, cover, cover@W (W >= 0) or pull")
    if args.nodes or args.entries or args.feats:
        base = WORKLOADS[args.workload if args.workload in WORKLOADS else "config5"]
        args.nodes, args.entries, args.feats = args.nodes or base[0], args.entries or base[1], args.feats or base[2]
        args.workload = "custom"
    else:
        args.nodes, args.entries, args.feats = WORKLOADS[args.workload]
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (nothing in this
    process has touched the GPU yet) and relay its one JSON line.  A failed child is a failure: nothing is retried."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(res.stdout)
    raise SystemExit(res.returncode)


def pmc_passes(args):
    """The in-run counter passes (N = 1, before this process touches the GPU; never from inside a profiler run)."""
    if not (args.gpus == 1 and args.pmc_in_run == "on" and not args.pmc_child and not args.force_sharded and args.workload in WORKLOADS
            and os.environ.get("GNX_BENCH_PMC", "1") != "0" and not any(k.startswith("ROCPROF") for k in os.environ)
            and "rocprof" not in os.environ.get("LD_PRELOAD", "")):
        return None
    # the headline workload as its own command; everything on the config-4 graph (the roofline run at C = 256, the other widths,
    # the training launches) in ONE further process cut into segments by a marker kernel
    wanted = [args.workload] + (["segments"] if not args.no_secondary else [])
    note("counter passes of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, child processes): " + ", ".join(wanted))
    t_ph = time.time()
    notes = measure_traffic_in_run(wanted, seconds=min(240.0, 0.4 * args.max_seconds), K=args.iterations)
    phase("pmc_passes_in_run", t_ph)
    note(f"counter passes: {notes}")
    return notes


def single_gpu_setup(args, device):
    """N = 1.  The timed step goes through the API the north star names: gnntf.APPNP (filter.py:25-35) with the reference's own layer
    list [Dropout, Dense, K x PPRIteration], its K propagation layers executed by the container's loop in eval mode exactly as
    architecture.predict() executes them (trainable.py:26-29 -> layered.py:52-55; Layered.run continues that loop from a value the
    caller holds).  H0 stands for the pre-MLP's output: it is planted as the value of the Dense layer the
    iterations read."""
    import torch
    import gnntf
    K, C, a = args.iterations, args.feats, args.alpha
    g, adj, prep = build_single(args, device)
    br.PHASES["startup"] = round(time.time() - br.T_START - prep["gen_s"] - prep["prep_s"] - br.PHASES.get("pmc_passes_in_run", 0.0), 2)
    br.PHASES.update(generate=prep["gen_s"], prep=prep["prep_s"])
    note(f"graph built: {g.n_rows} rows / {g.nnz} entries, prep {prep}")
    gen = torch.Generator(device=device).manual_seed(2)
    H0 = torch.rand(g.n_rows, C, device=device, generator=gen) * 2 - 1           # U(-1, 1), seed 2
    gnntf.set_seed(0)
    model = gnntf.APPNP(g, torch.zeros(g.n_rows, 1, device=device), num_classes=C, latent_dims=[], iterations=K, a=a)
    model.training_mode(False)
    first = len(model.layers()) - K
    iters, pre = model.layers()[first:], model.layers()[first - 1]
    assert all(type(l) is gnntf.PPRIteration for l in iters) and len(model.layers()) == 2 + K
    pre.value = H0
    api = {"timed_call": "architecture.run(H0, first=2): the K PPRIteration layers of gnntf.APPNP (filter.py:34-35), eval, no_grad",
           "layers": "Dropout, Dense, %d x PPRIteration" % K, "n_layers": len(model.layers())}
    adj = model.get_adjacency(0.5)                                              # the cached eval-mode adjacency the layers use (A2 once)

    def step():
        iters[-1].value = None                                                  # (the previous result: 41 GB at config 5)
        with torch.no_grad():
            model.run(H0, first=first)
    return argparse.Namespace(g=g, adj=adj, prep=prep, H0=H0, model=model, first=first, iters=iters, pre=pre, api=api, step=step)


def single_gpu_checks(args, device, s, step_ms):
    """After the timed region: the C entry on the same H0 (what the layer call must cost, and bit for bit what it must return), then
    the fixed point of the propagation through the timed path."""
    import torch
    from gnntf import _native as nat
    from gnntf.sharded import max_relative_deviation
    lib = nat.lib()
    K, C, a = args.iterations, args.feats, args.alpha
    out, work = torch.empty_like(s.H0), torch.empty_like(s.H0)
    direct = lambda: nat.check(lib.gnx_appnp_propagate(s.g.handle, nat.ptr(s.adj.vals), None, nat.ptr(s.H0), a, K, C, nat.ptr(out), nat.ptr(work),
                                                       nat.current_stream()))
    s.api["gnx_appnp_propagate_ms_per_step"] = median_ms(direct, reps=3, warm=1)
    s.api["layers_ms_per_step"] = br.step_statistics(step_ms)["median"]
    s.api["bitwise_equal_to_c_entry"] = bool(torch.equal(s.iters[-1].value, out))
    s.iters[-1].value = None
    del out, work, direct
    torch.cuda.empty_cache()
    deg = torch.empty(s.g.n_rows, dtype=torch.float32, device=device)
    nat.check(lib.gnx_graph_colsum(s.g.handle, 0.0, 0, 0, nat.ptr(deg), nat.current_stream()))
    E0 = deg.sqrt()[:, None] * (1.0 + torch.arange(C, dtype=torch.float32, device=device) / C)[None, :]
    del deg
    s.pre.value = E0
    with torch.no_grad():
        s.model.run(E0, first=s.first)
    err = max_relative_deviation(s.iters[-1].value, E0)
    s.iters[-1].value, s.pre.value = None, s.H0
    return err


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    br.start_heartbeat()
    deadline = br.Deadline(args.max_seconds)
    pmc_notes = pmc_passes(args) if world == 1 else None
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)")
    import torch
    # stdout carries exactly ONE JSON line: anything native libraries print there (RCCL's version banner)
    # is sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the propagation path has no CPU fallback")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import gnntf
    gnntf.set_default_device(device)
    if args.pmc_child == "segments":
        pmc_segments_child(args, device)
        return
    K, C, a = args.iterations, args.feats, args.alpha
    sharded_path = world > 1 or args.force_sharded
    pv, pf = world, 1
    if args.grid:
        pv, pf = (int(x) for x in args.grid.lower().split("x"))
        if pv * pf != world or C % pf != 0:
            raise SystemExit(f"bench.py: --grid {args.grid} needs PV * PF == {world} ranks and PF dividing the {C} features")
    dist = None
    if sharded_path:
        import torch.distributed as dist
        import bench_sharded
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("GNX_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of several ranks on ONE card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        s = bench_sharded.setup(args, device, dist, backend, world, rank, pv, pf, deadline)
        pv, pf = s.pv, s.pf
        n_local, nnz_local, nnz_global, C_local, halo, prep = s.n_local, s.nnz_local, s.nnz_global, s.C_local, s.halo, s.prep
    else:
        s = single_gpu_setup(args, device)
        n_local, nnz_local, nnz_global, C_local, halo, prep = s.g.n_rows, s.g.nnz, s.g.nnz, C, None, s.prep

    def barrier():
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    def everyone(flag):
        """The same yes / no on every rank (rank 0's clock decides: the deadline lives there)."""
        if not sharded_path:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], device=device, dtype=torch.int32)
        dist.broadcast(t, 0)
        return bool(int(t.item()))

    note(f"timing {args.warmup} + {args.steps} steps")
    t_ph = time.time()
    elapsed, step_ms = timed_steps(s.step, args.steps, args.warmup, barrier)
    phase("warmup_and_timed_steps", t_ph)
    if sharded_path:                                 # MAX over ranks: of the bracketed wall clock and of every step's event time
        t = torch.tensor([elapsed] + step_ms, device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, step_ms = float(t[0].item()), [float(x) for x in t[1:].tolist()]
    stats = br.step_statistics(step_ms)
    note(f"steps done: median {stats['median']:.1f} ms per step (min {stats['min']:.1f}, max {stats['max']:.1f}; wall {elapsed / args.steps * 1e3:.1f})")
    if sharded_path and world > 1:                   # measured, per iteration: the bare exchange and the bare kernels (collective calls)
        t_ph = time.time()
        t_x, t_c = s.sg.time_exchange(s.state), s.sg.time_compute(s.state, a)
        phase("exchange_and_kernels_alone", t_ph)
        halo_bytes = halo["max_halo_rows"] * C_local * 4
        halo.update(exchange_ms_alone=t_x * 1e3, compute_ms_alone=t_c * 1e3, halo_bytes_per_rank_per_iteration=halo_bytes,
                    ingress_GBs_per_rank=halo_bytes / max(t_x, 1e-9) / 1e9,
                    GBs_per_link_and_direction=halo_bytes / max(t_x, 1e-9) / 1e9 / max(pv - 1, 1),
                    pull_only_bytes_per_rank_per_iteration=halo["max_pull_only_rows"] * C_local * 4)
        note(f"exchange alone {halo['exchange_ms_alone']:.2f} ms, kernels alone {halo['compute_ms_alone']:.2f} ms per iteration")

    # in-run parity evidence: sqrt(degree) x s is a fixed point of the propagation on a symmetric graph -- more iterations through
    # the very path that was timed (for N > 1: the plan, the kernels AND the RCCL exchange) must reproduce it.  Behind schedule
    # (--max-seconds) the N > 1 check runs 2 iterations instead of K: every exchange direction and kernel is still exercised.
    note("self check: fixed point of the propagation")
    t_ph = time.time()
    check_iterations = K
    if sharded_path:
        if not everyone(deadline.left() > 3.0 * stats["median"] * 1e-3 + 30.0):
            check_iterations = min(K, 2)
            deadline.dropped.append(f"self check shortened to {check_iterations} iterations")
        check_err = s.sg.fixed_point_error(s.state, a, check_iterations)
    else:
        check_err = single_gpu_checks(args, device, s, step_ms)
    self_check = {"what": "H0 = sqrt(degree) x s_c is a fixed point of H <- (1-a) A_hat H + a H0: max rel. deviation through the timed path",
                  "iterations": check_iterations, "max_rel_err": check_err, "ok": bool(check_err < 1e-4)}
    note(f"self check: {check_err:.2e}")
    phase("self_check", t_ph)

    alt, kernel_blocks = None, (s.sg.graph.last_kernel() if sharded_path else None)
    if world > 1 and not args.no_alt_grid and not args.grid and C % world == 0:
        if everyone(deadline.room(4.0 * prep["gen_s"] + 3.0 * stats["median"] * 1e-3 + 20.0, "second field (feature slices)")):
            note("second field: the whole graph on every rank, C / N columns each")
            t_ph = time.time()
            ops = s.ops
            ops.graphs.clear()
            s.sg = s.state = s.H0 = s.step = ops.H0 = None
            torch.cuda.empty_cache()
            alt = bench_sharded.feature_slices(args, device, dist, world, barrier, deadline, ops)
            phase("alt_grid_feature_slices", t_ph)

    if rank == 0:
        t_ph = time.time()
        measured_peak = stream_copy_GBs(device)
        br.MEASURED_READ_PEAK[0] = stream_read_GBs(device)
        phase("stream_yardsticks", t_ph)
        name = workload_name(args.nodes, args.entries, C)
        if not sharded_path:
            launch_s = stats["median"] / 1e3 / K                     # one fused SpMM+mix launch (+ its long-row tail)
            roof = roofline_record(n_local, nnz_local, C, launch_s, K, name, measured_peak)
        else:                                                       # this rank's block: kernels alone (no exchange beside them)
            t_c = halo["compute_ms_alone"] * 1e-3 if world > 1 else s.sg.time_compute(s.state, a)
            # (the block's committed PMC passes are per plan: tools/sim_blocks.py --pmc-iterations under rocprofv3, profiles/summarize_blocks.py)
            block_name = name + f"_block_of_{pv}_{halo['plan']}_chunks{halo['chunks']}"
            roof = roofline_record(n_local, nnz_local, C_local, t_c, K, block_name, measured_peak,
                                   what="rank 0's vertex block, one iteration's kernels alone (pack + SpMM of every column chunk; no exchange beside them)")
        detail = {"roofline": roof, "halo": halo, "pmc_in_run": pmc_notes, "alt_grid_feature_slices": alt, "step_ms": step_ms}

        def optional(what, fn, default=None):
            """What follows the timed region must never cost the headline its line: a part that fails is recorded and left out."""
            try:
                return fn()
            except Exception as error:
                deadline.dropped.append(f"{what} failed: {error!r}"[:200])
                note(f"{what} failed and is left out: {error!r}")
                torch.cuda.empty_cache()
                return default
        cpu = None
        if not sharded_path and args.cpu_seconds > 0:
            from bench_cpu import cpu_baseline
            if deadline.left() < 4.0 * args.cpu_seconds + 120.0:    # behind schedule: a shorter sample rather than none (the contract asks for it)
                args.cpu_seconds = max(3.0, min(args.cpu_seconds, (deadline.left() - 120.0) / 4.0))
                deadline.dropped.append(f"cpu_baseline sample shortened to {args.cpu_seconds:.0f} s")
            note("CPU baselines (C / OpenMP port, scipy on one thread, torch.sparse on all threads; bounded samples)")
            t_ph = time.time()
            g_, adj_, H0_ = s.g, s.adj, s.H0
            cpu = optional("cpu_baseline", lambda: cpu_baseline(g_, adj_, H0_, args))
            del g_, adj_, H0_
            phase("cpu_baseline", t_ph)
        detail["cpu_baseline"] = cpu
        flat = {}
        if not sharded_path:
            kernel = s.g.last_kernel()
            for layer in s.model.layers():
                layer.value = None
            api = s.api
            s = None
            torch.cuda.empty_cache()
            if args.gather_yardstick == "on" and deadline.room(15.0, "no-reuse gather yardstick of the headline"):
                from bench_secondary import gather_yardstick
                note("no-reuse gather yardstick (d-regular random graph of the same N and C)")
                t_ph = time.time()
                yard = optional("gather yardstick", lambda: gather_yardstick(device, args.nodes, [C], a)[C])
                br.add_gather_ceiling(roof, yard)
                detail["no_reuse_gather_yardstick"] = yard
                phase("gather_yardstick", t_ph)
            if not args.no_secondary and deadline.room(40.0, "secondary workloads"):
                from bench_secondary import secondary_workloads
                t_ph = time.time()
                detail["secondary"], flat = optional("secondary workloads", lambda: secondary_workloads(
                    args, device, measured_peak, skip_config4=args.workload == "config4", deadline=deadline), default=({}, {}))
                phase("secondary_workloads", t_ph)
        else:
            kernel, api = kernel_blocks, None
        br.PHASES["total"] = round(time.time() - br.T_START, 2)
        edges_per_step = nnz_global * K
        result = {
            "metric": f"propagated edges/sec (APPNP K={K})", "value": edges_per_step / (stats["median"] * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": stats["median"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "step_ms_min": stats["min"], "step_ms_max": stats["max"], "step_ms_mean": stats["mean"],
            "wall_ms_per_step": elapsed / args.steps * 1e3, "timing": "median of the per-step event times (max over ranks per step)",
            "config": {"workload": f"{name}_appnp_K{K}" + ("" if args.workload == "custom" else f" (BASELINE {args.workload})"),
                       "global_rows": args.nodes, "stored_entries_total": nnz_global, "rows_per_rank": n_local,
                       "stored_entries_per_rank": nnz_local, "features": C, "iterations": K, "alpha": a,
                       "partition": (f"{pv}_vertex_blocks_x_{pf}_feature_slices" if sharded_path else "none"),
                       "halo": (bench_sharded.line_halo(halo) if sharded_path else None),
                       "prep": {k: prep[k] for k in ("gen_s", "prep_s")}, "kernel": kernel, "api": api,
                       "alt_grid_feature_slices": alt, "self_check": self_check, "phases_s": br.PHASES,
                       "pmc_in_run": ({k: ("in run" if v == "measured in this run" else v) for k, v in pmc_notes.items()} if pmc_notes else None),
                       "dropped": deadline.dropped, "detail_file": None},
            "roofline": dict(br.line_roofline(roof), **flat),
            "cpu_baseline": br.line_cpu_baseline(cpu),
        }
        detail["line"] = result
        result["config"]["detail_file"] = br.write_detail(detail, world)
        os.write(json_fd, (br.fit_line(result) + "\n").encode())
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()
    if not self_check["ok"]:                       # (check_err is all-reduced: every rank leaves the same way)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
