#!/usr/bin/env python3
"""bench.py -- propagated edges/s of the APPNP K=10 propagation on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one K-iteration propagation H <- (1-a) A_hat H + a H0 (K = 10) over the resident synthetic
graph, through the C ABI of libgnx.so.  The workload is the SAME global graph for every N (strong scaling):
BASELINE.json configs[4] -- RMAT 80M vertices / 1B stored entries, 128 float32 features -- on pv = N
contiguous vertex blocks with a pairwise RCCL exchange of pulled rows / pushed partial sums per iteration
(gnntf.sharded).  N = 1 is that graph on one GPU (it fits: ~131 GB) and additionally carries, in a
``secondary`` block timed in the same run, BASELINE.json configs[3] (the roofline run, RMAT 10M / 100M,
C = 256), other widths, config 3's arxiv-shaped GCN forward and a training-mode step.
``--workload config4`` makes the roofline run the primary line instead.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PUSH_WEIGHTS_DEFAULT = "0.5"        # weighted covers the N > 1 selection times beside "cover" and "pull" (under EMULATED link time
                                    # w = 0 wins at every rate; one intermediate weight is timed anyway, inside the selection's wall-clock budget,
                                    # because what real xGMI links do beside the kernels is exactly what no one-GPU rehearsal can show)
WORKLOADS = {"config5": (80_000_000, 1_000_000_000, 128),       # BASELINE.json configs[4]: the scaling graph (default)
             "config4": (10_000_000, 100_000_000, 256)}          # BASELINE.json configs[3]: the roofline run


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
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
                    help="--cover auto: weighted covers timed beside the plain one (cover_push_mask's push_weight: fewer entries summed on the sender's "
                         "side for more rows on the link), comma separated; empty = none")
    ap.add_argument("--chunks", type=int, default=0, help="column chunks whose exchange and SpMM overlap (0 = auto: 1, 2 and 4 are tried)")
    ap.add_argument("--early-pull", choices=["auto", "on", "off"], default="auto",
                    help="send the pulled rows ahead of the pushed partial sums (two messages per peer); auto = tried both ways")
    ap.add_argument("--select-seconds", type=float, default=120.0,
                    help="N > 1: wall-clock budget of the variant selection before the timed region; what does not fit is recorded as skipped")
    ap.add_argument("--overlap-probe", choices=["on", "off"], default="on",
                    help="N > 1: probe where the exchange runs beside the compute stream (Comm.tune_overlap; GNX_TUNE_OVERLAP=0 also disables it)")
    ap.add_argument("--pmc-in-run", choices=["on", "off"], default="on",
                    help="N = 1: before anything else, run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (one step each, child "
                         "processes) so that roofline.traffic is measured in this run; off / failure: the committed profiles/pmc_traffic.json")
    ap.add_argument("--gather-yardstick", choices=["on", "off"], default="on",
                    help="N = 1: time the same kernel on a d-regular random graph of the same N and C (no reuse: B_alg is DRAM traffic there) and "
                         "report it beside the roofline fraction")
    ap.add_argument("--pmc-child", choices=["", "segments"], default="", help=argparse.SUPPRESS)    # run by measure_traffic_in_run under rocprofv3
    ap.add_argument("--whole-rows", action="store_true", help="do not split interior / boundary rows")
    ap.add_argument("--force-sharded", action="store_true", help="run the vertex-partitioned path even with one rank (rehearsal)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads of the N = 1 line")
    ap.add_argument("--no-alt-grid", action="store_true", help="N > 1: skip the second field (the same graph on feature slices, no exchange)")
    args = ap.parse_args()
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


def alg_bytes_per_iteration(n, nnz, C):
    """nnz*(4 col + 4 val + 4C gathered row) + N*(4 rowptr + 4C H0 + 4C out)."""
    return nnz * (8 + 4 * C) + n * (4 + 8 * C)


def min_bytes_per_iteration(n, nnz, C):
    """compulsory bytes, every array touched once: 8 nnz + 4 N + 12 N C."""
    return 8 * nnz + 4 * n + 12 * n * C


def alg_bytes_dropped_iteration(n, nnz, kept, C, backward=False):
    """One TRAINING iteration (layered.py:47-50 + gnn.py:37-42 + filter.py:19-21) in the convention of alg_bytes_per_iteration:
    every stored entry's column index and RAW value are read (the draw needs them), only the ``kept`` entries gather a neighbour
    row.  Forward (gnx_spmm_dropped_chained, k >= 1): + per row rowptr, D[row], the next iteration's scale, H0 and out.
    Backward (gnx_spmm_dropped_back, a middle iteration, over the transposed structure): + per row rowptr, D[row], the next step's
    scale, the running gradient sum read and written, and the pre-scaled operand of the next step written."""
    if backward:
        return nnz * 8 + kept * 4 * C + n * (4 + 4 + 4 + 12 * C)
    return nnz * 8 + kept * 4 * C + n * (4 + 4 + 4 + 8 * C)


def min_bytes_dropped_iteration(n, nnz, C, backward=False):
    """Compulsory bytes of one TRAINING iteration, every array touched once (the convention of min_bytes_per_iteration): col + raw
    value of every stored entry, rowptr + D[row] + the next scale per row, the operand read once, H0 (forward) / the running sum
    read and written (backward), the result written."""
    return 8 * nnz + n * (4 + 4 + 4) + (16 if backward else 12) * n * C


def kept_entries(g, p, seed, first_stream, n_streams):
    """Stored entries that survive the edge dropout of each of ``n_streams`` consecutive dropout streams (counted from the
    materialised values, gnx_graph_normalize: a dropped entry is an explicit zero there)."""
    import gnntf
    return [int((gnntf.normalize(g, "symmetric", "none", dropout=p, seed=seed, stream_id=first_stream + k).vals != 0).sum())
            for k in range(n_streams)]


def workload_name(n, nnz, C):
    return f"rmat_n{n}_nnz{nnz}_C{C}"


# ---- synthetic graph ---------------------------------------------------------
def build_single(args, device):
    """The whole graph on one GPU: R-MAT pairs -> symmetrised unsorted COO -> device CSR (A0) -> normalise once (A2)."""
    import torch
    import gnntf
    from gnntf import sharded
    n, m = args.nodes, args.entries // 2
    t0 = time.time()
    u, v = sharded.rmat_relabelled_pairs(n, m, seed=1, device=device)
    idx = torch.cat([torch.stack([u, v], 1), torch.stack([v, u], 1)])      # symmetrised COO, unsorted
    del u, v
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, n)), device=device)      # A0: COO -> CSR on the device
    del idx, vals
    adj = gnntf.normalize(g, "symmetric")                                          # A2, once (eval mode)
    torch.cuda.synchronize()
    t_prep = time.time() - t0
    torch.cuda.empty_cache()
    return g, adj, dict(gen_s=round(t_gen, 2), prep_s=round(t_prep, 2))


def timed_steps(step, steps, warmup, barrier):
    """W untimed steps, then exactly K timed ones bracketed by barrier + synchronize; also per-step events on the
    launch stream.  Returns (wall seconds, [ms per step])."""
    import torch
    for _ in range(warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s, e in ev:                     # events sit on the stream the kernels are launched on
        s.record()
        step()
        e.record()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [s.elapsed_time(e) for s, e in ev]


def stream_read_GBs(device, nbytes=8 << 30, reps=5):
    """Device read-only streaming rate measured in this run (the SpMM is almost all reads)."""
    import torch
    from gnntf import _native as nat
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    sink = torch.zeros(64, dtype=torch.float32, device=device)
    read = lambda: nat.check(nat.lib().gnx_stream_read(nat.ptr(src), src.numel(), nat.ptr(sink), nat.current_stream()))
    read()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        read()
    e.record()
    torch.cuda.synchronize()
    return 1.0 * nbytes * reps / (s.elapsed_time(e) * 1e-3) / 1e9


def stream_copy_GBs(device, nbytes=4 << 30, reps=5):
    """Device stream-copy rate (read + write bytes per second) measured in this run: the achievable HBM peak."""
    import torch
    from gnntf import _native as nat
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    copy = lambda: nat.check(nat.lib().gnx_stream_copy(nat.ptr(src), nat.ptr(dst), src.numel(), nat.current_stream()))
    copy()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        copy()
    e.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (s.elapsed_time(e) * 1e-3) / 1e9


IN_RUN_TRAFFIC = {}                # workload name -> (bytes per launch, source): counter passes made by THIS run (measure_traffic_in_run)


def pmc_traffic(name):
    """Fabric (L2-miss) bytes per launch of this workload: from the rocprofv3 --pmc passes this very run made before its timed
    region when there are any (measure_traffic_in_run), else from the committed builder-run passes (profiles/pmc_traffic.json), else None."""
    if name in IN_RUN_TRAFFIC:
        return IN_RUN_TRAFFIC[name]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path)).get("workloads", {}).get(name)
    if not rec:
        return None, None
    return float(rec["fabric_bytes_per_launch"]), f"profiles/pmc_traffic.json [{rec.get('source', '')}] -- builder-run rocprofv3 --pmc passes of this command, NOT measured in this run"


def fabric_bytes_per_launch(fetch_csv, write_csv):
    """Bytes leaving the L2s per propagation launch from the counter_collection CSVs of a FETCH_SIZE and a WRITE_SIZE pass of one
    bench command (what profiles/summarize.py computes for the committed files): both counters are in KiB; on gfx950 FETCH_SIZE
    counts the 128-byte requests of wide coalesced reads as 64 bytes, so the read side is doubled (MI355X_MICROARCH.md, "HBM");
    one launch = one dispatch of every SpMM kernel, the row kernel of a very large graph being dealt in pieces."""
    import collections
    import csv
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in (fetch_csv, write_csv):
        for r in csv.DictReader(open(path)):
            if "k_spmm" in r["Kernel_Name"]:
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counts = [len(c["FETCH_SIZE"]) for c in per.values() if c["FETCH_SIZE"]]
    if not counts:
        return None
    launches = min(counts)
    total = 0.0
    for c in per.values():
        fetch = sum(c["FETCH_SIZE"]) / launches
        write = sum(c["WRITE_SIZE"]) * (len(c["FETCH_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)) / launches if c["WRITE_SIZE"] else 0.0
        total += 2 * fetch * 1024 + write * 1024
    return total


SEGMENT_WIDTHS = (256, 128, 64, 40, 8, 7)       # widths propagated on the config-4 graph in the segments pass (256 = the roofline run; 40 / 7 =
                                                  # the widths gnntf's own APPNP propagates on arxiv / Cora: filter.py:33-35, width = num_classes)
TRAIN_WIDTH, TRAIN_LAUNCHES = 64, 3               # the training launches of the segments pass


def segment_plan(K):
    """[(traffic-table name, launches)] of the measured segments of `bench.py --pmc-child segments`, in order."""
    n4, e4, _ = WORKLOADS["config4"]
    wl = workload_name(n4, e4, TRAIN_WIDTH)
    return [(workload_name(n4, e4, C), K) for C in SEGMENT_WIDTHS] + [("train_forward_" + wl, TRAIN_LAUNCHES), ("train_backward_" + wl, TRAIN_LAUNCHES)]


def fabric_bytes_by_segment(fetch_csv, write_csv, marker="k_stream"):
    """Bytes leaving the L2s per SEGMENT of a run that a marker kernel cuts into pieces (same corrections as fabric_bytes_per_launch:
    KiB, reads doubled on gfx950; only the SpMM kernels are counted).  Returns {segment number: bytes}, segment s = the dispatches
    between the s-th marker and the next one (0 = before the first marker)."""
    import collections
    import csv
    total = collections.defaultdict(float)
    for path, counter, factor in ((fetch_csv, "FETCH_SIZE", 2048.0), (write_csv, "WRITE_SIZE", 1024.0)):
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
        seg, marks = 0, set()
        for r in rows:
            if marker in r["Kernel_Name"]:
                if r["Dispatch_Id"] not in marks:                     # (one row per counter and dispatch)
                    marks.add(r["Dispatch_Id"])
                    seg += 1
            elif r["Counter_Name"] == counter and "k_spmm" in r["Kernel_Name"]:
                total[seg] += factor * float(r["Counter_Value"])
    return dict(total)


def pmc_segments_child(args, device):
    """`bench.py --pmc-child segments` (run by measure_traffic_in_run under rocprofv3 --pmc, never by hand for a result): ONE
    process on the config-4 graph in which a marker kernel (k_stream: a 64-float gnx_stream_read) brackets each measured piece --
    the K-iteration propagation at every width of SEGMENT_WIDTHS, then TRAIN_LAUNCHES forward and backward training launches at
    TRAIN_WIDTH -- each after an unmeasured warm-up of its own (lazily built handle parts).  Measured piece i is segment 2 i + 1."""
    import torch
    import gnntf
    from gnntf import _native as nat
    from gnntf import sparse as sp
    lib = nat.lib()
    K, a = args.iterations, args.alpha
    n4, e4, _ = WORKLOADS["config4"]
    g, adj, _ = build_single(argparse.Namespace(nodes=n4, entries=e4), device)
    n = g.n_rows
    mark_src, mark_sink = torch.zeros(64, device=device), torch.zeros(64, device=device)

    def bracket(fn, launches=1):
        fn()                                                                   # warm-up, outside the measured segment
        torch.cuda.synchronize()
        nat.check(lib.gnx_stream_read(nat.ptr(mark_src), 64, nat.ptr(mark_sink), nat.current_stream()))
        for _ in range(launches):
            fn()
        nat.check(lib.gnx_stream_read(nat.ptr(mark_src), 64, nat.ptr(mark_sink), nat.current_stream()))
        torch.cuda.synchronize()

    for C in SEGMENT_WIDTHS:
        gen = torch.Generator(device=device).manual_seed(2)
        H0 = torch.rand(n, C, device=device, generator=gen) * 2 - 1
        res, work = torch.empty_like(H0), torch.empty_like(H0)
        bracket(lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(res), nat.ptr(work),
                                                          nat.current_stream())))
        del H0, res, work
    C = TRAIN_WIDTH
    X = torch.rand(n, C, device=device) * 2 - 1
    gout = torch.rand(n, C, device=device)
    scales = sp.dropped_degree_scales(g, 0.5, 1, 0, K)
    adj1 = sp.dropped_adjacency(g, 0.5, 1, 1, D=scales[1])
    with torch.no_grad():
        bracket(lambda: sp._launch_chained(adj1, X, X, 1.0 - a, a, True, scales[2], skip_empty=True), TRAIN_LAUNCHES)
        S_run, Y_run = torch.zeros_like(gout), torch.empty_like(gout)
        bracket(lambda: sp._launch_back(adj1, gout, True, scales[0], S_run, 1.0, a * (1.0 - a), S_run, 1.0 - a, Y_run, skip_empty=True), TRAIN_LAUNCHES)
    torch.cuda.synchronize()


def measure_traffic_in_run(workloads, seconds=240.0, K=10):
    """rocprofv3 --pmc passes of THIS bench command, made by this process before it touches the GPU (child processes: the program
    after `--` is the interpreter itself): FETCH_SIZE and WRITE_SIZE, one pass each, per entry of ``workloads`` -- "config5" /
    "config4": one propagation step of the same timed call on the same box; "segments": the config-4 graph at every width of
    SEGMENT_WIDTHS plus the training launches, one process cut into segments by a marker kernel (pmc_segments_child).  Fills
    IN_RUN_TRAFFIC, so that every roofline record's traffic is measured in the driver's own run rather than read from committed
    files; whatever fails (no rocprofv3, no counter access, time) leaves the committed entries in charge and says so in the
    returned notes."""
    import glob
    import shutil
    import tempfile
    notes = {}
    if shutil.which("rocprofv3") is None:
        return {w: "rocprofv3 not on PATH" for w in workloads}
    t_start = time.time()
    for w in workloads:
        tmp = tempfile.mkdtemp(prefix="gnx_pmc_", dir="/tmp")
        csvs, problem = {}, None
        if w == "segments":
            child_args = ["--pmc-child", "segments", "--iterations", str(K)]
        else:
            child_args = ["--workload", w, "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-secondary", "--pmc-in-run", "off", "--gather-yardstick", "off"]
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            if time.time() - t_start > seconds:
                problem = "time budget spent"
                break
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "run", "--", "python3", os.path.abspath(__file__)] + child_args
            # the pass runs in a process group of its own, so that a pass that outlives its time limit can be ended WHOLE (profiler
            # and the python under it): a survivor would keep tens of GB of the card this process is about to use
            err_path = os.path.join(tmp, ctr + ".err")
            try:
                with open(err_path, "w") as errlog:
                    child = subprocess.Popen(cmd, cwd=tmp, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=errlog,
                                             start_new_session=True)
                    try:
                        rc = child.wait(timeout=max(30.0, seconds - (time.time() - t_start)))
                    except subprocess.TimeoutExpired:
                        import signal
                        os.killpg(child.pid, signal.SIGKILL)              # (its own session: pgid == pid of the process started here)
                        child.wait()
                        problem = f"pass {ctr} exceeded its time limit and was ended"
                if problem is not None:
                    break
            except Exception as error:
                problem = repr(error)[:200]
                break
            found = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if rc != 0 or not found:
                problem = f"pass {ctr} failed (rc {rc}): " + open(err_path).read()[-200:]
                break
            csvs[ctr] = max(found, key=os.path.getmtime)
        how = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py " + " ".join(child_args) + "`, run by THIS bench process on " \
              "this box before its timed region (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes, per launch)"
        if problem is None and w == "segments":
            try:
                by_segment = fabric_bytes_by_segment(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"])
                for i, (name, launches) in enumerate(segment_plan(K)):
                    if by_segment.get(2 * i + 1):
                        IN_RUN_TRAFFIC[name] = (by_segment[2 * i + 1] / launches, how)
                missing = [name for name, _ in segment_plan(K) if name not in IN_RUN_TRAFFIC]
                notes[w] = "measured in this run" if not missing else "measured in this run except " + ", ".join(missing)
            except Exception as error:
                problem = repr(error)[:200]
        elif problem is None:
            n, e, C = WORKLOADS[w]
            try:
                total = fabric_bytes_per_launch(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"])
            except Exception as error:
                total, problem = None, repr(error)[:200]
            if total:
                IN_RUN_TRAFFIC[workload_name(n, e, C)] = (total, how)
                notes[w] = "measured in this run"
            elif problem is None:
                problem = "no SpMM dispatch in the counter files"
        if problem is not None:
            notes[w] = "not measured in this run (" + problem + "): the committed entries of profiles/pmc_traffic.json are used"
        shutil.rmtree(tmp, ignore_errors=True)
    notes["seconds"] = round(time.time() - t_start, 1)
    return notes


MEASURED_READ_PEAK = [None]        # in-run read-only streaming rate (set once by main)


FRAC_LEVEL = "fabric: bytes leaving the L2s incl. Infinity-Cache hits -- NOT DRAM bandwidth (see dram_frac_*, no_reuse_gather_*)"


def roofline_record(n, nnz, C, launch_s, K, name, measured_peak, b_alg=None, b_min=None, what=None):
    """achieved = min(B_alg, B_rocprof) / t per launch; B_alg, B_min, B_rocprof and t printed together.
    ``frac`` is a FABRIC-level figure (``frac_level``): the counters see what leaves the L2s, and the Infinity Cache serves part of
    it.  What can be said about DRAM itself is carried beside it: ``dram_frac_lower_bound`` (every array touched once: B_min / t /
    peak) and ``dram_frac_upper_bound`` (DRAM cannot have moved more than the fabric did, nor faster than this box streams reads
    in this run) -- the fabric figure is never below either -- and, once add_gather_ceiling has run, the rate of the same kernel
    on a graph WITHOUT reuse, where fabric bytes are DRAM bytes.
    Without a PMC entry for ``name`` the min() cannot be taken: ``achieved`` / ``frac`` are null (``min_rule_applied`` false) and
    ``frac_bound_without_counters`` = min(B_alg / t, in-run read stream) / peak is all that is printed."""
    if b_alg is None:
        b_alg, b_min = alg_bytes_per_iteration(n, nnz, C), min_bytes_per_iteration(n, nnz, C)
    traffic, source = pmc_traffic(name)
    read_peak = MEASURED_READ_PEAK[0]
    stale = False
    if traffic:
        achieved = min(b_alg, traffic) / launch_s / 1e9
        frac = achieved / HBM_PEAK_GBS
        if frac > 1.0:            # more bytes per second than the memory system moves: the committed counters cannot belong to this launch
            achieved, frac, stale = None, None, True
    else:
        achieved = frac = None
    read_frac = (read_peak if read_peak else HBM_PEAK_GBS) / HBM_PEAK_GBS
    compulsory = (b_min / launch_s / 1e9 / HBM_PEAK_GBS) if b_min else None
    rec = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "frac_level": FRAC_LEVEL,
           "traffic": traffic, "traffic_source": source, "traffic_in_run": bool(source) and "THIS bench process" in source,
           "min_rule_applied": bool(traffic) and not stale, "traffic_entry_inconsistent_with_this_run": stale,
           "frac_bound_without_counters": None if frac is not None else min(b_alg / launch_s / 1e9 / HBM_PEAK_GBS, read_frac, 1.0),
           "alg_bytes_per_launch": b_alg, "min_bytes_per_launch": b_min,
           "frac_compulsory": compulsory, "dram_frac_lower_bound": compulsory,
           "dram_frac_upper_bound": min(frac, read_frac) if frac is not None else None,
           "launch_ms": launch_s * 1e3, "t_prop_ms": launch_s * 1e3 * K,
           "measured_peak": measured_peak, "measured_read_peak": read_peak,
           "fabric_rate_over_read_stream": (achieved / read_peak) if (read_peak and achieved) else None,
           "no_reuse_gather_GBs": None, "no_reuse_gather_frac": None, "frac_of_gather_ceiling": None,
           "note": (what or "one launch = one fused SpMM+mix iteration incl. its long-row kernels") + "; achieved = min(B_alg, traffic) / launch time; "
                   "traffic = FETCH_SIZE x2 + WRITE_SIZE (separate --pmc passes): bytes leaving the L2s, a fabric-level figure (Infinity-Cache hits "
                   "counted; fabric_rate_over_read_stream > 1 shows them); DRAM moved between dram_frac_lower_bound (B_min / t / peak) and "
                   "dram_frac_upper_bound (min(frac, in-run read stream / peak)); no_reuse_gather_*: the same kernel on a d-regular random graph "
                   "of the same N and C, where nothing is reused and B_alg IS the DRAM traffic; measured_peak = in-run stream copy, "
                   "measured_read_peak = in-run read-only stream"}
    return rec


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
        out[C] = {"GBs": alg_bytes_per_iteration(n, g.nnz, C) / ms / 1e6, "launch_ms": ms, "rows": n, "entries": g.nnz, "d": d, "C": C,
                  "kernel": g.last_kernel()}
        del H, H0, res
    del g, adj
    torch.cuda.empty_cache()
    return out


def add_gather_ceiling(rec, yard):
    """Puts the in-run no-reuse yardstick beside a roofline record: the DRAM-level rate of the kernel where it can be measured
    (no_reuse_gather_frac = that / peak), and the record's own rate relative to it (above 1 = reuse served on-die)."""
    if yard:
        rec["no_reuse_gather_GBs"] = yard["GBs"]
        rec["no_reuse_gather_frac"] = yard["GBs"] / HBM_PEAK_GBS
        rec["no_reuse_gather_launch_ms"] = yard["launch_ms"]
        rec["no_reuse_gather_entries"] = yard["entries"]
        if rec.get("achieved"):
            rec["frac_of_gather_ceiling"] = rec["achieved"] / yard["GBs"]
    return rec


FLAT_KEYS = ("ms_per_step", "launch_ms", "frac", "achieved", "traffic", "traffic_in_run", "alg_bytes_per_launch", "min_bytes_per_launch",
             "frac_compulsory", "dram_frac_upper_bound", "no_reuse_gather_frac", "frac_of_gather_ceiling", "edges_per_s")


def flat_keys(prefix, rec, **extra):
    """A secondary roofline record as FLAT scalar keys (``<prefix>_frac``, ``<prefix>_traffic`` ...) for the primary line's
    ``roofline`` object: the driver's record keeps flat scalars of that object and drops nested ones, and the line's tail is
    truncated -- so the roofline run (config 4) and the narrow widths must be recomputable from these keys alone."""
    both = dict(rec, **extra)
    return {f"{prefix}_{k}": both[k] for k in FLAT_KEYS if both.get(k) is not None}


# ---- CPU baselines on a bounded sample: SURVEY.md 8(d) (i) scipy, one thread; (ii) torch.sparse.mm, all threads; (iii) C / OpenMP port ----
def cpu_baseline(g, adj, H0, args):
    """One of the K iterations over a row prefix of the SAME workload, three ways (every cost is linear in the entries walked, the
    gathers span all of H).  ``value`` is the strongest of the three that does what the reference does -- the C / OpenMP port with
    the per-iteration renormalisation (gnn.py:36-50 called from filter.py:18): whole-graph normalisation timed once over all
    entries, SpMM + mix on the sample and scaled to all entries, value = nnz / (t_norm + t_spmm * nnz / e)."""
    import numpy as np
    import torch
    import __graft_entry__ as ge
    from oracle import cpu_baselines as cb
    lib = ctypes.CDLL(ge.build_oracle())
    sig = [ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
    lib.oracle_sample_iteration_par.restype = ctypes.c_int
    lib.oracle_sample_iteration_par.argtypes = sig
    lib.oracle_num_threads.restype = ctypes.c_int
    rowptr, colidx, vals = (t.cpu().numpy() for t in g.csr_arrays())
    nvals = adj.vals.cpu().numpy()                                                # A_hat's values (device normalisation)
    H = H0.cpu().numpy()
    n, C = H.shape
    nnz = int(rowptr[-1])
    budget = max(args.cpu_seconds, 3.0)

    def run(rows, renorm=1):
        out = np.empty((max(rows, 1), C), dtype=np.float32)
        given = vals if renorm else nvals                                         # renorm = 0: the values are A_hat's already
        t0 = time.time()
        rc = lib.oracle_sample_iteration_par(n, rows, rowptr.ctypes.data, colidx.ctypes.data, given.ctypes.data, H.ctypes.data,
                                             H.ctypes.data, args.alpha, C, out.ctypes.data, renorm)
        assert rc == 0
        return time.time() - t0, int(rowptr[rows]), out

    def rows_for(rate, seconds, floor):
        """Row prefix whose entries take about ``seconds`` at ``rate`` entries/s."""
        want = int(min(nnz, max(floor, seconds * rate)))
        return int(min(n, max(1, np.searchsorted(rowptr, want, side="left"))))

    # (iii) the oracle's C / OpenMP port, all host threads
    probe_rows = rows_for(1.0, 0.0, min(nnz, 2_000_000))
    t_probe, e_probe, _ = run(probe_rows, renorm=0)
    t_norm, _, _ = run(0)                                                         # the whole-graph renormalisation alone
    rows = rows_for(e_probe / max(t_probe, 1e-3), 0.4 * budget, e_probe)
    t_only, e, ref_out = run(rows, renorm=0)
    port = {"value": nnz / (t_norm + t_only * nnz / e), "unit": "edges/s", "cores": int(lib.oracle_num_threads()), "kind": "port",
            "spmm_only_value": e / t_only,
            "sample": f"C / OpenMP port of the oracle (oracle/propagate_ref.c), all host threads: 1 of {args.iterations} iterations; the whole-graph "
                      f"renormalisation the reference does in every iteration (gnn.py:36-50 called from filter.py:18) timed over all {nnz} "
                      f"entries ({t_norm:.2f} s), SpMM + mix over the first {rows} of {n} rows ({e} entries, C={C}: {t_only:.2f} s) and scaled "
                      f"to all entries: value = nnz / (t_norm + t_spmm * nnz / e); spmm_only_value = e / t_spmm with the adjacency "
                      f"normalised beforehand (what the GPU figure times); CPU restatement of gnntf's TF-CPU path (TensorFlow unavailable)"}
    # (i) scipy CSR on ONE thread, re-normalising inside the iteration (sample-sized: its cost is linear in the entries too)
    rows1 = rows_for(2e6, 0.0, min(nnz, 1_000_000))
    _, tn, ts = cb.scipy_iteration(rowptr, colidx, vals, H, H, args.alpha, rows1)
    e1 = int(rowptr[rows1])
    rows1 = rows_for(e1 / max(tn + ts, 1e-3), 0.35 * budget, e1)
    _, tn, ts = cb.scipy_iteration(rowptr, colidx, vals, H, H, args.alpha, rows1)
    e1 = int(rowptr[rows1])
    port["scipy_single_thread"] = {"value": e1 / (tn + ts), "unit": "edges/s", "cores": 1, "kind": "port", "spmm_only_value": e1 / ts,
                                   "sample": f"scipy.sparse CSR @ dense, float32, one thread (oracle/cpu_baselines.py): first {rows1} rows ({e1} entries): "
                                             f"column sums + divide_no_nan + two value scalings of those entries {tn:.2f} s, SpMM + mix {ts:.2f} s"}
    # (ii) torch.sparse.mm on all host threads, adjacency normalised beforehand
    rows2 = rows_for(e / t_only / 4, 0.25 * budget, min(nnz, 2_000_000))
    out2, t2, used = cb.torch_sparse_iteration(rowptr, colidx, nvals, H, H, args.alpha, rows2)
    e2 = int(rowptr[rows2])
    agree = None
    if rows2 <= rows:                                                            # same rows, same values: the two restatements must agree
        agree = float(np.abs(out2 - ref_out[:rows2]).max())
    port["torch_sparse_all_threads"] = {"value": e2 / t2, "unit": "edges/s", "cores": int(used), "kind": "port",
                                        "sample": f"torch.sparse.mm (CPU, COO, {used} threads) + mix over the first {rows2} rows ({e2} entries), adjacency "
                                                  f"normalised beforehand: {t2:.2f} s; max |difference| to the C port on those rows: {agree}"}
    port["host"] = {"os_cpu_count": os.cpu_count(), "torch_threads": int(torch.get_num_threads())}
    return port


# ---- secondary workloads (N = 1): timed in this same run so that they are driver-timed too ----------------
def median_ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def secondary_workloads(args, device, measured_peak, skip_config4=False):
    import torch
    import gnntf
    from gnntf import _native as nat
    lib = nat.lib()
    out = {}
    K, a = args.iterations, args.alpha
    n4, e4, c4 = WORKLOADS["config4"]
    g, adj, prep = build_single(argparse.Namespace(nodes=n4, entries=e4), device)
    n, nnz = g.n_rows, g.nnz
    widths = []
    flat = {}                                                    # flat scalar copies for the primary line's roofline object (flat_keys)
    yards = {}
    if args.gather_yardstick == "on":
        yards = gather_yardstick(device, n4, [w for w in SEGMENT_WIDTHS if not (skip_config4 and w == c4)], a)
        out["config4_no_reuse_gather_yardstick"] = [yards[w] for w in sorted(yards, reverse=True)]
    for C in ([] if skip_config4 else [c4]) + [w for w in SEGMENT_WIDTHS if w != c4]:
        gen = torch.Generator(device=device).manual_seed(2)
        H0 = torch.rand(n, C, device=device, generator=gen) * 2 - 1
        res, work = torch.empty_like(H0), torch.empty_like(H0)
        ms = median_ms(lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(res),
                                                                 nat.ptr(work), nat.current_stream())), reps=3, warm=1)
        roof = roofline_record(n, nnz, C, ms * 1e-3 / K, K, workload_name(n4, e4, C), measured_peak)
        rec = {"C": C, "kernel": g.last_kernel(), "ms_per_step": ms, "edges_per_s": nnz * K / ms * 1e3, "roofline": roof}
        add_gather_ceiling(roof, yards.get(C))
        if C == c4:
            out["config4_roofline_run"] = dict(rec, workload=workload_name(n4, e4, C) + f"_appnp_K{K}", prep=prep)
            flat.update(flat_keys("config4", roof, ms_per_step=ms, edges_per_s=rec["edges_per_s"]))
            flat.update(config4_workload=workload_name(n4, e4, C) + f"_appnp_K{K}", config4_rows=n, config4_entries=nnz, config4_kernel=rec["kernel"])
        else:
            widths.append(rec)
            flat.update(flat_keys(f"config4_graph_C{C}", roof, ms_per_step=ms, edges_per_s=rec["edges_per_s"]))
        del H0, res, work
    out["config4_graph_other_widths"] = widths
    # the same propagation through the API the north star names: architecture.predict() of gnntf.APPNP (filter.py:25-35 ->
    # trainable.py:26-29) on the config-4 graph.  APPNP builds filter.py:30-35's own list [Dropout, Dense(F -> C), K x PPRIteration]; the
    # container executes the K layers as one fused run, which must cost what gnx_appnp_propagate costs and return the same bits
    via_api = []
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
            t_direct = median_ms(lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(res),
                                                                           nat.ptr(work), nat.current_stream())), reps=3, warm=1)
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
        gnntf.set_seed(0)
        hand = gnntf.GNN(g, X)
        H0l = hand.add(gnntf.Dense(C, regularize=False))
        for _ in range(K):
            hand.add(gnntf.PPRIteration(H0l, a))
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
        via_api[-1]["hand_built_stack"] = {"layers": [type(l).__name__ for l in hand.layers()][:3] + ["..."], "forward_ms": t_hand,
                                           "dense_alone_ms": t_dense, "propagation_ms": t_hand - t_dense,
                                           "layer_by_layer_forward_ms": t_hand_layers,
                                           "max_abs_difference_to_layer_by_layer": float((fused_out - by_layer).abs().max()),
                                           "what": f"GNN(graph, X) + Dense({C}) + {K} x PPRIteration(H0, {a}) added by hand, eval mode: forward_ms with the "
                                                   f"container fusing the run (propagation_ms = forward - the Dense alone: to be compared with "
                                                   f"propagation_layers_ms), layer_by_layer_forward_ms with fuse_runs = False"}
        del hand, H0l, X, fused_out, by_layer
        torch.cuda.empty_cache()
    out["config4_via_layer_api"] = via_api
    # a graph WITH communities (planted partition x power-law degrees, same N, ~ the same entries: the structure the reference's
    # citation datasets have and R-MAT lacks): gnntf.APPNP in its default order against GNN(reorder="locality") -- label propagation
    # order + row windows (gnx_graph_set_row_window) -- at the widths gnntf's APPNP propagates; same model, same weights, the
    # outputs compared in the caller's order; prep = what the reordered model's construction costs beyond the plain one's
    from gnntf.rmat import community_pairs
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
            if C == 40:       # the same K layers in TRAINING mode (per-iteration edge dropout + renormalisation), forward + backward
                Ht = H0c.detach().clone().requires_grad_()
                gout_c = torch.rand_like(Ht)
                cm.layers()[1].value = Ht

                def train_once():
                    Ht.grad = None
                    cm.run(Ht, first=2).backward(gout_c)
                cm.training_mode(True)
                per[key + "_training_step_ms"] = median_ms(train_once, reps=3, warm=1)
                cm.training_mode(False)
                del Ht, gout_c
            if reorder:
                per["reorder_used"], per["locality_share"], per["entries"] = cm.reorder_used, cm.locality_share, cm.graph.nnz
            for layer in cm.layers():
                layer.value = None
            del cm, H0c
            torch.cuda.empty_cache()
        per["max_abs_difference_of_the_outputs"] = float((outs[None] - outs["locality"]).abs().max())
        per["argmax_equal_share"] = float((outs[None].argmax(1) == outs["locality"].argmax(1)).float().mean())
        per["bitwise_equal_share"] = float((outs[None] == outs["locality"]).float().mean())
        per["mean_abs_output"] = float(outs[None].abs().mean())
        per["time_ratio"] = per["locality_ms"] / per["default_ms"]
        comm_rec["widths"].append(per)
        if C == 40:
            flat.update(community_graph_C40_training_step_default_ms=per["default_training_step_ms"],
                        community_graph_C40_training_step_locality_order_ms=per["locality_training_step_ms"])
        flat.update({f"community_graph_C{C}_default_ms": per["default_ms"], f"community_graph_C{C}_locality_order_ms": per["locality_ms"],
                     f"community_graph_C{C}_locality_prep_s": per["locality_build_s"] - per["default_build_s"]})
        del outs
    flat.update(community_graph_locality_share=comm_rec["widths"][0]["locality_share"], community_graph_entries=comm_rec["widths"][0]["entries"])
    out["community_graph_locality_order"] = comm_rec
    del cidx, ccoo, Xc
    torch.cuda.empty_cache()
    # training-mode step: K = 10 iterations, each with its own dropped + re-normalised adjacency,
    # forward + backward through the fused loop node (masks regenerated in the backward), C = 64
    C = 64
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
    ms2 = median_ms(lambda: train_step(two_pass), reps=3, warm=1)
    # the two launches the step consists of, each timed alone with events and priced against its own byte model (one forward
    # iteration k >= 1 = gnx_spmm_dropped_chained; one backward iteration = gnx_spmm_dropped on the transposed structure)
    from gnntf import sparse as sp
    kept = kept_entries(g, 0.5, 1, 1, 1)[0]
    scales = sp.dropped_degree_scales(g, 0.5, 1, 0, K)
    adj1 = sp.dropped_adjacency(g, 0.5, 1, 1, D=scales[1])
    Xd = H0.detach()
    with torch.no_grad():
        ms_f = median_ms(lambda: sp._launch_chained(adj1, Xd, Xd, 1.0 - a, a, True, scales[2], skip_empty=True), reps=5, warm=2)
        kernel_f = g.last_kernel()
        S_run, Y_run = torch.zeros_like(gout), torch.empty_like(gout)
        ms_b = median_ms(lambda: sp._launch_back(adj1, gout, True, scales[0], S_run, 1.0, a * (1.0 - a), S_run, 1.0 - a, Y_run, skip_empty=True),
                         reps=5, warm=2)
        del S_run, Y_run
        ms_d = median_ms(lambda: sp.dropped_degree_scales(g, 0.5, 1, 0, K), reps=3, warm=1)
    wl = workload_name(n4, e4, C)
    roof_f = roofline_record(n, nnz, C, ms_f * 1e-3, K, "train_forward_" + wl, measured_peak,
                             b_alg=alg_bytes_dropped_iteration(n, nnz, kept, C), b_min=min_bytes_dropped_iteration(n, nnz, C),
                             what="one forward TRAINING iteration (gnx_spmm_dropped_chained, "
                             "a middle one: weights from the counter RNG inside the SpMM, only kept entries gathered, rows without entries left to the last "
                             "iteration) incl. its long-row kernels")
    roof_b = roofline_record(n, nnz, C, ms_b * 1e-3, K, "train_backward_" + wl, measured_peak,
                             b_alg=alg_bytes_dropped_iteration(n, nnz, kept, C, backward=True), b_min=min_bytes_dropped_iteration(n, nnz, C, backward=True),
                             what="one backward TRAINING iteration "
                             "(gnx_spmm_dropped_back over the transposed structure: the running gradient sum updated and the next step's "
                             "pre-scaled operand written in the epilogue) incl. its long-row kernels")
    flat.update(flat_keys("train_C64_forward", roof_f), **flat_keys("train_C64_backward", roof_b))
    flat.update(train_C64_step_ms=ms, train_C64_kept_entries=kept)
    out["training_step_C64"] = {"ms": ms, "two_pass_ms": ms2, "edges_per_s": 2 * nnz * K / ms * 1e3,
                                "forward_launch_ms": ms_f, "backward_launch_ms": ms_b, "degree_scales_all_streams_ms": ms_d,
                                "launches_share_of_step": (K * (ms_f + ms_b) + ms_d) / ms, "kept_entries": kept, "kernel": kernel_f,
                                "roofline": roof_f, "roofline_backward": roof_b,
                                "what": f"forward + backward of {K} PPR iterations with per-iteration edge dropout 0.5 + renormalisation, "
                                        f"config-4 graph, C=64; ms: weights produced inside the SpMM (gnx_spmm_dropped), two_pass_ms: "
                                        f"materialised per iteration (gnx_graph_normalize + gnx_spmm); roofline / roofline_backward: one "
                                        f"forward / backward iteration's launch timed alone, byte model alg_bytes_dropped_iteration (col + raw "
                                        f"value of EVERY entry, a neighbour row per KEPT entry, H0 + out + scales per row)"}
    del H0, gout, Xd, scales, adj1
    torch.cuda.empty_cache()
    # the matrix-core ends of the path at the config-4 size
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
    mf["dense_10M_x_256_to_64_relu"] = {"ms": t_dense, "torch_addmm_relu_ms": t_torch, "TFLOPs": 2.0 * n * 256 * 64 / t_dense / 1e9, "GBs": (n * 256 * 4 + n * 64 * 4) / t_dense / 1e6,
                                        "mfma_peak_TFLOPs": 157.3, "what": "gnx_dense (k_dense_wreg: W in registers, X through an LDS-DMA ring), float32 "
                                                                           "v_mfma_f32_16x16x4_f32; X read once from HBM"}
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
    out["matrix_core_kernels"] = mf
    del logits, nodes, labels, g, adj
    torch.cuda.empty_cache()
    # config 3: arxiv-shaped 2-layer GCN forward (N = 169,343; 1,166,243 undirected pairs -> 2,332,486 stored entries; 128 -> 64 -> 40)
    g3, adj3, _ = build_single(argparse.Namespace(nodes=169_343, entries=2_332_486), device)
    X = torch.randn(g3.n_rows, 128, device=device)
    model = gnntf.GCN(g3, X, num_classes=40)
    model.training_mode(False)
    with torch.no_grad():
        t_fwd = median_ms(lambda: model(model.features), reps=20, warm=5)
        X64 = torch.randn(g3.n_rows, 64, device=device)
        t128 = median_ms(lambda: gnntf.spmm(adj3, X), reps=20, warm=5)
        t64 = median_ms(lambda: gnntf.spmm(adj3, X64), reps=20, warm=5)
    # config 2: Cora-shaped APPNP (N = 2708, F = 1433 at 1.3 % density, C = 7, K = 10): the launch-latency regime -- ms per training
    # epoch of architecture.train(), eager and replayed from hipGraphs (train(capture=True)), and the eval forward
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
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
    out["config2_cora_shaped_appnp"] = cora
    out["config3_arxiv_shaped_gcn"] = {"nodes": g3.n_rows, "entries": g3.nnz, "forward_ms": t_fwd, "spmm128_ms": t128, "spmm64_ms": t64,
                                       "spmm128_edges_per_s": g3.nnz / t128 * 1e3, "spmm64_edges_per_s": g3.nnz / t64 * 1e3}
    out["flat"] = flat
    return out


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (nothing in this
    process has touched the GPU yet) and relay its one JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(res.stdout)
    raise SystemExit(res.returncode)


def note(msg):
    """Progress line on stderr (rank 0 only; stdout carries nothing but the JSON line)."""
    LAST_NOTE[0] = str(msg)
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - T_START, msg))
        sys.stderr.flush()


T_START = time.time()
LAST_NOTE = ["start"]


def start_heartbeat(every=60.0):
    """Rank 0 says it is alive once a minute (stderr): a long silent phase -- plan building at full size, a host-staged rehearsal
    step -- is otherwise indistinguishable from a hang for whoever watches the run."""
    import threading
    if int(os.environ.get("RANK", "0")) != 0:
        return

    def beat():
        while True:
            time.sleep(every)
            sys.stderr.write("[bench %7.1fs] ... still running (last: %s)\n" % (time.time() - T_START, LAST_NOTE[0][:120]))
            sys.stderr.flush()
    threading.Thread(target=beat, daemon=True).start()


PHASES = {}                        # seconds per phase of the run (rank 0's clock), printed in config.phases


def phase(name, t0):
    PHASES[name] = round(PHASES.get(name, 0.0) + time.time() - t0, 2)


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    start_heartbeat()
    pmc_notes = None
    if world == 1 and args.gpus == 1 and args.pmc_in_run == "on" and not args.pmc_child and not args.force_sharded and args.workload in WORKLOADS \
            and os.environ.get("GNX_BENCH_PMC", "1") != "0" and not any(k.startswith("ROCPROF") for k in os.environ) \
            and "rocprof" not in os.environ.get("LD_PRELOAD", ""):          # (never from inside a profiler run)
        # (nothing in this process has touched the GPU yet: the passes are child processes that come and go before it does)
        # the headline workload as its own command; everything on the config-4 graph (the roofline run at C = 256, the other widths,
        # the training launches) in ONE further process cut into segments by a marker kernel
        wanted = [args.workload] + (["segments"] if not args.no_secondary else [])
        note("counter passes of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, child processes): " + ", ".join(wanted))
        t_ph = time.time()
        pmc_notes = measure_traffic_in_run(wanted, K=args.iterations)
        phase("pmc_passes_in_run", t_ph)
        note(f"counter passes: {pmc_notes}")
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
    if sharded_path:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("GNX_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of several ranks on ONE card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if not sharded_path:
        g, adj, prep = build_single(args, device)
        PHASES["startup"] = round(time.time() - T_START - prep["gen_s"] - prep["prep_s"] - PHASES.get("pmc_passes_in_run", 0.0), 2)
        PHASES.update(generate=prep["gen_s"], prep=prep["prep_s"])
        n_local, nnz_local, nnz_global = g.n_rows, g.nnz, g.nnz
        note(f"graph built: {g.n_rows} rows / {g.nnz} entries, prep {prep}")
        gen = torch.Generator(device=device).manual_seed(2)
        H0 = torch.rand(n_local, C, device=device, generator=gen) * 2 - 1       # U(-1, 1), seed 2
        from gnntf import _native as nat
        lib = nat.lib()
        # The timed step goes through the API the north star names: gnntf.APPNP (filter.py:25-35) with the reference's own layer list
        # [Dropout, Dense, K x PPRIteration], its K propagation layers executed by the container's loop in eval mode exactly as
        # architecture.predict() executes them (trainable.py:26-29 -> layered.py:52-55; Layered.run continues that loop from a value
        # the caller holds).  H0 stands for the pre-MLP's output: it is planted as the value of the Dense layer
        # the iterations read.
        gnntf.set_seed(0)
        model = gnntf.APPNP(g, torch.zeros(n_local, 1, device=device), num_classes=C, latent_dims=[], iterations=K, a=a)
        model.training_mode(False)
        first = len(model.layers()) - K
        iters, pre = model.layers()[first:], model.layers()[first - 1]
        assert all(type(l) is gnntf.PPRIteration for l in iters) and len(model.layers()) == 2 + K
        pre.value = H0
        api = {"timed_call": "architecture.run(H0, first=2): the K PPRIteration layers of gnntf.APPNP(...) (filter.py:34-35) through the container's loop, "
                             "eval mode, torch.no_grad()",
               "layers": [type(l).__name__ for l in model.layers()], "n_layers": len(model.layers())}
        adj = model.get_adjacency(0.5)                                          # the cached eval-mode adjacency the layers use (A2 once)

        def step():
            iters[-1].value = None                                              # (the previous result: 41 GB at config 5)
            with torch.no_grad():
                model.run(H0, first=first)
        halo = None
        C_local = C
    else:
        from gnntf import rmat, sharded
        # (gloo rehearsals: an 8 GB broadcast staged through the host is the slow part there; every rank generates the same list instead)
        note(f"generating the graph ({args.nodes} vertices, {args.entries} entries) and handing every rank its block")
        idx, vals, bounds, comm, (gv, gf, pv, pf), t_gen = rmat.rmat_block_entries(args.nodes, args.entries, seed=1, device=device, grid=(pv, pf),
                                                                                   replicate=backend == "gloo" and world > 1)
        note(f"entries of this rank's block: {idx.shape[0]} ({t_gen:.1f} s)")
        PHASES["startup_and_process_group"] = round(time.time() - T_START - t_gen, 2)
        PHASES["generate_and_broadcast"] = round(t_gen, 2)
        C_local = C // pf                                                       # this rank's feature slice
        gen = torch.Generator(device=device).manual_seed(2 + rank)
        # Which halo plan / pipelining is fastest depends on what the links of THIS node sustain, which nothing on a one-GPU box
        # can tell: one step of every variant is timed before the timed region (max over ranks) and the timed steps run on the
        # fastest.  cover: pull/push vertex cover or the classic pull-only halo; chunks: column chunks whose exchange and SpMM
        # overlap; early_pull: the pulled rows leave as soon as they are gathered, ahead of the pushed partial sums.
        # plan labels: "cover" (fewest rows on the link), "cover@w" (weighted: fewer / shorter partial sums for more rows), "pull"
        weighted = [f"cover@{float(w):g}" for w in args.push_weights.split(",") if w.strip() and float(w) > 0]
        covers = ["cover"] + weighted + ["pull"] if args.cover == "auto" else [args.cover]
        if pv == 1:
            covers = covers[:1]                                                 # one vertex block: nothing is exchanged
        chunk_options = [k for k in (1, 2, 4) if k <= max(C_local // 32, 1)] if args.chunks <= 0 else [args.chunks]
        # most promising first (two chunks overlap exchange and SpMM at the least extra launches), so that a selection cut short by
        # its wall-clock budget (--select-seconds) has timed the likely winners
        chunk_order = [k for k in (2, 4, 1) if k in chunk_options] or chunk_options
        graphs, plan_s = {}, {}

        def build_plan(cover):
            t0 = time.time()
            kind, _, weight = cover.partition("@")
            graphs[cover] = sharded.ShardedGraph(idx, vals, bounds, comm=comm, cover=kind, chunks=chunk_options[-1], push_weight=float(weight or 0.0),
                                                 split_rows=not args.whole_rows, relabel=True, tune_overlap=args.overlap_probe == "on")
            torch.cuda.synchronize()
            plan_s[cover] = round(time.time() - t0, 2)

        build_plan(covers[0])
        sg = graphs[covers[0]]
        n_local, nnz_local, nnz_global = sg.n_local, sg.nnz_local, sg.nnz_global
        PHASES["plan_" + covers[0]] = plan_s[covers[0]]
        note(f"vertex blocks built ({covers[0]}): {pv} x {pf} grid, {n_local} rows / {nnz_local} entries on rank 0, {plan_s[covers[0]]} s")
        H0 = torch.rand(n_local, C_local, device=device, generator=gen) * 2 - 1

        def rank_max_ms(fn, reps=2):
            """Slowest rank's time of one call of ``fn`` (best of ``reps``), barrier + synchronize on both sides."""
            best = None
            for _ in range(reps):
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                best = float(t.item()) if best is None else min(best, float(t.item()))
            return best * 1e3

        t_select = time.perf_counter()

        def select_spent():
            """Seconds since the selection began on the SLOWEST rank: every rank sees the same number and takes the same branch."""
            t = torch.tensor([time.perf_counter() - t_select], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        variants, skipped = [], []
        selecting = sg.world > 1 and (len(covers) > 1 or len(chunk_options) > 1 or args.early_pull == "auto")
        if selecting:
            budget = float(args.select_seconds)
            for cover in covers:
                if cover not in graphs:
                    # a second plan costs what the first one did: it is built only if that still fits the budget
                    spent = select_spent()
                    if spent + plan_s[covers[0]] > budget:
                        skipped.append(dict(cover=cover, reason=f"plan not built: {spent:.1f} s of the {budget:.0f} s selection budget spent, a plan takes {plan_s[covers[0]]} s"))
                        note(f"selection budget: the {cover} plan is not built")
                        continue
                    build_plan(cover)
                    PHASES["plan_" + cover] = plan_s[cover]
                cand = graphs[cover]
                earlies = [False, True] if (args.early_pull == "auto" and cand.n_send_push_max > 0 and cand.n_send_pull_max > 0) \
                    else [args.early_pull == "on"]
                for chunks in chunk_order:
                    have_one = any(v["step_ms"] is not None for v in variants)
                    if have_one and select_spent() > budget:
                        skipped.append(dict(cover=cover, chunks=chunks, reason="selection budget spent"))
                        continue
                    # a variant that cannot be set up on SOME rank (memory) is dropped on EVERY rank: the decision is collective
                    state, problem = None, ""
                    try:
                        state = cand.make_state(H0, chunks=chunks)
                    except Exception as error:
                        problem = repr(error)[:200]
                    ok = torch.tensor([0 if problem else 1], device=device, dtype=torch.int32)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0:
                        variants.append(dict(cover=cover, chunks=chunks, early_pull=None, step_ms=None, error=problem or "setup failed on another rank"))
                        state = None
                        torch.cuda.empty_cache()
                        continue
                    alone = dict(exchange_ms_alone=cand.time_exchange(state, repeats=1) * 1e3, compute_ms_alone=cand.time_compute(state, a, repeats=1) * 1e3)
                    for early in earlies:
                        if any(v["step_ms"] is not None for v in variants) and select_spent() > budget:
                            skipped.append(dict(cover=cover, chunks=chunks, early_pull=early, reason="selection budget spent"))
                            continue
                        run = lambda: cand.propagate(state, a, K, early_pull=early)
                        run()                                                   # opens the connections / sizes the scratch of this variant
                        variants.append(dict(cover=cover, chunks=chunks, early_pull=early, step_ms=rank_max_ms(run), **alone))
                        note(f"variant {variants[-1]}")
                    del state
                    torch.cuda.empty_cache()
            timed = [v for v in variants if v["step_ms"] is not None]
            if not timed:
                raise SystemExit("bench.py: no halo variant could be set up")
            best = min(timed, key=lambda v: v["step_ms"])                        # the same numbers on every rank: the same choice
            PHASES["variant_selection"] = round(time.perf_counter() - t_select, 2)
        else:
            best = dict(cover=covers[0], chunks=chunk_options[-1] if args.chunks <= 0 else args.chunks, early_pull=args.early_pull == "on")
        del idx, vals
        torch.cuda.empty_cache()
        prep = dict(gen_s=round(t_gen, 2), prep_s=round(sum(plan_s.values()), 2), plans_built=list(graphs), plan_s=plan_s)
        sg = graphs[best["cover"]]
        for cover in list(graphs):
            if cover != best["cover"]:
                del graphs[cover]
        torch.cuda.empty_cache()
        state = sg.make_state(H0, chunks=best["chunks"])

        def step():
            sg.propagate(state, a, K, early_pull=best["early_pull"])
        halo = sg.halo_stats()
        halo.update(chunks=best["chunks"], early_pull=best["early_pull"], variants_timed_before_the_run=variants,
                    variants_skipped=skipped, select_seconds_budget=args.select_seconds if selecting else None,
                    overlap_probe=getattr(sg.comm, "overlap_probe", None), overlap_probe_status=getattr(sg.comm, "overlap_status", None),
                    chosen=dict(best), plan=best["cover"], pull_rows_sent=sg.n_send_pull_max, push_rows_sent=sg.n_send_push_max)

    def barrier():
        if sharded_path:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"timing {args.warmup} + {args.steps} steps")
    t_ph = time.time()
    elapsed, step_ms = timed_steps(step, args.steps, args.warmup, barrier)
    phase("warmup_and_timed_steps", t_ph)
    note(f"steps done: {elapsed / args.steps * 1e3:.1f} ms per step on this rank")
    if sharded_path:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if world > 1:                        # measured, per iteration: the bare exchange and the bare kernels (collective calls)
            t_ph = time.time()
            t_x, t_c = sg.time_exchange(state), sg.time_compute(state, a)
            phase("exchange_and_kernels_alone", t_ph)
            halo_bytes = halo["max_halo_rows"] * C_local * 4
            halo.update(exchange_ms_alone=t_x * 1e3, compute_ms_alone=t_c * 1e3, halo_bytes_per_rank_per_iteration=halo_bytes,
                        ingress_GBs_per_rank=halo_bytes / max(t_x, 1e-9) / 1e9,
                        GBs_per_link_and_direction=halo_bytes / max(t_x, 1e-9) / 1e9 / max(pv - 1, 1),
                        pull_only_bytes_per_rank_per_iteration=halo["max_pull_only_rows"] * C_local * 4)

    # in-run parity evidence: sqrt(degree) x s is a fixed point of the propagation on a symmetric graph -- K more iterations through
    # the very path that was timed (for N > 1: the plan, the kernels AND the RCCL exchange) must reproduce it
    note("self check: fixed point of the propagation")
    t_ph = time.time()
    if sharded_path:
        check_err = sg.fixed_point_error(state, a, K)
    else:
        # the C entry on the same H0: what the layer call must cost, and bit for bit what it must return
        out, work = torch.empty_like(H0), torch.empty_like(H0)
        direct = lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), a, K, C, nat.ptr(out), nat.ptr(work),
                                                           nat.current_stream()))
        api["gnx_appnp_propagate_ms_per_step"] = median_ms(direct, reps=3, warm=1)
        api["layers_ms_per_step"] = sum(step_ms) / len(step_ms)
        api["bitwise_equal_to_c_entry"] = bool(torch.equal(iters[-1].value, out))
        iters[-1].value = None
        del out, work, direct
        torch.cuda.empty_cache()
        deg = torch.empty(g.n_rows, dtype=torch.float32, device=device)
        nat.check(lib.gnx_graph_colsum(g.handle, 0.0, 0, 0, nat.ptr(deg), nat.current_stream()))
        E0 = deg.sqrt()[:, None] * (1.0 + torch.arange(C, dtype=torch.float32, device=device) / C)[None, :]
        del deg
        pre.value = E0
        with torch.no_grad():
            model.run(E0, first=first)
        from gnntf.sharded import max_relative_deviation
        check_err = max_relative_deviation(iters[-1].value, E0)
        iters[-1].value, pre.value = None, H0
        del E0
    self_check = {"what": "H0 = sqrt(degree) x s_c is a fixed point of H <- (1-a) A_hat H + a H0 on a symmetric graph: largest deviation "
                          "after K iterations through the timed path, relative to max(|H0|, 1), max over ranks",
                  "max_rel_err": check_err, "ok": bool(check_err < 1e-4)}
    note(f"self check: {check_err:.2e}")
    phase("self_check", t_ph)

    # N > 1, second field (never the headline): the SAME graph replicated on every rank, each rank propagating C / N of the feature
    # columns -- no exchange at all, graph memory and prep grow with N.  Tells how far the vertex blocks are from a link-free bound.
    alt = None
    if world > 1:
        note(f"exchange alone {halo['exchange_ms_alone']:.2f} ms, kernels alone {halo['compute_ms_alone']:.2f} ms per iteration")
    if world > 1 and not args.no_alt_grid and not args.grid and C % world == 0:
        note("second field: the whole graph on every rank, C / N columns each")
        t_ph = time.time()
        kernel_blocks = sg.graph.last_kernel()
        graphs.clear()
        del state, sg, H0
        torch.cuda.empty_cache()
        from gnntf import _native as nat
        # the second field must never take the headline down with it: every rank reports whether its setup worked, and the
        # timed part (which holds collectives) runs only if it did everywhere
        problem = ""
        try:
            g2, adj2, _ = build_single(args, device)
            Cs = C // world
            H2 = torch.rand(g2.n_rows, Cs, device=device) * 2 - 1
            out2, work2 = torch.empty_like(H2), torch.empty_like(H2)
        except Exception as error:                      # e.g. not enough memory for the whole graph beside what is still held
            problem = repr(error)[:300]
        ok = torch.tensor([0 if problem else 1], device=device, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            alt = {"grid": f"1_vertex_block_x_{world}_feature_slices", "value": None, "error": problem or "setup failed on another rank"}
            note("feature slices: skipped (" + alt["error"] + ")")
        else:
            def step2():
                nat.check(nat.lib().gnx_appnp_propagate(g2.handle, nat.ptr(adj2.vals), None, nat.ptr(H2), a, K, Cs, nat.ptr(out2), nat.ptr(work2),
                                                        nat.current_stream()))
            steps2 = max(2, args.steps // 4)
            e2, _ = timed_steps(step2, steps2, 1, barrier)
            t2 = torch.tensor([e2], device=device, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            alt = {"grid": f"1_vertex_block_x_{world}_feature_slices", "value": g2.nnz * K * steps2 / float(t2.item()), "unit": "edges/s",
                   "ms_per_step": float(t2.item()) / steps2 * 1e3, "columns_per_rank": Cs, "kernel": g2.last_kernel(),
                   "note": "graph replicated on every rank (memory and prep x N), no data-path communication; reported beside the headline "
                           "vertex-block grid, never instead of it"}
            del g2, adj2, H2, out2, work2
            note(f"feature slices: {alt['ms_per_step']:.1f} ms per step")
        phase("alt_grid_feature_slices", t_ph)
    else:
        kernel_blocks = sg.graph.last_kernel() if sharded_path else None

    if rank == 0:
        edges = nnz_global * K * args.steps
        t_ph = time.time()
        measured_peak = stream_copy_GBs(device)
        MEASURED_READ_PEAK[0] = stream_read_GBs(device)
        phase("stream_yardsticks", t_ph)
        name = workload_name(args.nodes, args.entries, C)
        if not sharded_path:
            launch_s = (sum(step_ms) / len(step_ms)) / 1e3 / K      # one fused SpMM+mix launch (+ its long-row tail)
            roof = roofline_record(n_local, nnz_local, C, launch_s, K, name, measured_peak)
        else:                                                       # this rank's block: kernels alone (no exchange beside them)
            t_c = halo["compute_ms_alone"] * 1e-3 if world > 1 else sg.time_compute(state, a)
            # (the block's committed PMC passes are per plan: tools/sim_blocks.py --pmc-iterations under rocprofv3, profiles/summarize_blocks.py)
            block_name = name + f"_block_of_{pv}_{halo['plan']}_chunks{halo['chunks']}"
            roof = roofline_record(n_local, nnz_local, C_local, t_c, K, block_name, measured_peak)
            roof["note"] = "rank 0's vertex block, one iteration's kernels alone (pack + SpMM of every column chunk; no exchange beside them); " + roof["note"]
        result = {
            "metric": f"propagated edges/sec (APPNP K={K})", "value": edges / elapsed, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}_appnp_K{K}" + ("" if args.workload == "custom" else f" (BASELINE {args.workload})"),
                       "global_rows": args.nodes, "stored_entries_total": nnz_global, "rows_per_rank": n_local,
                       "stored_entries_per_rank": nnz_local, "features": C, "iterations": K, "alpha": a,
                       "partition": (f"{pv}_vertex_blocks_x_{pf}_feature_slices" if sharded_path else "none"),
                       "halo": halo, "prep": prep, "kernel": (kernel_blocks if sharded_path else g.last_kernel()),
                       "api": (None if sharded_path else api),
                       "alt_grid_feature_slices": alt, "self_check": self_check, "phases_s": PHASES, "pmc_in_run": pmc_notes},
            "roofline": roof,
        }
        if not sharded_path and args.cpu_seconds > 0:
            note("CPU baselines (C / OpenMP port, scipy on one thread, torch.sparse on all threads; bounded samples)")
            t_ph = time.time()
            result["cpu_baseline"] = cpu_baseline(g, adj, H0, args)
            phase("cpu_baseline", t_ph)
        else:
            result["cpu_baseline"] = None
        if not sharded_path:
            for layer in model.layers():
                layer.value = None
            del g, adj, H0, model, iters, pre, layer
            torch.cuda.empty_cache()
            if args.gather_yardstick == "on":
                note("no-reuse gather yardstick (d-regular random graph of the same N and C)")
                t_ph = time.time()
                yard = gather_yardstick(device, args.nodes, [C], a)[C]
                add_gather_ceiling(roof, yard)
                result["config"]["no_reuse_gather_yardstick"] = yard
                phase("gather_yardstick", t_ph)
        if not sharded_path and not args.no_secondary:
            note("secondary workloads")
            t_ph = time.time()
            result["secondary"] = secondary_workloads(args, device, measured_peak, skip_config4=args.workload == "config4")
            roof.update(result["secondary"].pop("flat"))       # config 4, the other widths and the training launches as flat scalar keys
            phase("secondary_workloads", t_ph)
        PHASES["total"] = round(time.time() - T_START, 2)
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if sharded_path:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
