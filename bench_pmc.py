"""bench_pmc.py -- the in-run counter passes of bench.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this very command (child
processes started before the parent touches the GPU), and the reading of their counter files into fabric bytes per launch, with
the unit and gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md ("HBM": KiB; wide reads counted at half their size)."""
import argparse
import os
import subprocess
import time

import bench_record as br
from bench_record import WORKLOADS, workload_name

SEGMENT_WIDTHS = (256, 128, 64, 40, 8, 7)       # widths propagated on the config-4 graph in the segments pass (256 = the roofline run; 40 / 7 =
                                                  # the widths gnntf's own APPNP propagates on arxiv / Cora: filter.py:33-35, width = num_classes)
TRAIN_WIDTHS, TRAIN_LAUNCHES = (64, 40, 7), 3     # the training launches of the segments pass (40 / 7: the widths gnntf trains at, trainable.py:70-78)
TRAIN_WIDTH = TRAIN_WIDTHS[0]


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


def segment_plan(K):
    """[(traffic-table name, launches)] of the measured segments of `bench.py --pmc-child segments`, in order."""
    n4, e4, _ = WORKLOADS["config4"]
    plan = [(workload_name(n4, e4, C), K) for C in SEGMENT_WIDTHS]
    for C in TRAIN_WIDTHS:
        wl = workload_name(n4, e4, C)
        plan += [("train_forward_" + wl, TRAIN_LAUNCHES), ("train_backward_" + wl, TRAIN_LAUNCHES)]
    return plan


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


def training_launches(g, C, a, K, device):
    """The two launches one training iteration consists of at width ``C`` (a middle forward iteration = gnx_spmm_dropped_chained, a
    middle backward iteration = gnx_spmm_dropped_back over the transposed structure), as closures over their operands: what the
    segments pass counts and bench_secondary times.  The operands have the row width the training loop really launches at
    (sparse.friendly_width: C = 7 runs as 8 with a zero column, 16-byte loads) -- the byte model is still priced at C.
    Returns (forward, backward, keep-alive)."""
    import torch
    from gnntf import sparse as sp
    n = g.n_rows
    Cp = sp.friendly_width(C, n)
    X = torch.zeros(n, Cp, device=device)
    X[:, :C] = torch.rand(n, C, device=device) * 2 - 1
    gout = torch.zeros(n, Cp, device=device)
    gout[:, :C] = torch.rand(n, C, device=device)
    scales = sp.dropped_degree_scales(g, 0.5, 1, 0, K)
    adj1 = sp.dropped_adjacency(g, 0.5, 1, 1, D=scales[1])
    S_run, Y_run = torch.zeros_like(gout), torch.empty_like(gout)
    forward = lambda: sp._launch_chained(adj1, X, X, 1.0 - a, a, True, scales[2], skip_empty=True)
    backward = lambda: sp._launch_back(adj1, gout, True, scales[0], S_run, 1.0, a * (1.0 - a), S_run, 1.0 - a, Y_run, skip_empty=True)
    return forward, backward, (X, gout, scales, adj1, S_run, Y_run)


def pmc_segments_child(args, device):
    """`bench.py --pmc-child segments` (run by measure_traffic_in_run under rocprofv3 --pmc, never by hand for a result): ONE
    process on the config-4 graph in which a marker kernel (k_stream: a 64-float gnx_stream_read) brackets each measured piece --
    the K-iteration propagation at every width of SEGMENT_WIDTHS, then TRAIN_LAUNCHES forward and backward training launches at
    every width of TRAIN_WIDTHS -- each after an unmeasured warm-up of its own (lazily built handle parts).  Measured piece i is
    segment 2 i + 1."""
    import torch
    from bench_device import build_single
    from gnntf import _native as nat
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
    for C in TRAIN_WIDTHS:
        forward, backward, keep = training_launches(g, C, a, K, device)
        with torch.no_grad():
            bracket(forward, TRAIN_LAUNCHES)
            bracket(backward, TRAIN_LAUNCHES)
        del forward, backward, keep
    torch.cuda.synchronize()


def measure_traffic_in_run(workloads, seconds=240.0, K=10):
    """rocprofv3 --pmc passes of THIS bench command, made by this process before it touches the GPU (child processes: the program
    after `--` is the interpreter itself): FETCH_SIZE and WRITE_SIZE, one pass each, per entry of ``workloads`` -- "config5" /
    "config4": one propagation step of the same timed call on the same box; "segments": the config-4 graph at every width of
    SEGMENT_WIDTHS plus the training launches, one process cut into segments by a marker kernel (pmc_segments_child).  Fills
    bench_record.IN_RUN_TRAFFIC, so that every roofline record's traffic is measured in the driver's own run rather than read from
    committed files; whatever fails (no rocprofv3, no counter access, time) leaves the committed entries in charge and says so in
    the returned notes."""
    import glob
    import shutil
    import tempfile
    notes = {}
    if shutil.which("rocprofv3") is None:
        return {w: "rocprofv3 not on PATH" for w in workloads}
    script = os.path.join(br.ROOT, "bench.py")
    t_start = time.time()
    for w in workloads:
        tmp = tempfile.mkdtemp(prefix="gnx_pmc_", dir="/tmp")
        csvs, problem = {}, None
        if w == "segments":
            child_args = ["--pmc-child", "segments", "--iterations", str(K)]
        else:
            child_args = ["--workload", w, "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-secondary", "--pmc-in-run", "off",
                          "--gather-yardstick", "off"]
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            if time.time() - t_start > seconds:
                problem = "time budget spent"
                break
            out = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "run", "--", "python3", script] + child_args
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
        how = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py " + " ".join(child_args) + "`, run by " + br.IN_RUN_MARK + \
              " on this box before its timed region (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes, per launch)"
        if problem is None and w == "segments":
            try:
                by_segment = fabric_bytes_by_segment(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"])
                for i, (name, launches) in enumerate(segment_plan(K)):
                    if by_segment.get(2 * i + 1):
                        br.IN_RUN_TRAFFIC[name] = (by_segment[2 * i + 1] / launches, how)
                missing = [name for name, _ in segment_plan(K) if name not in br.IN_RUN_TRAFFIC]
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
                br.IN_RUN_TRAFFIC[workload_name(n, e, C)] = (total, how)
                notes[w] = "measured in this run"
            elif problem is None:
                problem = "no SpMM dispatch in the counter files"
        if problem is not None:
            notes[w] = "not measured in this run (" + problem + "): the committed entries of profiles/pmc_traffic.json are used"
        shutil.rmtree(tmp, ignore_errors=True)
    notes["seconds"] = round(time.time() - t_start, 1)
    return notes
