"""bench_record.py -- the byte models, the roofline record and the ONE stdout line of bench.py.

Everything here is host arithmetic on numbers bench.py measured: it runs without a GPU (tests/test_bench_record.py).
The stdout line is what the driver parses: contract keys + ``config`` + ``roofline`` + ``cpu_baseline``, at most LINE_LIMIT
bytes, no string above STRING_LIMIT characters; everything else goes to the detail file named in ``config.detail_file``.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
WORKLOADS = {"config5": (80_000_000, 1_000_000_000, 128),       # BASELINE.json configs[4]: the scaling graph (default)
             "config4": (10_000_000, 100_000_000, 256)}          # BASELINE.json configs[3]: the roofline run
LINE_LIMIT = 12_000      # bytes of the stdout line (round 4's 18.6 KB line reached the driver's record, round 5's 36.2 KB did not)
STRING_LIMIT = 120       # the driver's record cuts strings at 120 characters


# ---- byte models ---------------------------------------------------------------------------------------------------------
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


def workload_name(n, nnz, C):
    return f"rmat_n{n}_nnz{nnz}_C{C}"


# ---- counters -------------------------------------------------------------------------------------------------------------
IN_RUN_TRAFFIC = {}                # workload name -> (bytes per launch, source): counter passes made by THIS run (bench_pmc)
MEASURED_READ_PEAK = [None]        # in-run read-only streaming rate (set once by bench.main)
IN_RUN_MARK = "THIS bench process"


def pmc_traffic(name):
    """Fabric (L2-miss) bytes per launch of this workload: from the rocprofv3 --pmc passes this very run made before its timed
    region when there are any (bench_pmc.measure_traffic_in_run), else from the committed builder-run passes
    (profiles/pmc_traffic.json), else None."""
    if name in IN_RUN_TRAFFIC:
        return IN_RUN_TRAFFIC[name]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path)).get("workloads", {}).get(name)
    if not rec:
        return None, None
    return float(rec["fabric_bytes_per_launch"]), (f"profiles/pmc_traffic.json [{rec.get('source', '')}] -- builder-run rocprofv3 --pmc passes of "
                                                   f"this command, NOT measured in this run")


FRAC_LEVEL = "fabric: bytes leaving the L2s incl. Infinity-Cache hits -- NOT DRAM bandwidth (see dram_frac_*, no_reuse_gather_*)"
ROOFLINE_NOTE = ("one launch = one fused SpMM+mix iteration incl. its long-row kernels; achieved = min(B_alg, traffic) / launch time; "
                 "traffic = FETCH_SIZE x2 + WRITE_SIZE (separate --pmc passes): bytes leaving the L2s, a fabric-level figure (Infinity-Cache hits "
                 "counted; fabric_rate_over_read_stream > 1 shows them); DRAM moved between dram_frac_lower_bound (B_min / t / peak) and "
                 "dram_frac_upper_bound (min(frac, in-run read stream / peak)); no_reuse_gather_*: the same kernel on a d-regular random graph "
                 "of the same N and C, where nothing is reused and B_alg IS the DRAM traffic; measured_peak = in-run stream copy, "
                 "measured_read_peak = in-run read-only stream")


def roofline_record(n, nnz, C, launch_s, K, name, measured_peak, b_alg=None, b_min=None, what=None):
    """achieved = min(B_alg, B_rocprof) / t per launch; B_alg, B_min, B_rocprof and t printed together.
    ``frac`` is a FABRIC-level figure (``frac_level``): the counters see what leaves the L2s, and the Infinity Cache serves part of
    it.  What can be said about DRAM itself is carried beside it: ``dram_frac_lower_bound`` (every array touched once: B_min / t /
    peak) and ``dram_frac_upper_bound`` (DRAM cannot have moved more than the fabric did, nor faster than this box streams reads
    in this run) -- the fabric figure is never below either -- and, once add_gather_ceiling has run, the rate of the same kernel
    on a graph WITHOUT reuse, where fabric bytes are DRAM bytes.
    Without a PMC entry for ``name`` the min() cannot be taken: ``achieved`` / ``frac`` are null (``min_rule_applied`` false) and
    ``frac_bound_without_counters`` = min(B_alg / t, in-run read stream) / peak is all that is printed.
    This is the FULL record (detail file); line_roofline() picks what the stdout line carries."""
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
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "frac_level": FRAC_LEVEL,
            "traffic": traffic, "traffic_source": source, "traffic_in_run": bool(source) and IN_RUN_MARK in source,
            "min_rule_applied": bool(traffic) and not stale, "traffic_entry_inconsistent_with_this_run": stale,
            "frac_bound_without_counters": None if frac is not None else min(b_alg / launch_s / 1e9 / HBM_PEAK_GBS, read_frac, 1.0),
            "alg_bytes_per_launch": b_alg, "min_bytes_per_launch": b_min,
            "frac_compulsory": compulsory, "dram_frac_lower_bound": compulsory,
            "dram_frac_upper_bound": min(frac, read_frac) if frac is not None else None,
            "launch_ms": launch_s * 1e3, "t_prop_ms": launch_s * 1e3 * K,
            "measured_peak": measured_peak, "measured_read_peak": read_peak,
            "fabric_rate_over_read_stream": (achieved / read_peak) if (read_peak and achieved) else None,
            "no_reuse_gather_GBs": None, "no_reuse_gather_frac": None, "frac_of_gather_ceiling": None,
            "note": ((what + "; ") if what else "") + ROOFLINE_NOTE}


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


# ---- the stdout line --------------------------------------------------------------------------------------------------------
def sig(x, digits=6):
    """Numbers of the stdout line: 6 significant digits are beyond what any timing here resolves, and a third of the bytes.  A value
    that is not finite becomes null: json.dumps would print NaN / Infinity, which a strict parser refuses."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    return x


def shorten(x):
    """Recursively: floats to 6 significant digits, strings cut at STRING_LIMIT (the full text is in the detail file)."""
    if isinstance(x, dict):
        return {k: shorten(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [shorten(v) for v in x]
    if isinstance(x, str) and len(x) > STRING_LIMIT:
        return x[:STRING_LIMIT - 3] + "..."
    return sig(x)


LINE_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_in_run", "min_rule_applied", "frac_level",
                      "frac_bound_without_counters", "alg_bytes_per_launch", "min_bytes_per_launch", "launch_ms", "t_prop_ms",
                      "dram_frac_lower_bound", "dram_frac_upper_bound", "measured_peak", "measured_read_peak",
                      "fabric_rate_over_read_stream", "no_reuse_gather_GBs", "no_reuse_gather_frac", "frac_of_gather_ceiling")


def line_roofline(rec):
    """The headline keys of a roofline record as the stdout line carries them: frac is recomputable from them alone
    (min(alg_bytes_per_launch, traffic) / launch_ms / peak); the explanations are in the detail file."""
    out = {k: rec.get(k) for k in LINE_ROOFLINE_KEYS}
    out["note"] = "achieved = min(B_alg, traffic)/launch; traffic = FETCH_SIZE x2 + WRITE_SIZE (--pmc passes), fabric level"
    return out


BLOCK_KEYS = ("ms_per_step", "launch_ms", "frac", "achieved", "traffic", "traffic_in_run", "alg_bytes_per_launch", "min_bytes_per_launch",
              "frac_compulsory", "dram_frac_upper_bound", "no_reuse_gather_frac", "frac_of_gather_ceiling", "edges_per_s")


def flat_keys(prefix, rec, **extra):
    """A secondary roofline record as FLAT scalar keys (``<prefix>_frac``, ``<prefix>_traffic`` ...) for the primary line's
    ``roofline`` object (the driver's record keeps flat scalars of that object and drops nested ones): the roofline run
    (config 4) must be recomputable from these keys alone."""
    both = dict(rec, **extra)
    return {f"{prefix}_{k}": both[k] for k in BLOCK_KEYS if both.get(k) is not None}


def triple(prefix, rec):
    """One frac / traffic / ms triple of a secondary record (a width of the config-4 graph): ms = one launch."""
    out = {f"{prefix}_ms": rec["launch_ms"]}
    if rec.get("frac") is not None:
        out[f"{prefix}_frac"], out[f"{prefix}_traffic"] = rec["frac"], rec["traffic"]
    return out


def step_statistics(step_ms):
    """median of the >= 20 timed steps (each bracketed by events on the launch stream); min / max / mean beside it."""
    s = sorted(step_ms)
    m = len(s)
    median = s[m // 2] if m % 2 else 0.5 * (s[m // 2 - 1] + s[m // 2])
    return {"median": median, "min": s[0], "max": s[-1], "mean": sum(s) / m, "n": m}


def line_cpu_baseline(port):
    """cpu_baseline of the stdout line: the contract's five keys + the other two restatements as flat scalars."""
    if not port:
        return None
    out = {k: port[k] for k in ("value", "unit", "cores", "kind")}
    out["sample"] = port["sample_short"]
    out["spmm_only_value"] = port.get("spmm_only_value")
    for key, tag in (("scipy_single_thread", "scipy_1_thread"), ("torch_sparse_all_threads", "torch_sparse")):
        if port.get(key):
            out[tag + "_value"], out[tag + "_cores"] = port[key]["value"], port[key]["cores"]
    out["os_cpu_count"] = port.get("host", {}).get("os_cpu_count")
    return out


def fit_line(result, limit=LINE_LIMIT):
    """The line as it is written: numbers shortened, strings cut, and -- should it still be above ``limit`` -- optional blocks
    dropped in a fixed order (each drop recorded in config.dropped_from_line) until it fits.  Returns the JSON text."""
    line = shorten(result)
    order = [("roofline", "community_"), ("roofline", "train_"), ("roofline", "config4_graph_"), ("config", "halo_variants"),
             ("config", "phases_s"), ("config", "api"), ("roofline", "config4_")]
    dropped = []
    for where, prefix in order:
        if len(json.dumps(line)) <= limit:
            break
        keys = [k for k in (line.get(where) or {}) if k.startswith(prefix)]
        for k in keys:
            del line[where][k]
        if keys:
            dropped.append(f"{where}.{prefix}*")
            line["config"]["dropped_from_line"] = dropped
    text = json.dumps(line, allow_nan=False)
    if len(text) > limit:
        raise SystemExit(f"bench.py: the JSON line is {len(text)} bytes, above the {limit} it may have")
    return text


def write_detail(detail, n_gpus):
    """Everything the line does not carry (full roofline records with their notes, secondaries, yardsticks, the variant table):
    bench_detail_n<N>.json next to the script.  Returns the path as written into config.detail_file (None if it could not be written)."""
    name = f"bench_detail_n{n_gpus}.json"
    try:
        with open(os.path.join(ROOT, name), "w") as f:
            json.dump(detail, f, indent=1, default=str)       # (whatever is not JSON goes in as text: the detail file must never cost the line)
        return name
    except Exception as error:                                # read-only checkout, full disk, ...: the line is printed without it
        note(f"detail file not written: {error!r}")
        return None


# ---- progress, phases, deadline -----------------------------------------------------------------------------------------------
T_START = time.time()
LAST_NOTE = ["start"]
PHASES = {}                        # seconds per phase of the run (rank 0's clock), printed in config.phases_s


def note(msg):
    """Progress line on stderr (rank 0 only; stdout carries nothing but the JSON line)."""
    LAST_NOTE[0] = str(msg)
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - T_START, msg))
        sys.stderr.flush()


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


def phase(name, t0):
    PHASES[name] = round(PHASES.get(name, 0.0) + time.time() - t0, 2)


class Deadline:
    """--max-seconds: what the whole run may take.  ``left()`` on rank 0's clock; optional parts ask ``room(seconds)`` before they
    start and are recorded in ``dropped`` when there is none.  (N > 1: the caller makes the answer collective.)"""

    def __init__(self, seconds, clock=time.time, start=None):
        self.seconds, self.clock = float(seconds), clock
        self.start = T_START if start is None else start
        self.dropped = []

    def left(self):
        return self.seconds - (self.clock() - self.start)

    def room(self, seconds, what):
        if self.left() >= seconds:
            return True
        self.dropped.append(f"{what} (needs ~{seconds:.0f} s, {max(self.left(), 0.0):.0f} s left)")
        note(f"--max-seconds: dropped {self.dropped[-1]}")
        return False
