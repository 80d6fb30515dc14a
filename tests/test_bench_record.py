"""bench.py's roofline record: the contract's min(B_alg, B_rocprof) needs a committed PMC entry; without one the
line must say so instead of printing B_alg / t as if it were a bandwidth."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import bench_pmc
import bench_record


@pytest.fixture
def traffic_table(monkeypatch):
    table = {}
    monkeypatch.setattr(bench_record, "pmc_traffic", lambda name: (table.get(name), "test" if name in table else None))
    monkeypatch.setattr(bench_record, "MEASURED_READ_PEAK", [7000.0])
    return table


def test_without_a_pmc_entry_no_fraction_is_claimed(traffic_table):
    n, nnz, C = 10_000_000, 126_000_000, 128
    b_alg = bench.alg_bytes_per_iteration(n, nnz, C)
    rec = bench.roofline_record(n, nnz, C, b_alg / 9.0e12, 10, "rmat_block_of_8", 5600.0)        # B_alg / t = 9 TB/s: above the peak
    assert rec["traffic"] is None and rec["achieved"] is None and rec["frac"] is None
    assert rec["min_rule_applied"] is False and rec["traffic_in_run"] is False
    assert rec["frac_bound_without_counters"] == pytest.approx(7000.0 / 8000.0)                    # capped at the in-run read stream
    assert rec["fabric_rate_over_read_stream"] is None and rec["dram_frac_upper_bound"] is None
    assert "frac_upper_bound" not in rec                                                           # (round 4's name: it sat BELOW frac)
    json.dumps(rec)
    slow = bench.roofline_record(n, nnz, C, b_alg / 2.0e12, 10, "rmat_block_of_8", 5600.0)
    assert slow["frac"] is None and slow["frac_bound_without_counters"] == pytest.approx(0.25)


def test_with_a_pmc_entry_the_min_rule_holds(traffic_table):
    n, nnz, C = 10_000_000, 100_000_000, 256
    b_alg = bench.alg_bytes_per_iteration(n, nnz, C)
    assert b_alg == 100_000_000 * 1032 + 10_000_000 * 2052                                         # 123.7 GB
    traffic_table["w"] = 0.9 * b_alg
    rec = bench.roofline_record(n, nnz, C, 0.0145, 10, "w", 5600.0)
    assert rec["min_rule_applied"] and rec["achieved"] == pytest.approx(0.9 * b_alg / 0.0145 / 1e9)
    assert rec["frac"] == pytest.approx(rec["achieved"] / 8000.0) and rec["frac"] <= 1.0
    traffic_table["w"] = 3.0 * b_alg                                                               # wasteful traffic does not inflate it
    assert bench.roofline_record(n, nnz, C, 0.017, 10, "w", 5600.0)["achieved"] == pytest.approx(b_alg / 0.017 / 1e9)


def test_a_fraction_above_one_is_never_printed(traffic_table):
    n, nnz, C = 1_000_000, 10_000_000, 64
    b_alg = bench.alg_bytes_per_iteration(n, nnz, C)
    traffic_table["w"] = b_alg
    rec = bench.roofline_record(n, nnz, C, b_alg / 9.5e12, 10, "w", 5600.0)                        # the entry cannot belong to this launch
    assert rec["frac"] is None and rec["achieved"] is None and rec["traffic_entry_inconsistent_with_this_run"]
    assert rec["frac_bound_without_counters"] <= 1.0


def test_no_field_of_the_record_contradicts_another(traffic_table):
    """round 4's line carried frac 0.885 beside frac_upper_bound 0.854.  The fabric-level fraction is labelled
    as such, the DRAM-level bounds sit on either side of what DRAM can have moved and never above the fabric figure, and a bound
    'without counters' is printed only when there are none."""
    n, nnz, C = 80_000_000, 1_000_000_000, 128
    b_alg, b_min = bench.alg_bytes_per_iteration(n, nnz, C), bench.min_bytes_per_iteration(n, nnz, C)
    traffic_table["w"] = 543.9e9                                                                   # round 4's in-run counter figure
    monkey_read = bench_record.MEASURED_READ_PEAK[0]                                                      # 7000 GB/s (fixture)
    rec = bench.roofline_record(n, nnz, C, 0.07685, 10, "w", 5565.0)
    assert rec["frac"] == pytest.approx(543.9e9 / 0.07685 / 1e9 / 8000.0) and "fabric" in rec["frac_level"] and "NOT DRAM" in rec["frac_level"]
    assert rec["frac_bound_without_counters"] is None and len(rec["frac_level"]) <= 120          # (the driver's record cuts strings at 120 characters)
    assert rec["dram_frac_lower_bound"] == pytest.approx(b_min / 0.07685 / 1e9 / 8000.0) == rec["frac_compulsory"]
    assert rec["dram_frac_upper_bound"] == pytest.approx(min(rec["frac"], monkey_read / 8000.0))
    assert rec["dram_frac_lower_bound"] <= rec["dram_frac_upper_bound"] <= rec["frac"] <= 1.0
    assert rec["fabric_rate_over_read_stream"] == pytest.approx(rec["achieved"] / monkey_read)    # may exceed 1: named as a ratio, not a fraction
    assert not [k for k in rec if k.startswith("frac_of_measured")]
    # the in-run no-reuse yardstick beside it
    assert rec["no_reuse_gather_GBs"] is None and rec["frac_of_gather_ceiling"] is None
    bench_record.add_gather_ceiling(rec, {"GBs": 5800.0, "launch_ms": 130.0, "entries": 1_279_999_000})
    assert rec["no_reuse_gather_frac"] == pytest.approx(0.725) and rec["frac_of_gather_ceiling"] == pytest.approx(rec["achieved"] / 5800.0)
    assert bench_record.add_gather_ceiling(dict(rec), None)["no_reuse_gather_GBs"] == 5800.0               # no yardstick: nothing changes
    json.dumps(rec)


def test_flat_keys_carry_what_recomputing_the_fraction_needs(traffic_table):
    """the driver keeps flat scalars of ``roofline`` and drops ``secondary``; config 4 and the narrow widths ride
    in the primary object as <prefix>_* keys from which frac = min(alg, traffic) / (launch_ms) / peak can be recomputed."""
    n, nnz, C = 10_000_000, 100_000_000, 256
    traffic_table["c4"] = 106.7e9
    rec = bench.roofline_record(n, nnz, C, 0.01430, 10, "c4", 5600.0)
    flat = bench_record.flat_keys("config4", rec, ms_per_step=143.0, edges_per_s=7.0e9)
    for key in ("config4_ms_per_step", "config4_launch_ms", "config4_frac", "config4_traffic", "config4_alg_bytes_per_launch",
                "config4_min_bytes_per_launch", "config4_frac_compulsory", "config4_traffic_in_run", "config4_edges_per_s"):
        assert key in flat and not isinstance(flat[key], (dict, list)), key
    again = min(flat["config4_alg_bytes_per_launch"], flat["config4_traffic"]) / (flat["config4_launch_ms"] * 1e-3) / 1e9 / 8000.0
    assert again == pytest.approx(flat["config4_frac"]) and flat["config4_frac"] > 0.9
    assert "config4_no_reuse_gather_frac" not in flat                                               # absent values are left out, not null
    # the other widths ride as one frac / traffic / ms triple each (ms = one launch); no counters, no fraction
    narrow = bench_record.triple("config4_graph_C7", bench.roofline_record(n, nnz, 7, 0.0016, 10, "nothing", 5600.0))
    assert "config4_graph_C7_frac" not in narrow and narrow["config4_graph_C7_ms"] == pytest.approx(1.6)
    traffic_table["c7"] = 1.2e9
    narrow = bench_record.triple("config4_graph_C7", bench.roofline_record(n, nnz, 7, 0.0016, 10, "c7", 5600.0))
    assert sorted(narrow) == ["config4_graph_C7_frac", "config4_graph_C7_ms", "config4_graph_C7_traffic"]
    assert narrow["config4_graph_C7_frac"] == pytest.approx(min(bench.alg_bytes_per_iteration(n, nnz, 7), 1.2e9) / 0.0016 / 1e9 / 8000.0)
    assert [C for C in bench_pmc.SEGMENT_WIDTHS] == [256, 128, 64, 40, 8, 7]                             # 40 and 7: the widths gnntf's APPNP propagates (filter.py:33-35)


def test_segments_of_a_marked_counter_run(tmp_path):
    """The config-4-graph pass (`bench.py --pmc-child segments`): a marker kernel cuts ONE process into segments; measured piece i
    is segment 2 i + 1 (a warm-up of its own precedes each), bytes = FETCH x 2 KiB + WRITE KiB over the SpMM kernels only."""
    head = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    seq = [("k_build", 9), ("k_stream<8, false>", 0),                                                # segment 0 | marker -> 1
           ("k_spmm_wave<4, 8, 8>", 100), ("k_spmm_long_partial<4>", 10), ("at::fill", 77), ("k_stream<8, false>", 0),   # segment 1 (measured piece 0)
           ("k_spmm_group<4, 32, 4, false>", 55), ("k_stream<8, false>", 0),                         # segment 2 (warm-up of piece 1)
           ("k_spmm_group<4, 32, 4, false>", 50), ("k_gather_rows32", 5), ("k_stream<8, false>", 0)]  # segment 3 (measured piece 1)

    def csv_of(counter, scale):
        return head + "".join(f'{d + 1},"{k}",{counter},{v * scale}\n' for d, (k, v) in enumerate(seq))
    (tmp_path / "f.csv").write_text(csv_of("FETCH_SIZE", 1.0))
    (tmp_path / "w.csv").write_text(csv_of("WRITE_SIZE", 0.1))
    seg = bench.fabric_bytes_by_segment(str(tmp_path / "f.csv"), str(tmp_path / "w.csv"))
    assert seg[1] == pytest.approx(110 * 2048 + 11 * 1024) and seg[3] == pytest.approx(50 * 2048 + 5 * 1024)
    assert seg[2] == pytest.approx(55 * 2048 + 5.5 * 1024) and 0 not in seg
    plan = bench_pmc.segment_plan(10)
    assert [p[0] for p in plan[:2]] == ["rmat_n10000000_nnz100000000_C256", "rmat_n10000000_nnz100000000_C128"] and plan[0][1] == 10
    # the training launches at 64 and at the widths gnntf trains at (filter.py:33-35: num_classes = 40 on arxiv, 7 on Cora)
    assert plan[6] == ("train_forward_rmat_n10000000_nnz100000000_C64", bench_pmc.TRAIN_LAUNCHES) and plan[7][0].startswith("train_backward_")
    assert [p[0] for p in plan[8:]] == [f"train_{d}_rmat_n10000000_nnz100000000_C{C}" for C in (40, 7) for d in ("forward", "backward")]


def test_training_iteration_compulsory_bytes():
    n, nnz, C = 1000, 20000, 64
    assert bench.min_bytes_dropped_iteration(n, nnz, C) == 8 * nnz + 12 * n + 12 * n * C
    assert bench.min_bytes_dropped_iteration(n, nnz, C, backward=True) == 8 * nnz + 12 * n + 16 * n * C
    assert bench.min_bytes_dropped_iteration(n, nnz, C) <= bench.alg_bytes_dropped_iteration(n, nnz, nnz // 2, C)


def test_training_iteration_byte_model():
    n, nnz, kept, C = 1000, 20000, 9000, 64
    assert bench.alg_bytes_dropped_iteration(n, nnz, kept, C) == nnz * 8 + kept * 256 + n * (12 + 512)
    assert bench.alg_bytes_dropped_iteration(n, nnz, kept, C, backward=True) == nnz * 8 + kept * 256 + n * (12 + 768)
    # nothing dropped: the forward iteration moves what an eval iteration moves plus the two scale vectors
    assert bench.alg_bytes_dropped_iteration(n, nnz, nnz, C) == bench.alg_bytes_per_iteration(n, nnz, C) + 8 * n


def test_committed_pmc_entries_are_readable():
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"]
    name = bench.workload_name(80_000_000, 1_000_000_000, 128)
    assert name in table
    traffic, source = bench_record.pmc_traffic(name)
    assert traffic == table[name]["fabric_bytes_per_launch"] and "NOT measured in this run" in source
    assert bench_record.pmc_traffic("no_such_workload") == (None, None)


def test_fabric_bytes_from_counter_files(tmp_path):
    """bench.py's own reading of a FETCH_SIZE and a WRITE_SIZE pass (the passes it runs itself before the timed region): KiB,
    reads doubled, per launch = per dispatch of every SpMM kernel, a row kernel dealt in two pieces counted once per launch;
    kernels of other names (yardsticks, torch) ignored."""
    head = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"

    def rows(counter, values):
        out, d = head, 0
        for kernel, per_dispatch in values:
            for v in per_dispatch:
                d += 1
                out += f'{d},"{kernel}",{counter},{v}\n'
        return out
    fetch = rows("FETCH_SIZE", [("void k_spmm_group<4, 32, 4, false>(SpmmArgs)", [100.0, 100.0] * 3),      # two pieces per launch, 3 launches
                                ("void k_spmm_long_partial_group<4, 32, 4>(SpmmArgs)", [50.0] * 3),
                                ("void k_spmm_long_reduce<4>(SpmmArgs)", [1.0] * 3), ("k_stream<16, false>", [999.0] * 5)])
    write = rows("WRITE_SIZE", [("void k_spmm_group<4, 32, 4, false>(SpmmArgs)", [10.0, 10.0] * 3),
                                ("void k_spmm_long_partial_group<4, 32, 4>(SpmmArgs)", [5.0] * 3),
                                ("void k_spmm_long_reduce<4>(SpmmArgs)", [2.0] * 3), ("at::native::fill", [7.0])])
    (tmp_path / "f.csv").write_text(fetch)
    (tmp_path / "w.csv").write_text(write)
    total = bench.fabric_bytes_per_launch(str(tmp_path / "f.csv"), str(tmp_path / "w.csv"))
    assert total == pytest.approx(2 * (200 + 50 + 1) * 1024 + (20 + 5 + 2) * 1024)
    (tmp_path / "none.csv").write_text(head + '1,"k_stream",FETCH_SIZE,5\n')
    assert bench.fabric_bytes_per_launch(str(tmp_path / "none.csv"), str(tmp_path / "none.csv")) is None


def test_in_run_entries_take_precedence(monkeypatch):
    monkeypatch.setitem(bench_record.IN_RUN_TRAFFIC, "rmat_n80000000_nnz1000000000_C128", (1.0e9, "this run"))
    assert bench_record.pmc_traffic("rmat_n80000000_nnz1000000000_C128") == (1.0e9, "this run")


def test_in_run_counter_passes_with_a_stand_in_profiler(tmp_path, monkeypatch):
    """measure_traffic_in_run's plumbing without a GPU: a stand-in `rocprofv3` on PATH writes the counter file a pass would leave
    (or fails); the bytes per launch land in IN_RUN_TRAFFIC under the workload's / the segments' names, a failing pass leaves the
    committed table in charge and says why."""
    fake = tmp_path / "bin"
    fake.mkdir()
    script = fake / "rocprofv3"
    script.write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
ctr, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
child = a[a.index("--") + 1:]
if os.environ.get("FAKE_PROFILER_FAILS") == ctr:
    sys.stderr.write("no counter access on this box\\n"); sys.exit(3)
os.makedirs(os.path.join(out, "host", "1"), exist_ok=True)
rows, d = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"], 0
def emit(kernel, value):
    global d
    d += 1
    rows.append(f'{d},"{kernel}",{ctr},{value}')
if "--pmc-child" in child:                       # the segments process: marker | measured | marker | warm-up | marker | measured ...
    for i in range(12):
        emit("k_spmm_group<4, 4, 4, true>", 999)            # warm-up of piece i
        emit("k_stream<8, false>", 0)
        for _ in range(10 if i < 6 else 3):
            emit("k_spmm_group<4, 4, 4, true>", 100 + i)
        emit("k_stream<8, false>", 0)
else:
    for _ in range(20):
        emit("k_spmm_wave<4, 8, 8>", 50); emit("k_spmm_long_partial<4, 8>", 5); emit("k_spmm_long_reduce<4>", 1)
open(os.path.join(out, "host", "1", "run_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
''')
    script.chmod(0o755)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(bench_record, "IN_RUN_TRAFFIC", {})
    notes = bench.measure_traffic_in_run(["config4", "segments"], seconds=60.0, K=10)
    assert notes["config4"] == "measured in this run" and notes["segments"] == "measured in this run"
    plan = bench_pmc.segment_plan(10)
    for i, (name, launches) in enumerate(plan):
        per_launch = (100 + i) * 2048.0 + (100 + i) * 1024.0                       # FETCH x 2 KiB + WRITE KiB, per launch
        assert bench_record.IN_RUN_TRAFFIC[name][0] == pytest.approx(per_launch), name
    # (the segments pass carries C = 256 too: it overrides the one-command figure of config 4 with its own)
    rec = bench.roofline_record(10_000_000, 100_000_000, 256, 0.0142, 10, "rmat_n10000000_nnz100000000_C256", 5600.0)
    assert rec["traffic_in_run"] is True and "THIS bench process" in rec["traffic_source"]
    # a pass that fails: nothing is claimed for that workload, the note says why
    monkeypatch.setattr(bench_record, "IN_RUN_TRAFFIC", {})
    monkeypatch.setenv("FAKE_PROFILER_FAILS", "WRITE_SIZE")
    notes = bench.measure_traffic_in_run(["config5"], seconds=60.0)
    assert notes["config5"].startswith("not measured in this run (pass WRITE_SIZE failed (rc 3)") and "no counter access" in notes["config5"]
    assert bench_record.IN_RUN_TRAFFIC == {}


# ---- the stdout line: at most 12 KB, no string above 120 characters, fractions recomputable from it alone ----
def full_record_with_stand_in_numbers(traffic_table, n_gpus=1):
    """What bench.main assembles on rank 0, from the same pure functions, with round 5's magnitudes standing in for measurements."""
    n, nnz, C, K = 80_000_000, 1_000_000_000, 128, 10
    name = bench.workload_name(n, nnz, C)
    traffic_table[name] = 5.4386e11
    roof = bench.roofline_record(n, nnz, C, 0.07668, K, name, 5565.123456789)
    bench_record.add_gather_ceiling(roof, {"GBs": 5391.123456, "launch_ms": 131.123456, "entries": 1_279_999_000})
    flat = {}
    n4, e4 = 10_000_000, 100_000_000
    for W in bench_pmc.SEGMENT_WIDTHS:
        wl = bench.workload_name(n4, e4, W)
        traffic_table[wl] = 0.9 * bench.alg_bytes_per_iteration(n4, e4, W)
        rec = bench.roofline_record(n4, e4, W, bench.alg_bytes_per_iteration(n4, e4, W) / 7.1234567e12, K, wl, 5565.123456789)
        bench_record.add_gather_ceiling(rec, {"GBs": 5123.456789, "launch_ms": 12.3456789, "entries": 159_999_123})
        if W == 256:
            flat.update(bench_record.flat_keys("config4", rec, ms_per_step=142.123456789, edges_per_s=7.0123456789e9))
            flat.update(config4_workload=wl + "_appnp_K10", config4_rows=n4, config4_entries=e4, config4_kernel="spmm_wave64")
        else:
            flat.update(bench_record.triple(f"config4_graph_C{W}", rec))
            flat[f"config4_graph_C{W}_no_reuse_gather_frac"] = rec["no_reuse_gather_frac"]
    for W in bench_pmc.TRAIN_WIDTHS:
        for d in ("forward", "backward"):
            wl = f"train_{d}_" + bench.workload_name(n4, e4, W)
            traffic_table[wl] = 2.0e10
            rec = bench.roofline_record(n4, e4, W, 0.0045678912, K, wl, 5565.1, b_alg=bench.alg_bytes_dropped_iteration(n4, e4, e4 // 2, W, d == "backward"),
                                        b_min=bench.min_bytes_dropped_iteration(n4, e4, W, d == "backward"), what="x" * 300)
            flat.update(bench_record.triple(f"train_C{W}_{d}", rec))
        flat[f"train_C{W}_step_ms"] = 123.456789123
        if W != 64:
            flat[f"train_C{W}_step_degree_order_ms"] = 101.23456789
    for W in (256, 8):
        flat.update({f"config4_C{W}_via_layers_ms": 142.123456789, f"config4_C{W}_c_entry_ms": 142.023456789, f"config4_C{W}_layers_bitwise_equal_c_entry": True})
    flat.update(config4_C8_relu_fused_forward_ms=16.123456, config4_C8_relu_layer_by_layer_forward_ms=31.123456, config4_C8_relu_fused_max_abs_diff=2.3841858e-07,
                config3_gcn_forward_ms=0.4123456789, config3_spmm128_edges_per_s=2.1e10, config3_spmm64_edges_per_s=3.1e10,
                config2_eval_forward_ms=0.3123456789, config2_captured_train_ms_per_epoch=0.9123456789, train_kept_entries=50_001_234)
    cpu = {"value": 1.23456789e8, "unit": "edges/s", "cores": 16, "kind": "port", "spmm_only_value": 2.3456789e8, "sample": "y" * 560,
           "sample_short": "C/OpenMP oracle port, 16 thr: 1 of 10 iter.; renorm all 1000000000 entries 12.3s + SpMM first 8650000 rows 5.1s, scaled",
           "scipy_single_thread": {"value": 1.1e7, "cores": 1, "sample": "z" * 300}, "torch_sparse_all_threads": {"value": 5.5e7, "cores": 16, "sample": "w" * 300},
           "host": {"os_cpu_count": 16, "torch_threads": 16}}
    phases = {k: 12.34 for k in ("pmc_passes_in_run", "startup", "generate", "prep", "warmup_and_timed_steps", "self_check", "stream_yardsticks", "cpu_baseline",
                                 "gather_yardstick", "secondary_widths_on_config4", "secondary_via_layer_api", "secondary_training_step",
                                 "secondary_matrix_core_kernels", "secondary_community_graph", "secondary_small_configs", "secondary_workloads", "total")}
    stats = bench_record.step_statistics([766.8 + 0.01 * i for i in range(20)])
    halo = None
    if n_gpus > 1:
        table = [dict(cover=c, chunks=k, early_pull=e, step_ms=612.3456789, exchange_ms_alone=31.23456, compute_ms_alone=12.3456)
                 for c in ("cover", "pull", "cover@0.5") for k in (2, 4, 1) for e in (False, True)]
        halo = dict(max_pull_rows=12345678, max_push_rows=2345678, max_pull_only_rows=23456789, max_send_rows=13456789, max_interior_rows=3456789,
                    max_boundary_rows=6543211, max_local_rows=10000000, max_busiest_link_rows=2345678, max_push_entries=34567890, max_halo_rows=14691356,
                    cover="cover", split_rows=True, chunks=2, push_weight=0.0, early_pull=True, variants_timed_before_the_run=table,
                    variants_skipped=[dict(cover="cover@0.5", chunks=1, reason="selection budget spent " * 5)] * 4, select_seconds_budget=120.0,
                    overlap_probe={"lanes": 4, "ms": [1.0] * 16}, overlap_probe_status="probed", chosen=dict(table[0]), plan="cover",
                    pull_rows_sent=12345678, push_rows_sent=2345678, exchange_ms_alone=31.23456789, compute_ms_alone=12.3456789,
                    halo_bytes_per_rank_per_iteration=7522000000, ingress_GBs_per_rank=240.123456, GBs_per_link_and_direction=34.3456789,
                    pull_only_bytes_per_rank_per_iteration=12010000000)
    import bench_sharded
    return {"metric": "propagated edges/sec (APPNP K=10)", "value": nnz * K / (stats["median"] * 1e-3), "unit": "edges/s", "n_gpus": n_gpus, "steps": 20,
            "warmup": 5, "ms_per_step": stats["median"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "step_ms_min": stats["min"], "step_ms_max": stats["max"], "step_ms_mean": stats["mean"], "wall_ms_per_step": 767.123456,
            "timing": "median of the per-step event times (max over ranks per step)",
            "config": {"workload": name + "_appnp_K10 (BASELINE config5)", "global_rows": n, "stored_entries_total": nnz, "rows_per_rank": n // n_gpus,
                       "stored_entries_per_rank": nnz // n_gpus, "features": C, "iterations": K, "alpha": 0.1, "partition": "none",
                       "halo": bench_sharded.line_halo(halo), "prep": {"gen_s": 17.12, "prep_s": 6.34}, "kernel": "spmm_wave32",
                       "api": {"timed_call": "architecture.run(H0, first=2): the K PPRIteration layers of gnntf.APPNP (filter.py:34-35), eval, no_grad",
                               "layers": "Dropout, Dense, 10 x PPRIteration", "n_layers": 12, "gnx_appnp_propagate_ms_per_step": 766.123456789,
                               "layers_ms_per_step": 766.823456789, "bitwise_equal_to_c_entry": True},
                       "alt_grid_feature_slices": None,
                       "self_check": {"what": "H0 = sqrt(degree) x s_c is a fixed point of H <- (1-a) A_hat H + a H0: max rel. deviation through the timed path",
                                      "iterations": 10, "max_rel_err": 2.3841858e-07, "ok": True},
                       "phases_s": phases, "pmc_in_run": {"config5": "in run", "segments": "in run", "seconds": 47.1}, "dropped": [],
                       "detail_file": "bench_detail_n1.json"},
            "roofline": dict(bench_record.line_roofline(roof), **(flat if n_gpus == 1 else {})),
            "cpu_baseline": bench_record.line_cpu_baseline(cpu) if n_gpus == 1 else None}


def strings_of(x):
    if isinstance(x, dict):
        for v in x.values():
            yield from strings_of(v)
    elif isinstance(x, list):
        for v in x:
            yield from strings_of(v)
    elif isinstance(x, str):
        yield x


def test_the_line_fits(traffic_table):
    """Round 4's 18.6 KB line reached the driver's record, round 5's 36.2 KB did not: the N = 1 line stays at or below 12,000 bytes with
    every block present, no string above 120 characters, and what the judge recomputes is recomputable from the line alone."""
    result = full_record_with_stand_in_numbers(traffic_table)
    text = bench_record.fit_line(result)
    assert len(text) <= 12_000 and "\n" not in text
    line = json.loads(text)
    assert "dropped_from_line" not in line["config"]                                                   # it fits WITHOUT dropping a block
    assert max(len(s) for s in strings_of(line)) <= 120
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "config5" in line["config"]["workload"] and line["config"]["detail_file"] == "bench_detail_n1.json"
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and "secondary" not in line
    r = line["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert min(r["alg_bytes_per_launch"], r["traffic"]) / (r["launch_ms"] * 1e-3) / 1e9 / r["peak"] == pytest.approx(r["frac"], rel=1e-4)
    assert line["value"] == pytest.approx(line["config"]["stored_entries_total"] * 10 / (line["ms_per_step"] * 1e-3), rel=1e-4)
    assert line["step_ms_min"] <= line["ms_per_step"] <= line["step_ms_max"]
    assert min(r["config4_alg_bytes_per_launch"], r["config4_traffic"]) / (r["config4_launch_ms"] * 1e-3) / 1e9 / 8000.0 == pytest.approx(r["config4_frac"], rel=1e-4)
    for W in (128, 64, 40, 8, 7):                                                                      # one frac / traffic / ms triple per width
        assert {f"config4_graph_C{W}_{k}" for k in ("frac", "traffic", "ms")} <= set(r)
        b_alg = bench.alg_bytes_per_iteration(r["config4_rows"], r["config4_entries"], W)
        assert min(b_alg, r[f"config4_graph_C{W}_traffic"]) / (r[f"config4_graph_C{W}_ms"] * 1e-3) / 1e9 / 8000.0 == pytest.approx(r[f"config4_graph_C{W}_frac"], rel=1e-4)
    for W in (64, 40, 7):                                                                              # the widths gnntf trains at are in the line
        assert f"train_C{W}_forward_frac" in r and f"train_C{W}_backward_frac" in r
    # the multi-GPU line (variant table, halo block) stays below 8 KB
    multi = bench_record.fit_line(full_record_with_stand_in_numbers(traffic_table, n_gpus=8))
    assert len(multi) <= 8_000 and max(len(s) for s in strings_of(json.loads(multi))) <= 120
    assert len(json.loads(multi)["config"]["halo"]["halo_variants"]) == 18


def test_a_line_that_would_not_fit_sheds_blocks_in_a_fixed_order(traffic_table):
    result = full_record_with_stand_in_numbers(traffic_table)
    result["roofline"].update({f"community_graph_{i}": 1.0 * i for i in range(400)})
    line = json.loads(bench_record.fit_line(result))
    assert line["config"]["dropped_from_line"] == ["roofline.community_*"] and "config4_frac" in line["roofline"]
    with pytest.raises(SystemExit):
        bench_record.fit_line(dict(result, unknown_block={f"k{i}": 1.0 * i for i in range(2000)}))


def test_the_line_is_strict_json(traffic_table):
    """A NaN or an infinity anywhere in the record would make json.dumps print a token strict parsers refuse: they become null."""
    result = full_record_with_stand_in_numbers(traffic_table)
    result["roofline"]["config4_graph_C7_no_reuse_gather_frac"] = float("nan")
    result["config"]["self_check"]["max_rel_err"] = float("inf")
    text = bench_record.fit_line(result)
    line = json.loads(text, parse_constant=lambda token: pytest.fail("non-standard JSON token " + token))
    assert line["roofline"]["config4_graph_C7_no_reuse_gather_frac"] is None and line["config"]["self_check"]["max_rel_err"] is None


def test_median_of_the_timed_steps():
    s = bench_record.step_statistics([5.0, 1.0, 3.0, 100.0])
    assert s == {"median": 4.0, "min": 1.0, "max": 100.0, "mean": 27.25, "n": 4}
    assert bench_record.step_statistics([2.0, 9.0, 1.0])["median"] == 2.0
    assert bench.parse([]).steps == 20 and bench.parse([]).warmup == 5                                # >= 20 timed runs after >= 5 warm-ups


def test_deadline_drops_optional_parts_and_says_so():
    now = [0.0]
    d = bench_record.Deadline(100.0, clock=lambda: now[0], start=0.0)
    assert d.room(40.0, "a") and d.dropped == []
    now[0] = 70.0
    assert not d.room(40.0, "second field") and d.room(20.0, "b")
    assert len(d.dropped) == 1 and d.dropped[0].startswith("second field (needs ~40 s, 30 s left)")
