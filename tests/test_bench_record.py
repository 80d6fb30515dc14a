"""bench.py's roofline record: the contract's min(B_alg, B_rocprof) needs a committed PMC entry; without one the
line must say so instead of printing B_alg / t as if it were a bandwidth."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


@pytest.fixture
def traffic_table(monkeypatch):
    table = {}
    monkeypatch.setattr(bench, "pmc_traffic", lambda name: (table.get(name), "test" if name in table else None))
    monkeypatch.setattr(bench, "MEASURED_READ_PEAK", [7000.0])
    return table


def test_without_a_pmc_entry_no_fraction_is_claimed(traffic_table):
    n, nnz, C = 10_000_000, 126_000_000, 128
    b_alg = bench.alg_bytes_per_iteration(n, nnz, C)
    rec = bench.roofline_record(n, nnz, C, b_alg / 9.0e12, 10, "rmat_block_of_8", 5600.0)        # B_alg / t = 9 TB/s: above the peak
    assert rec["traffic"] is None and rec["achieved"] is None and rec["frac"] is None
    assert rec["min_rule_applied"] is False
    assert rec["frac_upper_bound"] == pytest.approx(7000.0 / 8000.0)                               # capped at the in-run read stream
    assert rec["frac_of_measured_peak"] is None and rec["frac_of_measured_read_peak"] is None
    json.dumps(rec)
    slow = bench.roofline_record(n, nnz, C, b_alg / 2.0e12, 10, "rmat_block_of_8", 5600.0)
    assert slow["frac"] is None and slow["frac_upper_bound"] == pytest.approx(0.25)


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
    assert rec["frac_upper_bound"] <= 1.0


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
    traffic, source = bench.pmc_traffic(name)
    assert traffic == table[name]["fabric_bytes_per_launch"] and "NOT measured in this run" in source
    assert bench.pmc_traffic("no_such_workload") == (None, None)


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
    monkeypatch.setitem(bench.IN_RUN_TRAFFIC, "rmat_n80000000_nnz1000000000_C128", (1.0e9, "this run"))
    assert bench.pmc_traffic("rmat_n80000000_nnz1000000000_C128") == (1.0e9, "this run")
