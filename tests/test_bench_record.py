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
