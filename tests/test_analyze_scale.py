"""tools/analyze_scale.py on synthetic N = 1 / 2 / 4 / 8 bench lines (the driver's SCALE record holds such lines)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(n, ms, kernels=None, exchange=None):
    halo = None
    if n > 1:
        halo = {"compute_ms_alone": kernels, "exchange_ms_alone": exchange, "GBs_per_link_and_direction": 40.0, "halo_bytes_per_rank_per_iteration": 3.5e9}
    return {"metric": "propagated edges/sec (APPNP K=10)", "value": 1e9 * 10 / (ms * 1e-3), "unit": "edges/s", "n_gpus": n, "ms_per_step": ms,
            "config": {"iterations": 10, "halo": halo, "self_check": {"max_rel_err": 2e-7, "ok": True},
                       "alt_grid_feature_slices": ({"value": 1e9 * 10 / (ms * 0.8e-3), "ms_per_step": ms * 0.8} if n == 4 else None)}}


def test_table_of_a_synthetic_curve(tmp_path):
    lines = [line(1, 800.0), line(2, 700.0, 40.0, 65.0), line(4, 330.0, 21.0, 30.0), line(8, 130.0, 11.0, 9.0)]
    (tmp_path / "bench.json").write_text(json.dumps(lines[0]) + "\n")
    (tmp_path / "scale.json").write_text(json.dumps({"runs": [{"n": r["n_gpus"], "parsed": r} for r in lines[1:]]}))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "analyze_scale.py"), str(tmp_path / "bench.json"), str(tmp_path / "scale.json")],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rows = {int(l.split()[0]): l for l in res.stdout.splitlines() if l.strip() and l.split()[0].isdigit()}
    assert sorted(rows) == [1, 2, 4, 8]
    assert " 6.15 " in rows[8] and "kernels" in rows[8]                       # 800 / 130 = 6.15 x, kernel-bound at 8
    assert " 1.14 " in rows[2] and "exchange" in rows[2]                      # exchange-bound at 2
    assert "feature slices" in res.stdout and res.stdout.count("self check") == 4


def test_no_lines_is_an_error(tmp_path):
    (tmp_path / "empty.json").write_text(json.dumps({"status": "skipped"}))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "analyze_scale.py"), str(tmp_path / "empty.json")], capture_output=True, text=True)
    assert res.returncode != 0 and "no bench lines" in res.stderr
