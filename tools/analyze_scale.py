#!/usr/bin/env python3
"""Reads bench.py lines (one JSON object per line, e.g. the driver's N = 1, 2, 4, 8 runs concatenated, or a SCALE_rNN.json holding
them in a list / dict) and prints, per world size, what decides the time of a propagation step: the kernels alone, the exchange
alone, the achieved link rate and the efficiency against the N = 1 line.

    python tools/analyze_scale.py BENCH_r02.json SCALE_r02.json ...
"""
import json
import sys


def lines_of(path):
    text = open(path).read().strip()
    try:
        doc = json.loads(text)
    except json.JSONDecodeError:
        return [json.loads(l) for l in text.splitlines() if l.strip().startswith("{")]
    found = []

    def walk(x):
        if isinstance(x, dict):
            if "n_gpus" in x and "value" in x:
                found.append(x)
            else:
                for v in x.values():
                    walk(v)
        elif isinstance(x, list):
            for v in x:
                walk(v)
    walk(doc)
    return found


def main():
    recs = {}
    for path in sys.argv[1:]:
        for r in lines_of(path):
            recs[int(r["n_gpus"])] = r
    if not recs:
        raise SystemExit("no bench lines found")
    base = recs.get(1)
    print("%4s %12s %10s %8s | %10s %10s %12s %10s | %s" % ("N", "G edges/s", "ms/step", "speedup", "kernels ms", "exch ms", "GB/s / link", "halo GB", "bound"))
    for n in sorted(recs):
        r = recs[n]
        K = r["config"].get("iterations", 10)
        halo = r["config"].get("halo") or {}
        speed = r["value"] / base["value"] if base else float("nan")
        kern, exch = halo.get("compute_ms_alone"), halo.get("exchange_ms_alone")
        link = halo.get("GBs_per_link_and_direction")
        gb = halo.get("halo_bytes_per_rank_per_iteration", 0) / 1e9
        per_iter = r["ms_per_step"] / K
        bound = "-"
        if kern is not None and exch is not None:
            bound = "exchange" if exch > kern else "kernels"
            bound += " (step %.1f ms/iter vs max(kernels, exchange) %.1f)" % (per_iter, max(kern, exch))
        print("%4d %12.2f %10.1f %8.2f | %10s %10s %12s %10.2f | %s" % (
            n, r["value"] / 1e9, r["ms_per_step"], speed, "%.2f" % kern if kern is not None else "-", "%.2f" % exch if exch is not None else "-",
            "%.1f" % link if link is not None else "-", gb, bound))
        alt = r["config"].get("alt_grid_feature_slices")
        if alt and alt.get("value"):
            print("%4s %12.2f %10.1f %8.2f | feature slices (graph replicated, no exchange)" % ("", alt["value"] / 1e9, alt["ms_per_step"],
                                                                                               alt["value"] / base["value"] if base else float("nan")))
        check = r["config"].get("self_check")
        if check:
            print("%4s self check: max relative deviation %.2e (%s)" % ("", check["max_rel_err"], "ok" if check["ok"] else "FAILED"))


if __name__ == "__main__":
    main()
