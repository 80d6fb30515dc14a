#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/...) into the small files committed under profiles/.

    python profiles/summarize.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [--workload NAME] [--only REGEX]

--only: count only the SpMM kernels whose short name matches (C = 7: `bench.py --feats 7` runs the layer path, which pads 7 to 8
and launches the VEC = 4 kernels, AND the C entry at 7 floats per row, the VEC = 1 kernels -- `--only "<1,"` keeps the latter, which
is what the in-run segments pass measures under this workload's name).

Writes profiles/<tag>_kernel_stats.csv (top kernels of `rocprofv3 --kernel-trace --stats`) and, when
the two PMC passes are given, profiles/<tag>_pmc.csv plus this workload's entry of profiles/pmc_traffic.json
(bytes leaving the L2s per propagation launch; Infinity-Cache hits are counted in them).  PMC correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 128-B requests of wide (16 B/lane) coalesced reads as 64 B, so
the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return max(files, key=os.path.getmtime)      # gpurun merges new runs beside old ones: take the newest


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else re.sub(r"\s+", " ", name)[:90]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = only = None
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args = [a for a in args if a != workload]
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1]
        args = [a for a in args if a != only]
    tag, stats_dir = args[0], args[1]
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*_kernel_stats.csv")))))
    with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,percent,min_ms,max_ms\n")
        for r in rows[:14]:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{int(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e6:.4f},"
                    f"{r['Percentage']},{int(r['MinNs'])/1e6:.4f},{int(r['MaxNs'])/1e6:.4f}\n")
    if len(args) < 4:
        return
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args[2:4]:
        for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
            if "k_spmm" in r["Kernel_Name"] and (only is None or re.search(only, short(r["Kernel_Name"]))):
                k = short(r["Kernel_Name"])
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                per[k]["ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    total = 0.0
    # one propagation iteration = one dispatch of every kernel, except that the row kernel of a very large graph is dealt in
    # pieces (several dispatches per iteration): the iteration count is the smallest dispatch count among the SpMM kernels, and
    # every kernel's bytes are its SUM over the run divided by that count
    iterations = min(len(c["FETCH_SIZE"]) for c in per.values() if c["FETCH_SIZE"])
    with open(os.path.join(HERE, f"{tag}_pmc.csv"), "w") as f:
        f.write("kernel,dispatches,dispatches_per_iteration,avg_ms,FETCH_SIZE_KiB_raw_per_iteration,WRITE_SIZE_KiB_per_iteration,"
                "hbm_read_bytes_corrected(x2),hbm_write_bytes,hbm_bytes\n")
        for k, c in sorted(per.items()):
            fetch = sum(c["FETCH_SIZE"]) / iterations
            write = sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]) / max(len(c["FETCH_SIZE"]), 1), 1) / iterations
            rd, wr = 2 * fetch * 1024, write * 1024
            total += rd + wr
            f.write(f"\"{k}\",{len(c['FETCH_SIZE'])},{len(c['FETCH_SIZE']) / iterations:g},{sum(c['ms'])/len(c['ms']):.4f},{fetch:.1f},{write:.1f},"
                    f"{rd:.4e},{wr:.4e},{rd+wr:.4e}\n")
    path = os.path.join(HERE, "pmc_traffic.json")
    rec = json.load(open(path)) if os.path.exists(path) else {"workloads": {}}
    rec.setdefault("workloads", {})[workload] = {"fabric_bytes_per_launch": total, "source": f"{tag}_pmc.csv"}
    json.dump(rec, open(path, "w"), indent=1)
    print("fabric bytes per launch: %.4e" % total)


if __name__ == "__main__":
    main()
