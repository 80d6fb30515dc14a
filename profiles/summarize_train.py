#!/usr/bin/env python3
"""Condenses the rocprofv3 passes of tools/train_roofline.py into profiles/<tag>_pmc.csv and the two training entries of
profiles/pmc_traffic.json (``train_forward_<workload>`` / ``train_backward_<workload>``: bytes leaving the L2s per launch).

    python profiles/summarize_train.py <tag> <fetch_dir> <write_dir> <workload> [<stats_dir>]

The run is cut into segments by its marker kernel (k_stream): 1 = forward launches, 2 = backward launches, 3 = degree scales of
all K streams, 4 = whole steps (see the tool's docstring).  Same corrections as profiles/summarize.py: FETCH_SIZE / WRITE_SIZE in
KiB, the read side doubled on gfx950."""
import collections
import csv
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SEGMENTS = {1: "forward_launches", 2: "backward_launches", 3: "degree_scales", 4: "whole_steps"}


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return max(files, key=os.path.getmtime)


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else re.sub(r"\s+", " ", name)[:90]


def segments(path, counter):
    """{segment: {kernel: [(value, ms), ...]}} of one pass."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter or "k_stream" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seg, seen, out = 0, set(), collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "k_stream" in r["Kernel_Name"]:
            if r["Dispatch_Id"] not in seen:             # (one row per counter and dispatch)
                seen.add(r["Dispatch_Id"])
                seg += 1
            continue
        out[seg][short(r["Kernel_Name"])].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return out


def main():
    tag, fetch_dir, write_dir, workload = sys.argv[1:5]
    fetch = segments(one(os.path.join(fetch_dir, "**", "*_counter_collection.csv")), "FETCH_SIZE")
    write = segments(one(os.path.join(write_dir, "**", "*_counter_collection.csv")), "WRITE_SIZE")
    traffic = {}
    with open(os.path.join(HERE, f"{tag}_pmc.csv"), "w") as f:
        f.write("segment,kernel,dispatches,launches,avg_ms_per_dispatch,FETCH_SIZE_KiB_raw_per_launch,WRITE_SIZE_KiB_per_launch,"
                "read_bytes_corrected(x2)_per_launch,write_bytes_per_launch,bytes_per_launch\n")
        for seg, label in SEGMENTS.items():
            spmm = [v for k, v in fetch[seg].items() if "k_spmm" in k]
            # one launch of the library = one dispatch of every SpMM kernel (pieces of a huge row kernel aside)
            launches = min((len(v) for v in spmm), default=0) or max((len(v) for v in fetch[seg].values()), default=1)
            total = 0.0
            for k in sorted(fetch[seg]):
                fv, wv = fetch[seg][k], write[seg].get(k, [])
                fk = sum(v for v, _ in fv) / launches
                wk = sum(v for v, _ in wv) / launches
                rd, wr = 2 * fk * 1024, wk * 1024
                if label in ("forward_launches", "backward_launches") and "k_spmm" not in k:
                    continue
                total += rd + wr
                f.write(f"{label},\"{k}\",{len(fv)},{launches},{sum(ms for _, ms in fv) / len(fv):.4f},{fk:.1f},{wk:.1f},{rd:.4e},{wr:.4e},{rd + wr:.4e}\n")
            traffic[label] = total
            f.write(f"{label},TOTAL,,{launches},,,,,,{total:.4e}\n")
    path = os.path.join(HERE, "pmc_traffic.json")
    rec = json.load(open(path))
    for label, key in (("forward_launches", "train_forward_"), ("backward_launches", "train_backward_")):
        rec["workloads"][key + workload] = {"fabric_bytes_per_launch": traffic[label], "source": f"{tag}_pmc.csv"}
    json.dump(rec, open(path, "w"), indent=1)
    print(json.dumps(traffic))
    if len(sys.argv) > 5:                                  # kernel stats of the same command
        rows = list(csv.DictReader(open(one(os.path.join(sys.argv[5], "**", "*_kernel_stats.csv")))))
        with open(os.path.join(HERE, f"{tag}_kernel_stats.csv"), "w") as f:
            f.write("kernel,calls,total_ms,avg_ms,percent,min_ms,max_ms\n")
            for r in rows[:16]:
                f.write(f"\"{short(r['Name'])}\",{r['Calls']},{int(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e6:.4f},"
                        f"{r['Percentage']},{int(r['MinNs'])/1e6:.4f},{int(r['MaxNs'])/1e6:.4f}\n")


if __name__ == "__main__":
    main()
