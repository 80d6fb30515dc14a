#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of `tools/sim_blocks.py --world P --pmc-iterations N` (rank 0's vertex block of the P-GPU run,
rehearsed on one GPU) into the block's entry of profiles/pmc_traffic.json, so that the N > 1 line of bench.py can take the
contract's min(B_alg, B_rocprof) for its roofline record.

    python profiles/summarize_blocks.py <tag> <fetch_dir> <write_dir> <entry name>

Counts every kernel between the run's two marker launches (k_stream) -- the SpMM of the interior and boundary rows and the two pack
launches of every column chunk: exactly what bench.py times as compute_ms_alone -- and divides by the iterations in between.
Same corrections as profiles/summarize.py: FETCH_SIZE / WRITE_SIZE in KiB, the read side doubled on gfx950."""
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
    return max(files, key=os.path.getmtime)


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else re.sub(r"\s+", " ", name)[:90]


def between_markers(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "k_stream" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit(f"{path}: expected two marker launches, found {len(marks)}")
    per = collections.defaultdict(list)
    for r in rows[marks[-2] + 1:marks[-1]]:
        per[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return per


def main():
    tag, fetch_dir, write_dir, name = sys.argv[1:5]
    info = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1]) if len(sys.argv) > 5 else {}
    fetch = between_markers(one(os.path.join(fetch_dir, "**", "*_counter_collection.csv")), "FETCH_SIZE")
    write = between_markers(one(os.path.join(write_dir, "**", "*_counter_collection.csv")), "WRITE_SIZE")
    iterations = int(info.get("iterations_between_markers", 0)) or min(len(v) for k, v in fetch.items() if "k_spmm" in k)
    total = 0.0
    with open(os.path.join(HERE, f"{tag}_pmc.csv"), "w") as f:
        f.write("kernel,dispatches,iterations,avg_ms_per_dispatch,FETCH_SIZE_KiB_raw_per_iteration,WRITE_SIZE_KiB_per_iteration,"
                "read_bytes_corrected(x2)_per_iteration,write_bytes_per_iteration,bytes_per_iteration\n")
        for k in sorted(fetch):
            fk = sum(v for v, _ in fetch[k]) / iterations
            wk = sum(v for v, _ in write.get(k, [])) / iterations
            rd, wr = 2 * fk * 1024, wk * 1024
            total += rd + wr
            f.write(f"\"{k}\",{len(fetch[k])},{iterations},{sum(ms for _, ms in fetch[k]) / len(fetch[k]):.4f},{fk:.1f},{wk:.1f},{rd:.4e},{wr:.4e},{rd + wr:.4e}\n")
        f.write(f"TOTAL,,{iterations},,,,,,{total:.4e}\n")
    path = os.path.join(HERE, "pmc_traffic.json")
    rec = json.load(open(path))
    rec["workloads"][name] = {"fabric_bytes_per_launch": total, "source": f"{tag}_pmc.csv", "what": "rank 0's vertex block, one iteration's kernels (SpMM of every chunk + pack)",
                              "plan": {k: info.get(k) for k in ("world", "cover", "chunks", "rows", "entries", "features")} if info else None}
    json.dump(rec, open(path, "w"), indent=1)
    print(name, "fabric bytes per iteration: %.4e" % total)


if __name__ == "__main__":
    main()
