#!/usr/bin/env python3
"""profiles/r02_full_bench_kernel_stats.csv from a `rocprofv3 --kernel-trace --stats` run of the FULL default bench
(tools/gpu_profile_set.sh: profile_full): this repo's kernels only (k_*), torch's own elementwise / sort kernels left out.

    python profiles/full_stats.py gpurun_out/r2full/stats/run_kernel_stats.csv profiles/r02_full_bench_kernel_stats.csv
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    name = r["Name"]
    if "::k_" not in name and not name.startswith("k_"):
        continue
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").replace("gnx::", "")
    name = name[:name.index("(")] if "(" in name else name
    out.append((name, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"]),
                float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
out.sort(key=lambda x: -x[2])
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls,total_ms,avg_ms,percent,min_ms,max_ms\n")
    for n, c, t, a, p, mn, mx in out:
        f.write(f'"{n}",{c},{t:.3f},{a:.4f},{p:.4g},{mn:.4f},{mx:.4f}\n')
print(len(out), "kernels")
