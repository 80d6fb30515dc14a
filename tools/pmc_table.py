#!/usr/bin/env python3
"""Per-kernel means of every counter collected by tools/pmc_passes.sh (kernels whose name contains the filter)."""
import collections
import csv
import glob
import os
import re
import sys

out, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"][:60]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, c in sorted(per.items()):
    print(k, "  dispatches per pass ~", len(c["_ms"]) // max(len(c) - 1, 1), "  avg ms %.3f" % (sum(c["_ms"]) / len(c["_ms"])))
    for name, v in sorted(c.items()):
        if name != "_ms":
            print("    %-44s %.4e" % (name, sum(v) / len(v)))
