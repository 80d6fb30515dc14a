"""Per-kernel averages of a rocprofv3 --pmc run of the dense kernels (counter_collection.csv + kernel_trace.csv in one directory tree)."""
import csv, collections, glob, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "dense"
cc = glob.glob(root + "/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(root + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    if pat in r["Kernel_Name"]:
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in agg.items():
    d = sorted(dur[k]); ms = d[len(d) // 2]
    c = {n: sum(x) / len(x) for n, x in v.items()}
    print(k, "median ms", round(ms, 3))
    for n, x in c.items():
        print("   ", n, x)
    if "SQ_BUSY_CYCLES" in c:
        cyc = c["SQ_BUSY_CYCLES"] / 32
        print("    clock GHz ~", round(cyc / ms / 1e6, 3), " MFMA busy share", round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc), 3))
