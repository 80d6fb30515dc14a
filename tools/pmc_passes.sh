#!/bin/bash
# Generic rocprofv3 counter passes of one PYTHON script, each pass in a run of its own (no tracing beside the counters).
#   bash tools/pmc_passes.sh TAG "tools/x.py args" "CTR_A CTR_B" "CTR_C" ...
# The second argument is a script path (relative to the repo root) and its arguments -- NOT a command line: the profiler's
# preloaded library initialises the GPU before the program starts, so what follows `--` must be the interpreter itself
# (`python3 script.py ...`), never `env`, `bash -c`, a `#!/usr/bin/env` script or any other launcher that execs again.
# Results: gpurun_out/TAG/passN/..._counter_collection.csv and a per-kernel table of the counters' means in gpurun_out/TAG/table.txt
TAG=$1; CMD=$2; shift 2
set -- "$@"
SCRIPT=${CMD%% *}
case "$SCRIPT" in
  *.py) ;;
  *) echo "pmc_passes.sh: the command must start with a .py script (got '$SCRIPT'); it is run as 'python3 $CMD'"; exit 2;;
esac
[ -f "$SCRIPT" ] || { echo "pmc_passes.sh: no such script: $SCRIPT"; exit 2; }
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}
mkdir -p $OUT
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pass$i -o run -- python3 $CMD > $OUT/pass$i.out 2> $OUT/pass$i.err || { echo "pass $i failed"; tail -5 $OUT/pass$i.err; exit 1; }
done
python3 - $OUT "${PMC_FILTER:-k_spmm}" <<'PY' | tee $OUT/table.txt
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
PY
