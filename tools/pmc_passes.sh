#!/bin/bash
# Generic rocprofv3 counter passes of one PYTHON script, each pass in a run of its own (no tracing beside the counters).
#   bash tools/pmc_passes.sh TAG "tools/x.py args" "CTR_A CTR_B" "CTR_C" ...
# The second argument is a script path (relative to the repo root) and its arguments -- NOT a command line: the profiler's
# preloaded library initialises the GPU before the program starts, so what follows `--` must be the interpreter itself
# (`python3 script.py ...`), never `env`, `bash -c`, a `#!/usr/bin/env` script or any other launcher that execs again.
# Results: gpurun_out/TAG/passN/..._counter_collection.csv and a per-kernel table (tools/pmc_table.py) in gpurun_out/TAG/table.txt
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
python3 tools/pmc_table.py $OUT "${PMC_FILTER:-k_spmm}" | tee $OUT/table.txt
