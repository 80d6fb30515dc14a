#!/bin/bash
# kernel stats of the FULL default bench run (primary workload + secondary block: config 4, widths, training step, matrix-core kernels, config 3)
TAG=${1:-r2full}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pmc-in-run off > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
rm -f $OUT/stats/run_kernel_trace.csv
ls -la $OUT/stats
