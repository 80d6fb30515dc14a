#!/bin/bash
# rocprofv3 passes of the default bench command (kernel stats, then FETCH_SIZE and WRITE_SIZE in passes of their own).
#   gpurun --timeout 900 -- 'bash tools/gpu_profile.sh TAG [bench args]'
TAG=${1:-r2}; shift
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --pmc-in-run off $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
echo "stats done"; tail -c 400 $OUT/bench_under_rocprof.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || exit 1
echo "write done"
# keep only the small CSVs (the traces can be large)
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls -la $OUT/*
