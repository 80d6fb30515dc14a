#!/bin/bash
# One GPU-box visit: the -m gpu tests, then (unless the tests were killed / crashed) the default bench line.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh TAG [pytest args]'
TAG=${1:-r2}; shift
mkdir -p gpurun_out
timeout -k 10 780 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}_tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests were killed or crashed: not starting the bench"; exit $rc; fi
timeout -k 10 380 python bench.py --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
brc=$?
echo "bench rc=$brc"; tail -c 1500 gpurun_out/${TAG}_bench_n1.json; tail -5 gpurun_out/${TAG}_bench_n1.err
[ $rc -eq 0 ] && [ $brc -eq 0 ]
