#!/bin/bash
# Validation of a round in one visit: the -m gpu tests (incl. the fixed fuzz slice and the pinned fuzz case), smoke(), then the default
# bench line (with its in-run counter passes and yardsticks), then a reduced-size all-auto rehearsal of N = 2 on gloo ranks with a
# weighted cover in the selection.
#   gpurun --timeout 1200 -- 'bash tools/gpu_validate_round.sh [outdir]'
O=${1:-gpurun_out/r6a}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x -s > $O/tests.log 2>&1; rc=$?
tail -4 $O/tests.log; grep "seed 303" $O/tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; src=$?; tail -1 $O/smoke.log; echo "smoke rc=$src"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; brc=$?
echo "bench rc=$brc, line $(wc -c < $O/bench_n1.json) bytes"; grep "^\[bench" $O/bench_n1.err | tail -24
cp bench_detail_n1.json $O/ 2>/dev/null
GNX_REHEARSE_ARGS="--push-weights 0.5" bash tools/rehearse_bench.sh $O 2; rrc=$?
[ $rc -eq 0 ] && [ $src -eq 0 ] && [ $brc -eq 0 ] && [ $rrc -eq 0 ]
