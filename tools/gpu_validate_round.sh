#!/bin/bash
# Validation of a round in one visit: the -m gpu tests, the default bench line (with its in-run counter passes), reduced-size all-auto
# rehearsals of N > 1 on gloo ranks, a seeded kernel fuzz.
#   gpurun --timeout 1200 -- 'bash tools/gpu_validate_round.sh'
mkdir -p gpurun_out/r4k
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4k/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r4k/tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --steps 5 --warmup 2 > gpurun_out/r4k/bench_n1.json 2> gpurun_out/r4k/bench_n1.err; brc=$?
echo "bench rc=$brc"; grep "^\[bench" gpurun_out/r4k/bench_n1.err | tail -12
bash tools/rehearse_bench.sh gpurun_out/r4k 2 5; rrc=$?
timeout -k 10 300 python3 tests/fuzz_kernels.py 800 77 > gpurun_out/r4k/fuzz_77.txt 2>&1; tail -1 gpurun_out/r4k/fuzz_77.txt
[ $rc -eq 0 ] && [ $brc -eq 0 ] && [ $rrc -eq 0 ]
