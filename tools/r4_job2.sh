#!/bin/bash
# Round-4 GPU visit 2: the -m gpu tests, the narrow widths through the library's own (hub-grouped) relabelled copy, the default bench line.
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
rc=$?
tail -8 $O/tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests were killed or crashed: stopping"; exit $rc; fi
timeout -k 10 300 python3 tools/narrow_order_experiment.py --only workload --feats 8,16,32,64 > $O/narrow_library.jsonl 2> $O/narrow_library.err || { echo "narrow failed"; tail -5 $O/narrow_library.err; exit 1; }
cat $O/narrow_library.jsonl
timeout -k 10 420 python bench.py --steps 5 --warmup 2 > $O/bench_n1.json 2> $O/bench_n1.err
brc=$?
echo "bench rc=$brc"; tail -c 600 $O/bench_n1.json; tail -5 $O/bench_n1.err
[ $rc -eq 0 ] && [ $brc -eq 0 ]
