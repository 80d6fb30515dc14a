#!/bin/bash
# tests, then the headline alone (no secondary) to see what skipping the settled rows of the work buffer buys
mkdir -p gpurun_out/r4i
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4i/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r4i/tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-secondary --cpu-seconds 0 --pmc-in-run off > gpurun_out/r4i/bench_headline.json 2> gpurun_out/r4i/bench_headline.err; echo "bench rc=$?"
python3 -c "import json; d=json.load(open('gpurun_out/r4i/bench_headline.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['self_check'])"
timeout -k 10 300 python bench.py --workload config4 --steps 5 --warmup 2 --no-secondary --cpu-seconds 0 --pmc-in-run off > gpurun_out/r4i/bench_c4.json 2> gpurun_out/r4i/bench_c4.err
python3 -c "import json; d=json.load(open('gpurun_out/r4i/bench_c4.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
[ $rc -eq 0 ]
