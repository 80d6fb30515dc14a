#!/bin/bash
mkdir -p gpurun_out/r4j
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4j/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r4j/tests.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/train_roofline.py > gpurun_out/r4j/train_plain.json 2> gpurun_out/r4j/train_plain.err; cat gpurun_out/r4j/train_plain.json
GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so timeout -k 10 300 python3 tools/settled_ab.py --workload config4 2>/dev/null
GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so timeout -k 10 300 python3 tools/settled_ab.py --workload config5 2>/dev/null
[ $rc -eq 0 ]
