#!/bin/bash
# Round 5: window length of a locality order (gnx_graph_set_row_window; the XCD-aware block map takes one window per chunk), community graph.
export TMPDIR=/tmp
O=${1:-gpurun_out/r5g}; mkdir -p $O
T=tools/narrow_order_experiment.py
run() { echo "== $*"; "$@" 2>> $O/err.txt | tee -a $O/sweep.jsonl | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-9s %-8s w=%-7d C=%-3d %.2f ms' % (d['graph'], d['order'], d['window'], d['C'], d['ms_per_K10']))
" | tee -a $O/sweep.txt; }
for W in 4096 8192 16384 32768 65536; do run timeout -k 10 300 python3 $T --graph community --window $W --feats 7,8,40,64 --only workload,planted,lpa; done
run timeout -k 10 300 python3 $T --graph community --window 16384 --feats 128,256 --only workload,planted,lpa
run timeout -k 10 300 python3 $T --graph rmat --window 16384 --feats 8,40 --only workload,lpa
echo done
