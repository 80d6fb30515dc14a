#!/bin/bash
# The P = 8 schedule of rank 0's block with EMULATED link time (tools/sim_blocks.py --transport sleep:GBs: one spinning thread per exchange
# for as long as the busiest link would need; no HBM traffic, no CUs): what is left of a step beyond max(kernels, exchange) is the schedule's.
mkdir -p gpurun_out
for cfg in "sleep:64 2 " "sleep:50 2 " "sleep:50 2 --early-pull" "sleep:50 4 " "sleep:50 4 --early-pull" "sleep:42 2 " "sleep:42 2 --early-pull" "sleep:42 4 --early-pull" "sleep:35 2 " "sleep:35 2 --early-pull" "sleep:35 4 --early-pull" "sleep:35 1 "; do
  set -- $cfg
  timeout -k 10 200 python tools/sim_blocks.py --world 8 --rank 0 --transport $1 --chunks $2 --lane-skip 1 $3 2>/dev/null | python3 -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):])
print('$1 chunks=$2 $3', 'kernels/iter %.2f'%d['kernels_ms_per_iteration'], 'exchange/iter %.2f'%d['loopback_copy_ms_per_iteration'], 'step/10 %.2f'%(d['step_ms_K10_loopback']/10))
"
done
