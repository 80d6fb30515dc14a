#!/bin/bash
# the weighted cover (cover_push_mask's push_weight) on rank 0's block of the config-5 graph at P = 2 / 4 / 8 --
# rows on the busiest link against the block's kernel time, and the K = 10 step under emulated link time (tools/sim_blocks.py --sweep).
#   gpurun --timeout 1200 -- 'bash tools/cover_weight_sweep.sh [outdir]'
O=${1:-gpurun_out/r5c}
mkdir -p $O
for P in 8 4 2; do
  timeout -k 10 420 python3 tools/sim_blocks.py --world $P --rank 0 --chunks 2 --sweep 0,0.02,0.1,0.5,4,pull --sweep-rates 35,42,50,64 > $O/cover_sweep_p$P.jsonl 2> $O/cover_sweep_p$P.err \
      || { echo "sweep P=$P failed"; tail -5 $O/cover_sweep_p$P.err; exit 1; }
  python3 -c "
import json,sys
for l in open('$O/cover_sweep_p$P.jsonl'):
    d=json.loads(l); print('P=%d plan=%-5s halo %.2f GB busiest link %.0f MB push entries %.1fM kernels %.2f ms  steps/10 %s' % (d['world'], d['plan'], d['halo_GB_per_iteration'], d['busiest_link_MB_per_iteration'], d['push_entries']/1e6, d['kernels_ms_per_iteration'], d['step_over_10_ms']))
"
done
echo "cover sweep done"
