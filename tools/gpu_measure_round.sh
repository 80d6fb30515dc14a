#!/bin/bash
# Everything profiles/ holds for a round, re-measured in one visit: the profile set (tools/gpu_profile_set.sh), the training step's
# kernel-stats / FETCH_SIZE / WRITE_SIZE passes (tools/train_roofline.py), the counter passes of rank 0's vertex blocks and their
# rehearsals (tools/sim_blocks.py).  Summaries: profiles/summarize.py, summarize_train.py, summarize_blocks.py.
#   gpurun --timeout 1200 -- 'bash tools/gpu_measure_round.sh [skip-profiles]'
#   gpurun --timeout 1200 -- 'bash tools/gpu_measure_round.sh cover-sweep [outdir]'   the weighted cover (cover_push_mask's push_weight) on rank 0's
#       block of the config-5 graph at P = 8 / 4 / 2: rows on the busiest link against the block's kernel time, and the K = 10 step under
#       emulated link time (tools/sim_blocks.py --sweep) -- VERDICT r4 item 3
export TMPDIR=/tmp
if [ "$1" = "cover-sweep" ]; then
  shift
  O=${1:-gpurun_out/cover_sweep}
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
  exit 0
fi
if [ "$1" != "skip-profiles" ]; then bash tools/gpu_profile_set.sh || exit 1; fi
O=gpurun_out/${SET:-r6g}_train
mkdir -p $O
rm -rf $O/train_stats $O/train_fetch $O/train_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o run -- python3 tools/train_roofline.py > $O/train_under_stats.json 2> $O/train_stats.err || { echo "train stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/train_fetch -o run -- python3 tools/train_roofline.py > $O/train_fetch.json 2> $O/train_fetch.err || { echo "train fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/train_write -o run -- python3 tools/train_roofline.py > $O/train_write.json 2> $O/train_write.err || { echo "train write failed"; exit 1; }
find $O -name "*_kernel_trace.csv" -size +20M -delete
echo "train passes done"
O=gpurun_out/${SET:-r6g}_blocks
mkdir -p $O
for cfg in "8 cover 2" "8 cover 4" "8 pull 2" "4 cover 2" "2 cover 2"; do
  set -- $cfg; P=$1; COVER=$2; CH=$3
  T=p${P}_${COVER}_c${CH}
  rm -rf $O/${T}_FETCH_SIZE $O/${T}_WRITE_SIZE
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/${T}_$ctr -o run -- python3 tools/sim_blocks.py --world $P --cover $COVER --chunks $CH --pmc-iterations 5 \
        > $O/${T}_$ctr.json 2> $O/${T}_$ctr.err || { echo "$T $ctr failed"; tail -5 $O/${T}_$ctr.err; exit 1; }
  done
  echo "$T done: $(tail -c 200 $O/${T}_FETCH_SIZE.json)"
done
for P in 2 4 8; do timeout -k 10 300 python3 tools/sim_blocks.py --world $P > gpurun_out/${SET:-r6g}_blocks/sim_blocks_p$P.json 2> gpurun_out/${SET:-r6g}_blocks/sim_blocks_p$P.err || { echo "sim $P failed"; exit 1; }; done
echo "all done"
