#!/bin/bash
# Round-4 GPU visit 1: partition experiment, narrow-width order experiment (+ FETCH_SIZE per order), training-step passes.
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout -k 10 400 python3 tools/partition_experiment.py --world 8 > $O/partition_p8.json 2> $O/partition_p8.log || { echo "partition failed"; tail -5 $O/partition_p8.log; exit 1; }
echo "partition done"; cat $O/partition_p8.log | cut -c1-600
timeout -k 10 400 python3 tools/narrow_order_experiment.py --feats 8,16,32 > $O/narrow_order.jsonl 2> $O/narrow_order.err || { echo "narrow order failed"; tail -5 $O/narrow_order.err; exit 1; }
echo "narrow order done"; cat $O/narrow_order.jsonl
for ord in workload hub_grouped bfs; do
  PMC_FILTER=k_spmm bash tools/pmc_passes.sh r4a/pmc_$ord "tools/narrow_order_experiment.py --feats 8 --rounds 1 --only $ord" "FETCH_SIZE" > $O/pmc_$ord.txt 2>&1 || { echo "pmc $ord failed"; tail -5 $O/pmc_$ord.txt; exit 1; }
  echo "pmc $ord done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o run -- python3 tools/train_roofline.py > $O/train_under_stats.json 2> $O/train_stats.err || { echo "train stats failed"; tail -5 $O/train_stats.err; exit 1; }
echo "train stats done"; cat $O/train_under_stats.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/train_fetch -o run -- python3 tools/train_roofline.py > $O/train_fetch.json 2> $O/train_fetch.err || { echo "train fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/train_write -o run -- python3 tools/train_roofline.py > $O/train_write.json 2> $O/train_write.err || { echo "train write failed"; exit 1; }
find $O -name "*_kernel_trace.csv" -size +20M -delete
echo "all done"
