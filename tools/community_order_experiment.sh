#!/bin/bash
# do the narrow-width locality orders pay on a COMMUNITY-structured graph (planted partition x power-law degrees,
# N = 10M, 100M entries, randomly relabelled -- tools/narrow_order_experiment.py --graph community)?  Times the K = 10 loop at
# C = 7 / 8 / 40 under every order, then takes FETCH_SIZE / WRITE_SIZE passes at C = 8 for the bench labelling, the BFS order and the
# generator's own communities.  Kill criterion: BFS not below 1.6 x B_alg at C = 8.
#   gpurun --timeout 1200 -- 'bash tools/community_order_experiment.sh [outdir]'
export TMPDIR=/tmp
O=${1:-gpurun_out/r5b}
mkdir -p $O
timeout -k 10 400 python3 tools/narrow_order_experiment.py --graph community --feats 7,8,40 > $O/community_timing.jsonl 2> $O/community_timing.err || { echo timing failed; tail -5 $O/community_timing.err; exit 1; }
cat $O/community_timing.jsonl | cut -c1-260
timeout -k 10 300 python3 tools/narrow_order_experiment.py --graph community --feats 40,64 --pure --only bfs,planted > $O/community_timing_pure.jsonl 2> $O/community_timing_pure.err || { echo pure timing failed; exit 1; }
cat $O/community_timing_pure.jsonl | cut -c1-260
for order in workload bfs planted; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${order}_$ctr
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${order}_$ctr -o run -- python3 tools/narrow_order_experiment.py --graph community --feats 8 --only $order --rounds 1 \
        > $O/pmc_${order}_$ctr.json 2> $O/pmc_${order}_$ctr.err || { echo "$order $ctr failed"; tail -5 $O/pmc_${order}_$ctr.err; exit 1; }
  done
  python3 - "$O" "$order" <<'PY'
import glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
O, order = sys.argv[1:3]
one = lambda d: max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
total = bench.fabric_bytes_per_launch(one(f"{O}/pmc_{order}_FETCH_SIZE"), one(f"{O}/pmc_{order}_WRITE_SIZE"))
rec = json.loads(open(f"{O}/pmc_{order}_FETCH_SIZE.json").read().strip().splitlines()[-1])
line = dict(order=order, C=8, fabric_GB_per_launch=total / 1e9, alg_GB_per_launch=rec["alg_GB_per_launch"], ratio=total / 1e9 / rec["alg_GB_per_launch"])
print(json.dumps(line))
open(f"{O}/community_fetch_C8.jsonl", "a").write(json.dumps(line) + "\n")
PY
done
find $O -name "*_counter_collection.csv" -size +30M -delete
echo "community experiment done"
