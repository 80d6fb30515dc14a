#!/bin/bash
# do locality orders pay at the narrow widths on a COMMUNITY-structured graph (planted partition x power-law
# degrees, N = 10M, ~100M entries, randomly relabelled -- tools/narrow_order_experiment.py --graph community)?
#   part "bins"    (round 5, first visit): the orders INSIDE the library's global degree bins, as round 4 tried on R-MAT
#   part "windows" the orders handed over as locality orders (gnx_graph_set_row_window): rows taken in windows of the numbering
#   part "window-sweep" window lengths 4096 ... 65,536 of the shipped block map (round 5's locality_window_sweep.sh)
#   part "robustness" the shipped configuration at other mixing ratios and at the config-5 size
#   part "shipped" the default order against what GNN(reorder="locality") does: kernel stats + counter passes at C = 8 and 40
# Each part times the K = 10 loop at C = 7 / 8 / 40 / 64 and takes FETCH_SIZE / WRITE_SIZE passes at C = 8.
#   gpurun --timeout 1200 -- 'bash tools/community_order_experiment.sh OUTDIR [bins|windows|window-sweep|shipped|robustness]'
export TMPDIR=/tmp
O=${1:-gpurun_out/r5b}
PART=${2:-windows}
mkdir -p $O
T=tools/narrow_order_experiment.py

fetch_pass() {   # LABEL then the tool's arguments: FETCH_SIZE + WRITE_SIZE passes, bytes per launch appended to $O/fetch.jsonl
  local label=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${label}_$ctr
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${label}_$ctr -o run -- python3 $T "$@" --rounds 1 \
        > $O/pmc_${label}_$ctr.json 2> $O/pmc_${label}_$ctr.err || { echo "$label $ctr failed"; tail -5 $O/pmc_${label}_$ctr.err; return 1; }
  done
  python3 - "$O" "$label" <<'PY'
import glob, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
O, label = sys.argv[1:3]
one = lambda d: max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
total = bench.fabric_bytes_per_launch(one(f"{O}/pmc_{label}_FETCH_SIZE"), one(f"{O}/pmc_{label}_WRITE_SIZE"))
rec = json.loads(open(f"{O}/pmc_{label}_FETCH_SIZE.json").read().strip().splitlines()[-1])
line = dict(run=label, graph=rec["graph"], order=rec["order"], window=rec.get("window", 0), C=rec["C"], ms_per_K10_under_the_profiler=rec["ms_per_K10"],
            fabric_GB_per_launch=total / 1e9, alg_GB_per_launch=rec["alg_GB_per_launch"], ratio=total / 1e9 / rec["alg_GB_per_launch"])
print(json.dumps(line))
open(f"{O}/fetch.jsonl", "a").write(json.dumps(line) + "\n")
PY
  find $O/pmc_${label}_* -name "*_counter_collection.csv" -size +30M -delete
}

if [ "$PART" = "shipped" ]; then
  # what GNN(reorder="locality") does (label-propagation order, window 4096, one window per XCD chunk) against the default order:
  # kernel stats + FETCH_SIZE / WRITE_SIZE passes at the widths gnntf's APPNP propagates
  for C in 8 40; do
    for spec in "default:--only workload" "locality:--window 4096 --only lpa"; do
      name=${spec%%:*}; args=${spec#*:}
      fetch_pass shipped_${name}_C$C --graph community --feats $C $args || exit 1
      rm -rf $O/stats_${name}_C$C
      timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${name}_C$C -o run -- python3 $T --graph community --feats $C $args --rounds 1 \
          > $O/stats_${name}_C$C.json 2> $O/stats_${name}_C$C.err || { echo "stats $name $C failed"; exit 1; }
      rm -f $O/stats_${name}_C$C/*kernel_trace.csv
    done
  done
  echo "community experiment (shipped) done"; exit 0
fi
if [ "$PART" = "window-sweep" ]; then
  # window length of a locality order (the XCD-aware block map takes one window per chunk), label propagation and planted orders
  for W in 4096 8192 16384 32768 65536; do
    timeout -k 10 300 python3 $T --graph community --window $W --feats 7,8,40,64 --only workload,planted,lpa >> $O/window_sweep.jsonl 2>> $O/window_sweep.err || exit 1
  done
  timeout -k 10 300 python3 $T --graph community --window 16384 --feats 128,256 --only workload,planted,lpa >> $O/window_sweep.jsonl 2>> $O/window_sweep.err || exit 1
  timeout -k 10 300 python3 $T --graph rmat --window 16384 --feats 8,40 --only workload,lpa >> $O/window_sweep.jsonl 2>> $O/window_sweep.err || exit 1
  python3 -c "
import json
for l in open('$O/window_sweep.jsonl'):
    d=json.loads(l); print('  %-9s %-8s w=%-7d C=%-3d %.2f ms' % (d['graph'], d['order'], d['window'], d['C'], d['ms_per_K10']))
"
  echo "community experiment (window-sweep) done"; exit 0
fi
if [ "$PART" = "robustness" ]; then
  # how the shipped configuration behaves away from the graph it was tuned on: more pairs leaving their community (mix 0.4 / 0.6),
  # and the config-5 size (80M vertices / ~1B entries, C = 128)
  for MIX in 0.4 0.6; do
    timeout -k 10 300 python3 $T --graph community --mix $MIX --window 4096 --feats 8,40 --only workload,lpa,planted > $O/robust_mix$MIX.jsonl 2> $O/robust_mix$MIX.err || { echo "mix $MIX failed"; exit 1; }
  done
  timeout -k 10 800 python3 $T --graph community --nodes 80000000 --entries 1000000000 --window 4096 --feats 128 --only workload,lpa --rounds 2 \
      > $O/robust_80M_C128.jsonl 2> $O/robust_80M_C128.err || { echo "80M failed"; tail -3 $O/robust_80M_C128.err; exit 1; }
  python3 -c "
import json, glob
for f in sorted(glob.glob('$O/robust_*.jsonl')):
    for l in open(f):
        d=json.loads(l); print('  %-26s %-8s C=%-3d %.2f ms  share %s  order %.2f s' % (f.split('/')[-1], d['order'], d['C'], d['ms_per_K10'], d.get('share_of_pairs_within_a_window'), d['order_seconds']))
"
  echo "community experiment (robustness) done"; exit 0
fi
if [ "$PART" = "bins" ]; then
  timeout -k 10 400 python3 $T --graph community --feats 7,8,40 > $O/community_timing.jsonl 2> $O/community_timing.err || { echo timing failed; tail -5 $O/community_timing.err; exit 1; }
  cut -c1-260 $O/community_timing.jsonl
  for order in workload bfs planted; do fetch_pass bins_$order --graph community --feats 8 --only $order || exit 1; done
else
  timeout -k 10 500 python3 $T --graph community --window 65536 --feats 7,8,40,64 --only workload,planted,lpa,bfs > $O/windows_community_w64k.jsonl 2> $O/windows_community_w64k.err \
      || { echo "community windows failed"; tail -5 $O/windows_community_w64k.err; exit 1; }
  cut -c1-300 $O/windows_community_w64k.jsonl
  for W in 16384 262144 1048576; do
    timeout -k 10 300 python3 $T --graph community --window $W --feats 8,40 --only planted,lpa > $O/windows_community_w$W.jsonl 2> $O/windows_community_w$W.err || { echo "window $W failed"; exit 1; }
    cut -c1-300 $O/windows_community_w$W.jsonl
  done
  timeout -k 10 400 python3 $T --graph rmat --window 65536 --feats 8,40 --only workload,lpa,bfs > $O/windows_rmat_w64k.jsonl 2> $O/windows_rmat_w64k.err || { echo "rmat windows failed"; exit 1; }
  cut -c1-300 $O/windows_rmat_w64k.jsonl
  fetch_pass windows_planted --graph community --window 65536 --feats 8 --only planted || exit 1
  fetch_pass windows_lpa --graph community --window 65536 --feats 8 --only lpa || exit 1
fi
echo "community experiment ($PART) done"
