#!/bin/bash
# The round's profile set -- kernel stats + FETCH_SIZE / WRITE_SIZE passes (each in a run of its own) of the headline command and of
# the config-4 graph at the widths SURVEY 8(d) names and the widths gnntf's APPNP propagates (40, 7), then kernel stats of the FULL
# default bench.  Summaries: profiles/summarize.py, profiles/full_stats.py.
#   gpurun --timeout 1200 -- 'SET=r6p bash tools/gpu_profile_set.sh'
export TMPDIR=/tmp
SET=${SET:-r6g}
mkdir -p gpurun_out/$SET

profile_one() {      # TAG [bench args]: the three passes of one bench command (no secondary block, no in-run passes, no yardstick graph)
  local OUT=gpurun_out/$1; shift
  mkdir -p $OUT
  local ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --pmc-in-run off --gather-yardstick off $@"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || return 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || return 1
  find $OUT -name "*_kernel_trace.csv" -size +20M -delete      # keep only the small CSVs
}

profile_full() {     # TAG: kernel stats of the full default bench run (primary workload + yardsticks + secondary block)
  local OUT=gpurun_out/$1
  mkdir -p $OUT
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --pmc-in-run off \
      > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || return 1
  rm -f $OUT/stats/run_kernel_trace.csv
}

G4="--nodes 10000000 --entries 100000000"
for spec in "n80M_nnz1B_C128:" "n10M_nnz100M_C256:--workload config4" "n10M_nnz100M_C128:$G4 --feats 128" "n10M_nnz100M_C64:$G4 --feats 64" \
            "n10M_nnz100M_C40:$G4 --feats 40" "n10M_nnz100M_C8:$G4 --feats 8" "n10M_nnz100M_C7:$G4 --feats 7"; do
  tag=${spec%%:*}; args=${spec#*:}
  profile_one $SET/$tag $args > gpurun_out/${SET}_$tag.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${SET}_$tag.log; exit 1; }
  echo "$tag done"
done
profile_full $SET/full > gpurun_out/${SET}_full.log 2>&1 || { echo "full failed"; tail -5 gpurun_out/${SET}_full.log; exit 1; }
echo "all done"
