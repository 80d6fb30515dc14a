#!/bin/bash
# Round-4 GPU visit 7: the round's profile set -- kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline command and of the
# config-4 graph at every width SURVEY 8(d) names, then kernel stats of the full default bench.
for spec in "n80M_nnz1B_C128:" "n10M_nnz100M_C256:--workload config4" "n10M_nnz100M_C128:--nodes 10000000 --entries 100000000 --feats 128" \
            "n10M_nnz100M_C64:--nodes 10000000 --entries 100000000 --feats 64" "n10M_nnz100M_C8:--nodes 10000000 --entries 100000000 --feats 8"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/gpu_profile.sh r4g/$tag $args > gpurun_out/r4g_$tag.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/r4g_$tag.log; exit 1; }
  echo "$tag done"
done
bash tools/gpu_profile_full.sh r4g/full > gpurun_out/r4g_full.log 2>&1 || { echo "full failed"; tail -5 gpurun_out/r4g_full.log; exit 1; }
echo "all done"
