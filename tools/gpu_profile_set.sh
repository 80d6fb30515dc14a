#!/bin/bash
# The round's profile set -- kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline command and of the
# config-4 graph at every width SURVEY 8(d) names, then kernel stats of the full default bench.
mkdir -p gpurun_out/${SET:-r4g}
for spec in "n80M_nnz1B_C128:" "n10M_nnz100M_C256:--workload config4" "n10M_nnz100M_C128:--nodes 10000000 --entries 100000000 --feats 128" \
            "n10M_nnz100M_C64:--nodes 10000000 --entries 100000000 --feats 64" "n10M_nnz100M_C8:--nodes 10000000 --entries 100000000 --feats 8"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/gpu_profile.sh ${SET:-r4g}/$tag $args > gpurun_out/${SET:-r4g}_$tag.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/${SET:-r4g}_$tag.log; exit 1; }
  echo "$tag done"
done
bash tools/gpu_profile_full.sh ${SET:-r4g}/full > gpurun_out/${SET:-r4g}_full.log 2>&1 || { echo "full failed"; tail -5 gpurun_out/${SET:-r4g}_full.log; exit 1; }
echo "all done"
