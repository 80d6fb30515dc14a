#!/bin/bash
# Rehearsals of `bench.py --gpus N` on ONE card: N ranks share cuda:0 and exchange over gloo (staged through the host, so rates mean
# nothing).  Exercises everything the driver's multi-GPU run does except RCCL itself: launcher environment, plan building for every
# halo plan, the variant table, the timed region, the self check through the exchange, the JSON line.
# A GPU box admits at most 6 processes on its card and the launcher is one of them, so N <= 5 here; all 8 blocks of one graph run as
# threads of one process in tests/test_gpu_fullsize.py (same plan, same kernels, exchange through shared memory).
#   tools/rehearse_bench.sh OUTDIR [N ...]          reduced graph, all-auto selection (GNX_REHEARSE_ARGS: further bench arguments)
#   tools/rehearse_bench.sh OUTDIR exact [N]        the EXACT driver command -- `python3 bench.py --gpus N`, every option at its default
#                                                   (config 5 at full size): what counts is that every phase runs and how long each takes
set -o pipefail
export TMPDIR=/tmp
out=${1:-gpurun_out}; shift
mkdir -p "$out"
if [ "$1" = "exact" ]; then
  # (the driver's command is `bench.py --gpus N --steps 20 --warmup 5`; a host-staged step takes tens of seconds, so fewer steps here)
  GNX_BENCH_BACKEND=gloo timeout -k 10 1120 python3 bench.py --gpus ${2:-2} --steps ${GNX_REHEARSE_STEPS:-3} --warmup 1 > $out/exact_gloo.json 2> $out/exact_gloo.err
  rc=$?
  echo "rc=$rc, line $(wc -c < $out/exact_gloo.json) bytes"; grep "^\[bench" $out/exact_gloo.err | tail -40; tail -c 1500 $out/exact_gloo.json
  cp bench_detail_n${2:-2}.json $out/ 2>/dev/null
  exit $rc
fi
ranks=${@:-2 4 5}
for n in $ranks; do
  port=$((29600 + n))
  GNX_BENCH_BACKEND=gloo timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
      bench.py --gpus "$n" --nodes 1500000 --entries 18000000 --feats 64 --steps 2 --warmup 1 --no-alt-grid $GNX_REHEARSE_ARGS \
      > "$out/rehearsal_gloo_n$n.json" 2> "$out/rehearsal_gloo_n$n.err" || { echo "rehearsal with $n ranks failed"; tail -5 "$out/rehearsal_gloo_n$n.err"; exit 1; }
  echo "rehearsal $n ranks: $(python -c "import json,sys; d=json.load(open('$out/rehearsal_gloo_n$n.json')); print(d['ms_per_step'], d['config']['halo']['chosen'], d['config']['self_check']['ok'])")"
done
