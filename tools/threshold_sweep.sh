#!/bin/bash
# A/B of the long-row threshold / chunk size over builds gnn-tf_amd/lib/libgnx_T<row>_<chunk>.so
# (hipcc -DGNX_LONG_ROW=<row> -DGNX_LONG_CHUNK=<chunk>), each swapped in for the product library in turn.
#   bash tools/threshold_sweep.sh widths     config-4 graph at C = 32 and 256 (tools/bench_widths.py)
#   bash tools/threshold_sweep.sh bench      the headline (config 5, C = 128) and config 4 through bench.py
MODE=${1:-widths}
cd $GRAFT_REPO_ROOT
cp gnn-tf_amd/lib/libgnx.so /tmp/libgnx_default.so
for T in default $(ls gnn-tf_amd/lib | grep "libgnx_T" | sed 's/libgnx_T//; s/.so//'); do
  if [ $T = default ]; then cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so; else cp gnn-tf_amd/lib/libgnx_T$T.so gnn-tf_amd/lib/libgnx.so; fi
  echo "== LONG_ROW_CHUNK=$T"
  if [ $MODE = widths ]; then
    timeout -k 10 300 python tools/bench_widths.py --widths 32,256 --skip-train --skip-arxiv 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for w in d['widths']: print(w['C'], round(w['ms'],3))"
  else
    for W in config5 config4; do
      timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary --pmc-in-run off 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$W ms/iteration', round(d['roofline']['launch_ms'],2))"
    done
  fi
done
cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so
