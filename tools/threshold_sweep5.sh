#!/bin/bash
# long-row threshold / chunk A/B on config 5 (RMAT 80M / 1B, C=128): builds are gnn-tf_amd/lib/libgnx_T<row>_<chunk>.so
cd $GRAFT_REPO_ROOT
cp gnn-tf_amd/lib/libgnx.so /tmp/libgnx_default.so
for T in default $(ls gnn-tf_amd/lib | grep "libgnx_T" | sed 's/libgnx_T//; s/.so//'); do
  if [ $T = default ]; then cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so; else cp gnn-tf_amd/lib/libgnx_T$T.so gnn-tf_amd/lib/libgnx.so; fi
  echo "== LONG_ROW_CHUNK=$T"
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('config5 ms/iteration', round(d['roofline']['launch_ms'],2))"
  timeout -k 10 300 python bench.py --workload config4 --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('config4 ms/iteration', round(d['roofline']['launch_ms'],2))"
done
cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so
