#!/bin/bash
# A/B of the long-row threshold: builds are gnn-tf_amd/lib/libgnx_T<thr>.so (hipcc -DGNX_LONG_ROW=<thr>)
cd $GRAFT_REPO_ROOT
cp gnn-tf_amd/lib/libgnx.so /tmp/libgnx_default.so
for T in default 128 256 1024 2048; do
  if [ $T = default ]; then cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so; else cp gnn-tf_amd/lib/libgnx_T$T.so gnn-tf_amd/lib/libgnx.so; fi
  echo "== LONG_ROW=$T"
  timeout -k 10 300 python tools/bench_widths.py --widths 32,256 --skip-train --skip-arxiv 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for w in d['widths']: print(w['C'], round(w['ms'],3))"
done
cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so
