#!/bin/bash
# A/B of the long-row threshold / chunk size: builds are gnn-tf_amd/lib/libgnx_T<row>_<chunk>.so
# (hipcc -DGNX_LONG_ROW=<row> -DGNX_LONG_CHUNK=<chunk>)
cd $GRAFT_REPO_ROOT
cp gnn-tf_amd/lib/libgnx.so /tmp/libgnx_default.so
for T in default $(ls gnn-tf_amd/lib | grep "libgnx_T" | sed 's/libgnx_T//; s/.so//'); do
  if [ $T = default ]; then cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so; else cp gnn-tf_amd/lib/libgnx_T$T.so gnn-tf_amd/lib/libgnx.so; fi
  echo "== LONG_ROW_CHUNK=$T"
  timeout -k 10 300 python tools/bench_widths.py --widths 32,256 --skip-train --skip-arxiv 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for w in d['widths']: print(w['C'], round(w['ms'],3))"
done
cp /tmp/libgnx_default.so gnn-tf_amd/lib/libgnx.so
