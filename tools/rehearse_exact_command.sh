#!/bin/bash
# The EXACT command the driver runs for N > 1 -- `python3 bench.py --gpus 5`, every option at its default
# (config 5 at full size, all-auto variant selection within its wall-clock budget, overlap probe, self check) -- rehearsed with five
# gloo ranks sharing the one card (GNX_BENCH_BACKEND=gloo: the exchange is staged through the host, so rates mean nothing; what
# counts is that every phase runs, how long each takes, and that the line is well-formed).
export TMPDIR=/tmp
O=gpurun_out/r4e
mkdir -p $O
GNX_BENCH_BACKEND=gloo timeout -k 10 1120 python3 bench.py --gpus ${1:-5} > $O/exact_gloo.json 2> $O/exact_gloo.err
rc=$?
echo "rc=$rc"; grep "^\[bench" $O/exact_gloo.err | tail -40; tail -c 1500 $O/exact_gloo.json
exit $rc
