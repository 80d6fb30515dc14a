#!/bin/bash
# Round-4 GPU visit 3: the -m gpu tests, the training step after the masked column sums and the chained backward, and its PMC passes.
export TMPDIR=/tmp
O=gpurun_out/r4c
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
rc=$?
tail -8 $O/tests.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests were killed or crashed: stopping"; exit $rc; fi
timeout -k 10 300 python3 tools/train_roofline.py > $O/train_plain.json 2> $O/train_plain.err || { echo "train failed"; tail -5 $O/train_plain.err; exit 1; }
cat $O/train_plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -o run -- python3 tools/train_roofline.py > $O/train_under_stats.json 2> $O/train_stats.err || { echo "train stats failed"; tail -5 $O/train_stats.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/train_fetch -o run -- python3 tools/train_roofline.py > $O/train_fetch.json 2> $O/train_fetch.err || { echo "train fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/train_write -o run -- python3 tools/train_roofline.py > $O/train_write.json 2> $O/train_write.err || { echo "train write failed"; exit 1; }
find $O -name "*_kernel_trace.csv" -size +20M -delete
timeout -k 10 200 python3 - > $O/c8_diff.txt 2>&1 <<'PY'
import sys, argparse, torch
sys.path[:0] = ["/root/repo", "/root/repo/gnn-tf_amd"]
import bench, gnntf
dev = torch.device("cuda:0"); gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
H0 = torch.randn(g.n_rows, 8, device=dev)
a = gnntf.appnp_propagate(adj, H0, 0.1, 10)
H = H0
for _ in range(10):
    H = gnntf.ppr_step(adj, H, H0, 0.1)
print("C=8 K loop (relabelled copy) vs step by step: max abs diff", float((a - H).abs().max()), "equal", bool(torch.equal(a, H)), g.last_kernel())
PY
cat $O/c8_diff.txt
echo "all done"
[ $rc -eq 0 ]
