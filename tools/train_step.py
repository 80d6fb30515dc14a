#!/usr/bin/env python3
"""Training-mode micro-benchmark for rocprofv3: K = 10 PPR iterations forward + backward with per-iteration edge dropout +
renormalisation on the config-4 graph.   python tools/train_step.py [C] [fused|two_pass]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
dev = torch.device("cuda:0")
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
H0 = (torch.rand(g.n_rows, C, device=dev) * 2 - 1).requires_grad_()
gout = torch.rand(g.n_rows, C, device=dev)
kept = {}
def fused(k, bwd=False):
    if k not in kept:
        kept[k] = gnntf.sparse.dropped_adjacency(g, 0.5, 1, k)
    return kept.pop(k) if bwd else kept[k]
two_pass = lambda k, bwd=False: gnntf.normalize(g, "symmetric", "none", dropout=0.5, seed=1, stream_id=k, transposed_only=bwd)
make = fused if mode == "fused" else two_pass
for it in range(3):
    H0.grad = None
    out = gnntf.ppr_loop(make, H0, 0.1, 10)
    out.backward(gout)
torch.cuda.synchronize()
print("ok", mode)
