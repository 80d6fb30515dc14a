#!/usr/bin/env python3
"""Training-mode micro-benchmark for rocprofv3: per iteration  dropout+renormalise -> fused step -> backward
(C=64 on the config-4 graph)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
dev = torch.device("cuda:0")
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H0 = (torch.rand(g.n_rows, C, device=dev) * 2 - 1).requires_grad_()
gout = torch.rand(g.n_rows, C, device=dev)
make = lambda k, bwd=False: gnntf.normalize(g, "symmetric", "none", dropout=0.5, seed=1, stream_id=k, transposed_only=bwd)
for it in range(3):
    H0.grad = None
    out = gnntf.ppr_loop(make, H0, 0.1, 10)
    out.backward(gout)
torch.cuda.synchronize()
print("ok")
