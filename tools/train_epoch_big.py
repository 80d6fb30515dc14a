#!/usr/bin/env python3
"""One full APPNP training epoch (filter.py:25-35 defaults: Dropout, Dense 64 relu, Dense C, K = 10 with edge dropout 0.5) through
train() on an R-MAT graph with dense random features -- where does an epoch of the WHOLE model go at scale?  Prints ms per epoch;
run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=4_000_000)
ap.add_argument("--entries", type=int, default=40_000_000)
ap.add_argument("--features", type=int, default=128)
ap.add_argument("--classes", type=int, default=40)
ap.add_argument("--epochs", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=a.nodes, entries=a.entries), dev)
n = g.n_rows
X = torch.randn(n, a.features, device=dev)
labels = torch.randint(0, a.classes, (n,), device=dev)
nodes = torch.randperm(n, device=dev)
train, valid = nodes[: n // 10], nodes[n // 10: n // 5]
gnntf.set_seed(0)
model = gnntf.APPNP(g, X, num_classes=a.classes)
task = lambda idx: gnntf.NodeClassification(idx, labels[idx])
model.train(train=task(train), valid=task(valid), epochs=2, patience=100)
torch.cuda.synchronize(); t0 = time.time()
model.train(train=task(train), valid=task(valid), epochs=a.epochs, patience=100)
torch.cuda.synchronize()
print(json.dumps({"nodes": n, "entries": a.entries, "features": a.features, "classes": a.classes, "ms_per_epoch_incl_validation": (time.time() - t0) / a.epochs * 1e3}))
