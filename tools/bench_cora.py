#!/usr/bin/env python3
"""Config 2 timing (Cora-shaped APPNP, K=10, C=7): latency-bound regime.  ms per training epoch
(layer-by-layer and fused, eager and replayed from hipGraphs) and eval forward latency on the HIP path."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import gnntf, graphs

gnntf.set_default_device("cuda:0")
coo, vals, shape, X = graphs.cora_shaped(seed=0)
labels = np.random.default_rng(0).integers(0, 7, size=shape[0])
train = list(range(140)); valid = list(range(140, 640))
out = {}
for fused in (False, True):
    gnntf.set_seed(0)
    model = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), X, num_classes=7, fused=fused)
    model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]), epochs=5, patience=5)
    torch.cuda.synchronize(); t0 = time.time()
    model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]), epochs=100, patience=1000)
    torch.cuda.synchronize()
    out["train_ms_per_epoch_fused" if fused else "train_ms_per_epoch_layers"] = (time.time() - t0) * 10
    with torch.no_grad():
        for _ in range(5): model(model.features)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(100): model(model.features)
        torch.cuda.synchronize()
    out["eval_forward_ms_fused" if fused else "eval_forward_ms_layers"] = (time.time() - t0) * 10
for fused in (False, True):                                 # the same epochs replayed from hipGraphs (train(capture=True))
    gnntf.set_seed(0)
    model = gnntf.APPNP(gnntf.SparseCOO(coo, vals, shape), X, num_classes=7, fused=fused)
    model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]), epochs=5, patience=5, capture=True)
    times = []
    for epochs in (100, 300):
        torch.cuda.synchronize(); t0 = time.time()
        model.train(train=gnntf.NodeClassification(train, labels[train]), valid=gnntf.NodeClassification(valid, labels[valid]), epochs=epochs, patience=1000, capture=True)
        torch.cuda.synchronize(); times.append(time.time() - t0)
    out["captured_train_ms_per_epoch_fused" if fused else "captured_train_ms_per_epoch_layers"] = (times[1] - times[0]) / 200 * 1e3   # capture cost cancels
    out["capture_setup_ms_fused" if fused else "capture_setup_ms_layers"] = (times[0] - 100 * (times[1] - times[0]) / 200) * 1e3
print(json.dumps(out, indent=1))
