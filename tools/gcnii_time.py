#!/usr/bin/env python3
"""The GCNII layer on the config-4 graph: one fused launch vs SpMM+mix followed by the dense kernel, inference and training."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf

dev = torch.device("cuda:0")
gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
n, a = g.n_rows, 0.1
for C in (32, 64, 128):
    H, H0 = torch.rand(n, C, device=dev) * 2 - 1, torch.rand(n, C, device=dev) * 2 - 1
    M = 0.6 * torch.eye(C, device=dev) + 0.4 * torch.randn(C, C, device=dev) / 8
    with torch.no_grad():
        t_fused = bench.median_ms(lambda: gnntf.gcnii_step(adj, H, H0, a, M, relu=True), reps=5, warm=2)
        kernel = g.last_kernel()
        t_two = bench.median_ms(lambda: gnntf.dense(gnntf.ppr_step(adj, H, H0, a), M, None, relu=True), reps=5, warm=2)
        t_spmm = bench.median_ms(lambda: gnntf.ppr_step(adj, H, H0, a), reps=5, warm=2)
    Ht, H0t, Mt = H.clone().requires_grad_(), H0.clone().requires_grad_(), M.clone().requires_grad_()
    up = torch.rand(n, C, device=dev)

    def train(fused, backward=True):
        for t in (Ht, H0t, Mt):
            t.grad = None
        out = gnntf.gcnii_step(adj, Ht, H0t, a, Mt, relu=True) if fused else gnntf.dense(gnntf.ppr_step(adj, Ht, H0t, a), Mt, None, relu=True)
        if backward:
            out.backward(up)
    res = {"C": C, "kernel": kernel, "inference_fused_ms": t_fused, "inference_two_launch_ms": t_two, "spmm_mix_alone_ms": t_spmm,
           "train_forward_fused_ms": bench.median_ms(lambda: train(True, False), reps=3, warm=1),
           "train_forward_two_launch_ms": bench.median_ms(lambda: train(False, False), reps=3, warm=1),
           "train_fwd_bwd_fused_ms": bench.median_ms(lambda: train(True), reps=3, warm=1),
           "train_fwd_bwd_two_launch_ms": bench.median_ms(lambda: train(False), reps=3, warm=1)}
    print(json.dumps(res), flush=True)
    del H, H0, Ht, H0t, Mt, up
    torch.cuda.empty_cache()
