#!/usr/bin/env python3
"""The GCNII layer at C = 128: gnx_gcnii_step (one launch with GNX_GCNII_DMA=1 in the tuning build, else SpMM+mix then the dense kernel)
against a float64 restatement on a sample of rows (incl. hub rows and rows without entries), inference and with the mixed rows kept."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
from gnntf import sparse

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--entries", type=int, default=10_000_000)
ap.add_argument("--time", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=a.nodes, entries=a.entries), dev)
n, C, alpha = g.n_rows, 128, 0.1
gen = torch.Generator(device=dev).manual_seed(1)
H, H0 = torch.rand(n, C, device=dev, generator=gen) * 2 - 1, torch.rand(n, C, device=dev, generator=gen) * 2 - 1
M = 0.6 * torch.eye(C, device=dev) + 0.4 * torch.randn(C, C, device=dev, generator=gen) / 8
res = {"nodes": n, "entries": a.entries, "dma": os.environ.get("GNX_GCNII_DMA", "0")}
with torch.no_grad():
    out = gnntf.gcnii_step(adj, H, H0, alpha, M, relu=True)
    res["kernel"] = g.last_kernel()
    mixed_ref = gnntf.ppr_step(adj, H, H0, alpha)                       # the fused SpMM+mix (its own parity tests cover it)
    deg = torch.diff(g.rowptr_tensor()) if hasattr(g, "rowptr_tensor") else None
    rows = torch.cat([torch.arange(0, 4096, device=dev), torch.randint(0, n, (16384,), device=dev, generator=gen), torch.arange(n - 4096, n, device=dev)])
    want = torch.relu(mixed_ref[rows].double() @ M.double())
    err = (out[rows].double() - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    res["max_rel_err_vs_f64_on_sample"] = err
    full = torch.relu(mixed_ref @ M)
    res["max_abs_diff_vs_torch_f32_all_rows"] = (out - full).abs().max().item()
    res["nonfinite"] = int((~torch.isfinite(out)).sum().item())
    # the training form: the same launch also writes the mixed rows
    out2, mixed = sparse._gcnii_launch(adj, H, H0, alpha, M, True, True)
    res["kept_mixed_max_abs_diff"] = (mixed - mixed_ref).abs().max().item()
    res["kept_out_equal"] = bool(torch.equal(out2, out))
    if a.time:
        res["ms"] = bench.median_ms(lambda: gnntf.gcnii_step(adj, H, H0, alpha, M, relu=True), reps=7, warm=2)
        res["ms_keep_mixed"] = bench.median_ms(lambda: sparse._gcnii_launch(adj, H, H0, alpha, M, True, True), reps=5, warm=1)
print(json.dumps(res), flush=True)
