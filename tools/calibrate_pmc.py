#!/usr/bin/env python3
"""Known-byte-count launch for calibrating FETCH_SIZE / WRITE_SIZE on this kernel's access pattern
(MI355X_MICROARCH.md "HBM": calibrate before trusting an absolute).  The graph is a random
permutation matrix: every row has one entry and every H row is gathered exactly once, so one launch
reads  N*(8 rowptr + 4 col + 4 val + 4C gathered + 4C H0)  and writes  N*4C  bytes, with no reuse.
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; prints the expected byte counts."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

import gnntf
from gnntf.sparse import _launch

n, C = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
perm = torch.randperm(n, device=dev, generator=gen)
idx = torch.stack([torch.arange(n, device=dev), perm], 1)
g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, torch.ones(n, device=dev), (n, n)), device=dev)
adj = gnntf.Adjacency(g)
H = torch.rand(n, C, device=dev)
H0 = torch.rand(n, C, device=dev)
out = torch.empty_like(H)
for _ in range(5):
    _launch(adj, H, H0, 0.9, 0.1, 0, out=out)
torch.cuda.synchronize()
assert torch.allclose(out[:1000], 0.9 * H[perm[:1000]] + 0.1 * H0[:1000], rtol=1e-6)
print(json.dumps({"kernel": g.last_kernel(), "C": C, "expected_read_bytes": n * (8 + 4 + 4 + 8 * C), "expected_write_bytes": n * 4 * C}))
