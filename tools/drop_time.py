import sys, argparse, torch, time
sys.path[:0] = ["/root/repo", "/root/repo/gnn-tf_amd"]
import bench, gnntf
from gnntf.sparse import _launch
dev = torch.device("cuda:0")
g, adj, _ = bench.build_single(argparse.Namespace(nodes=10_000_000, entries=100_000_000), dev)
C = 64
X = torch.rand(g.n_rows, C, device=dev); H0 = torch.rand(g.n_rows, C, device=dev); out = torch.empty_like(X)
fused = gnntf.sparse.dropped_adjacency(g, 0.5, 1, 3)
for tr in (False, True):
    for _ in range(3): _launch(fused, X, H0, 0.9, 0.1, 0, transposed=tr, out=out)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): _launch(fused, X, H0, 0.9, 0.1, 0, transposed=tr, out=out)
    torch.cuda.synchronize(); print("transposed" if tr else "forward", round((time.time() - t0) * 100, 3), "ms", g.last_kernel())
for _ in range(3): _launch(adj, X, H0, 0.9, 0.1, 0, out=out)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): _launch(adj, X, H0, 0.9, 0.1, 0, out=out)
torch.cuda.synchronize(); print("eval", round((time.time() - t0) * 100, 3), "ms")
