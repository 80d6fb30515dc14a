"""A/B of the dense kernels at 10M x 256 -> 64 in ONE process: W-in-registers (default), LDS-DMA ring (GNX_DENSE_WREG=0) and the
register-staged kernel (also GNX_DENSE_RING=0) are separate processes of the tuning build; this script times whichever the
environment selects and checks the result against a float64 product on a sample of rows."""
import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import gnntf
n, F, O = int(float(os.environ.get("N", 10_000_000))), int(os.environ.get("F", 256)), int(os.environ.get("O", 64))
X = torch.randn(n, F, device="cuda"); W = torch.randn(F, O, device="cuda"); b = torch.randn(1, O, device="cuda")
out = gnntf.dense(X, W, b, relu=True)
rows = torch.cat([torch.arange(0, 4096, device="cuda"), torch.randint(0, n, (8192,), device="cuda"), torch.arange(n - 4096, n, device="cuda")])
ref = torch.relu(X[rows].double() @ W.double() + b.double())
err = float((out[rows].double() - ref).abs().max())
for _ in range(3): gnntf.dense(X, W, b, relu=True)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): gnntf.dense(X, W, b, relu=True)
torch.cuda.synchronize(); ms = (time.time() - t0) * 50
print(dict(wreg=os.environ.get("GNX_DENSE_WREG", "on"), ring=os.environ.get("GNX_DENSE_RING", "on"), n=n, F=F, O=O, ms=round(ms, 3), TF=round(2 * n * F * O / ms / 1e9, 1),
           GBs=round(4 * n * (F + O) / ms / 1e6, 1), max_err_vs_f64=err), flush=True)
