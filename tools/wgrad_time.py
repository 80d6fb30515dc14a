"""gnx_dense_wgrad (dW = X^T . G over n rows) at the shapes the training paths use; ms, TF, GB/s, error against float64 on a row sample
restated as a full float64 product on the device, and torch's own X.t() @ G beside it."""
import sys, os, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import gnntf
from gnntf import sparse
n = int(float(os.environ.get("N", 10_000_000)))
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "256x64,128x128,64x64,128x64,256x256,64x40").split(",")]
for F, O in shapes:
    X = torch.randn(n, F, device="cuda"); G = torch.randn(n, O, device="cuda")
    got = sparse._dense_wgrad(X, G)
    m = min(n, 200_000)
    ref = sum((X[i:i + m].double().t() @ G[i:i + m].double()) for i in range(0, n, m))
    err = float(((got.double() - ref).abs().max()) / ref.abs().max())
    again = sparse._dense_wgrad(X, G)
    def ms(fn, reps=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e3
    t = ms(lambda: sparse._dense_wgrad(X, G))
    tt = ms(lambda: X.t() @ G, reps=3)
    print(dict(n=n, F=F, O=O, ms=round(t, 3), TF=round(2 * n * F * O / t / 1e9, 1), GBs=round(4 * n * (F + O) / t / 1e6, 1), torch_ms=round(tt, 3),
               rel_err=err, repeatable=bool(torch.equal(got, again))), flush=True)
    del X, G
