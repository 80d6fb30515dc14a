import sys, os, torch, time
sys.path[:0] = ["/root/repo", "/root/repo/gnn-tf_amd"]
import gnntf
for (n,F,O) in [(10_000_000,256,64),(10_000_000,128,128),(10_000_000,64,256),(10_000_000,128,64)]:
    X = torch.randn(n, F, device="cuda"); W = torch.randn(F, O, device="cuda"); b = torch.randn(1, O, device="cuda")
    for _ in range(3): gnntf.dense(X, W, b, relu=True)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(10): gnntf.dense(X, W, b, relu=True)
    torch.cuda.synchronize(); ms=(time.time()-t0)*100
    print(os.environ.get("GNX_DENSE_RING", "default"), n,F,O, round(ms,3),"ms", round(2*n*F*O/ms/1e9,1),"TF", round(4*n*(F+O)/ms/1e6,1), "GB/s", flush=True)
    del X, W, b
