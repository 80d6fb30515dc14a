import sys, os, torch
sys.path[:0] = ["/root/repo", "/root/repo/gnn-tf_amd"]
import gnntf
n, F, O = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = torch.randn(n, F, device="cuda"); W = torch.randn(F, O, device="cuda"); b = torch.randn(1, O, device="cuda")
for _ in range(6): gnntf.dense(X, W, b, relu=True)
torch.cuda.synchronize()
