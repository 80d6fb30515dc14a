#!/usr/bin/env python3
"""Does an RCCL point-to-point transfer run CONCURRENTLY with a bandwidth-bound kernel of this library on another stream?
One-rank "nccl" group as its own peer (send / recv pairs = device-local copies through RCCL's kernels), the library's read-only
stream kernel as the compute load.  Prints the times alone and together; together ~ max means overlap, ~ sum means none."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch, torch.distributed as dist
from gnntf import _native as nat

json_fd = os.dup(1); os.dup2(2, 1)                       # RCCL prints a banner on stdout
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
lib = nat.lib()
src = torch.empty((8 << 30) // 4, dtype=torch.float32, device=dev).normal_()
sink = torch.zeros(64, device=dev)
a = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev).normal_()      # 1 GiB message
b = torch.empty_like(a)
side = torch.cuda.Stream(dev)

def compute(n=8):
    for _ in range(n):
        nat.check(lib.gnx_stream_read(nat.ptr(src), src.numel(), nat.ptr(sink), nat.current_stream()))

def transfer(kind, n=2):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind == "rccl":
                for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, a, 0)]):
                    req.wait()
            else:
                b.copy_(a)

def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

A = torch.randn(4096, 4096, device=dev); B = torch.randn(4096, 4096, device=dev); Cm = torch.empty_like(A)

def matmuls(n=12):                                         # matrix-core bound, cache resident: no competition for HBM
    for _ in range(n):
        torch.mm(A, B, out=Cm)

out = {"stream_read_alone_ms": timed(compute), "matmul_alone_ms": timed(matmuls)}
for kind in ("copy", "rccl"):
    out[f"{kind}_alone_ms"] = timed(lambda: transfer(kind))
    out[f"{kind}_with_stream_read_ms"] = timed(lambda: (transfer(kind), compute()))
    out[f"stream_read_then_{kind}_issue_order_ms"] = timed(lambda: (compute(), transfer(kind)))
    out[f"{kind}_with_matmul_ms"] = timed(lambda: (transfer(kind), matmuls()))
    out[f"matmul_then_{kind}_issue_order_ms"] = timed(lambda: (matmuls(), transfer(kind)))
# (tried: the compute load on a CU-masked stream, hipExtStreamCreateWithCUMask leaving 16 / 32 / 64 CUs free: the matmuls take twice
#  as long on any mask and RCCL's kernel still adds its whole duration -- not a way)
# a HIGH-PRIORITY side stream: are its kernels dispatched beside a saturating kernel of the default stream?
hi = torch.cuda.Stream(dev, priority=-1)
def copy_hi(n=2):
    with torch.cuda.stream(hi):
        for _ in range(n):
            b.copy_(a)
out["copy_on_high_priority_stream_with_matmul_ms"] = timed(lambda: (copy_hi(), matmuls()))
out["matmul_then_copy_on_high_priority_stream_ms"] = timed(lambda: (matmuls(), copy_hi()))
out["matmul_then_copy_on_high_priority_stream_with_stream_read_ms"] = timed(lambda: (compute(), copy_hi()))
out["torch_nccl_high_priority_env"] = os.environ.get("TORCH_NCCL_HIGH_PRIORITY", "")
dist.destroy_process_group()
os.write(json_fd, (json.dumps(out, indent=1) + "\n").encode())
