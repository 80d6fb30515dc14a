#!/usr/bin/env python3
"""Which HIP streams' kernels run BESIDE a saturating kernel of the default stream?  Streams are multiplexed onto a few hardware
queues; this probes (a) plain copies issued from the n-th stream created, n = 0..7, (b) RCCL send / recv pairs (one-rank group as
its own peer) after creating k dummy streams before the process group, k given on the command line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch, torch.distributed as dist

k_dummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
json_fd = os.dup(1); os.dup2(2, 1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dummies = [torch.cuda.Stream(dev) for _ in range(k_dummy)]
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(1, device=dev)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
A = torch.randn(4096, 4096, device=dev); B = torch.randn(4096, 4096, device=dev); Cm = torch.empty_like(A)
a = torch.empty((1 << 30) // 4, dtype=torch.float32, device=dev).normal_(); b = torch.empty_like(a)

def matmuls(n=12):
    for _ in range(n):
        torch.mm(A, B, out=Cm)

def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

def rccl(stream, n=2):
    with torch.cuda.stream(stream):
        for _ in range(n):
            for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, a, 0)]):
                req.wait()

def copies(stream, n=2):
    with torch.cuda.stream(stream):
        for _ in range(n):
            b.copy_(a)

out = {"dummy_streams_before_nccl": k_dummy, "matmul_alone_ms": timed(matmuls)}
side = torch.cuda.Stream(dev)
out["rccl_alone_ms"] = timed(lambda: rccl(side))
out["rccl_with_matmul_ms"] = timed(lambda: (rccl(side), matmuls()))
streams = [torch.cuda.Stream(dev) for _ in range(8)]
out["copy_alone_ms"] = timed(lambda: copies(streams[0]))
out["copy_from_stream_n_with_matmul_ms"] = [round(timed(lambda st=st: (copies(st), matmuls())), 3) for st in streams]
dist.destroy_process_group()
os.write(json_fd, (json.dumps(out) + "\n").encode())
