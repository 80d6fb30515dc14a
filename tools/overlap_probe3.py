#!/usr/bin/env python3
"""Which streams run beside the compute stream?  One fresh process:
 A. 12 fresh streams, each timed with a 1 GiB device copy beside matrix products of the default stream;
 B. RCCL send / recv to itself (one-rank group) issued from each of those streams (the exchange lane's role);
 C. the same through a few further process groups (torch picks another stream for each group's RCCL kernels), lane fixed to the
    best stream of B.
Prints one JSON object; run under different GPU_MAX_HW_QUEUES to see the hardware-queue multiplexing."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch, torch.distributed as dist
fd = os.dup(1); os.dup2(2, 1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
order = os.environ.get("PROBE_ORDER", "streams_first")
if order == "group_first":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
A = torch.randn(4096, 4096, device=dev); B = torch.randn(4096, 4096, device=dev); C = torch.empty_like(A)
src = torch.empty(256 << 20, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)


def products():
    for _ in range(8):
        torch.mm(A, B, out=C)


def ms(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def exposed(transfer):
    t_c, t_x = ms(products), ms(transfer)
    t_b = ms(lambda: (transfer(), products()))
    return round(max(t_b - t_c, 0) / t_x, 2)


streams = [torch.cuda.Stream(dev) for _ in range(12)]
out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "order": order}


def copy_on(s):
    def f():
        with torch.cuda.stream(s):
            dst.copy_(src); dst.copy_(src)
    return f


out["A_copy_exposed_by_stream"] = [exposed(copy_on(s)) for s in streams]
if order != "group_first":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)


def rccl_on(s, group=None):
    def f():
        with torch.cuda.stream(s):
            for _ in range(2):
                for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0, group), dist.P2POp(dist.isend, src, 0, group)]):
                    req.wait()
    return f


rccl_on(streams[0])(); torch.cuda.synchronize()
out["B_rccl_exposed_by_lane_stream"] = b = [exposed(rccl_on(s)) for s in streams]
lane = streams[min(range(len(b)), key=lambda i: b[i])]
groups = [None] + [dist.new_group([0], backend="nccl") for _ in range(5)]
res = []
for g in groups:
    rccl_on(lane, g)(); torch.cuda.synchronize()
    res.append(exposed(rccl_on(lane, g)))
out["C_rccl_exposed_by_group_best_lane"] = res
worst = streams[max(range(len(b)), key=lambda i: b[i])]
out["C_rccl_exposed_by_group_worst_lane"] = [exposed(rccl_on(worst, g)) for g in groups]
# compute on a side stream instead of the default stream, exchange lane = each stream
comp = torch.cuda.Stream(dev)


def exposed_on(transfer):
    def prod():
        with torch.cuda.stream(comp):
            products()
    t_c, t_x = ms(prod), ms(transfer)
    t_b = ms(lambda: (transfer(), prod()))
    return round(max(t_b - t_c, 0) / t_x, 2)


out["D_rccl_exposed_by_lane_compute_on_side_stream"] = [exposed_on(rccl_on(s)) for s in streams]
dist.destroy_process_group()
os.write(fd, (json.dumps(out) + "\n").encode())
