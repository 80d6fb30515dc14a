"""gnx_halo_pack / gnx_halo_exchange over a real RCCL communicator, from a plain C program (tests/c_abi_rccl.c).
On a one-GPU box the block lists itself as its only peer, so a one-rank communicator carries real ncclSend / ncclRecv pairs;
with two GPUs the two vertex blocks of P4 run as two processes."""
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "c_abi_rccl")
    lib = os.path.join(ROOT, "gnn-tf_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-D_DEFAULT_SOURCE", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_rccl.c"), "-L", lib, "-lgnx", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-lm",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_halo_exchange_over_rccl_loop_back(tmp_path):
    """Pulled rows and pushed partial sums through ncclSend / ncclRecv (one group, then two groups with bound entry points)."""
    res = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0 and "RCCL loop-back OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2 or os.environ.get("GNX_TEST_TWO_GPUS") != "1",
                    reason="needs two GPUs and GNX_TEST_TWO_GPUS=1 (the two-process RCCL client has only ever run where two GPUs were at hand)")
def test_halo_exchange_over_rccl_two_gpus(tmp_path):
    exe, idfile = build(tmp_path), str(tmp_path / "nccl_id")
    ranks = [subprocess.Popen([exe, "2", str(r), idfile], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, proc in enumerate(ranks):
        out, _ = proc.communicate(timeout=240)
        assert proc.returncode == 0 and "RCCL two-rank exchange OK" in out, (r, out[-2000:])


NCCL_SELF = r'''
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
sys.path[:0] = [sys.argv[2], os.path.join(sys.argv[2], "gnn-tf_amd")]
from gnntf import sharded

class SelfPeer(sharded.Comm):                      # a one-rank group that exchanges with itself: what the N > 1 path does per peer
    def __init__(self):
        self.group, self.solo, self.rank, self.size = None, False, -1, 2      # rank -1: "rank 0" is a peer like any other

comm = SelfPeer()
lanes = sharded._Lanes(dev)
a1, a2 = torch.arange(3000., device=dev).reshape(-1, 4), -torch.arange(5000., device=dev).reshape(-1, 4)
b1, b2 = torch.zeros_like(a1), torch.zeros_like(a2)
# both halves of one peer's message in ONE batch (pulled rows, then pushed sums): matched in order
comm.exchange_pairs([(0, a1), (0, a2)], [(0, b1), (0, b2)])
torch.cuda.synchronize()
assert torch.equal(a1, b1) and torch.equal(a2, b2), "two messages to one peer in one batch were not matched in order"
# the same on the exchange lane behind an event, as propagate() issues it (early_pull: two batches back to back)
c1, c2 = torch.zeros_like(a1), torch.zeros_like(a2)
x = a1 * 2
ready = lanes.mark()
with lanes.exchange_lane():
    lanes.wait(ready, on_exchange_lane=True)
    comm.exchange_pairs([(0, x)], [(0, c1)])
    comm.exchange_pairs([(0, a2)], [(0, c2)])
    arrived = lanes.mark(on_exchange_lane=True)
lanes.wait(arrived)
y = c1 + 1
torch.cuda.synchronize()
assert torch.equal(y, a1 * 2 + 1) and torch.equal(c2, a2)
# placement of the exchange: the probe must end on a lane stream and a group whose transfers hide beside the compute stream's kernels
real = sharded.Comm(group=None)
table = real.tune_overlap(dev, force=True)
print("overlap probe:", table)
chosen = [rec for rec in table if rec["chosen"]]
# correctness only: exactly one placement is chosen, it is the least exposed one the probe saw, and the status record says so.
# HOW exposed it is (measured: 0.5-0.6 when placed well, 0.85-1.1 when a hardware queue is shared) is a timing on a shared box:
# reported, not asserted
assert len(chosen) == 1, table
print("overlap probe: exposed share of the chosen placement =", round(chosen[0]["exposed_share"], 3), "status:", real.overlap_status)
assert real.overlap_status["ran"] and real.overlap_status["groups_tried"] >= 1
assert chosen[0]["exposed_share"] <= min(rec["exposed_share"] for rec in table) + 0.2 + 1e-9      # (ties within 0.1 keep the first lane / group)
assert real.lane_stream is not None
x2, y2 = torch.arange(64., device=dev), torch.zeros(64, device=dev)
real.size, real.rank = 2, -1                      # (as SelfPeer: exchange with "rank 0" = itself, over the group the probe settled on)
with torch.cuda.stream(real.lane_stream):
    real.exchange_pairs([(0, x2)], [(0, y2)])
torch.cuda.synchronize()
assert torch.equal(x2, y2)
dist.destroy_process_group()
print("NCCL SELF OK")
'''


def test_torch_rccl_point_to_point_batches(tmp_path):
    """The transport of the N > 1 path on real RCCL, as far as one GPU can take it: torch.distributed's "nccl" backend (= RCCL) with
    a one-rank group whose only peer is itself -- Comm.exchange_pairs with TWO messages to the same peer in one batch (the pulled
    and the pushed half of a region), and two batches back to back on the exchange lane behind events (early_pull)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "nccl_self.py"
    script.write_text(NCCL_SELF)
    res = subprocess.run([__import__("sys").executable, str(script), str(port), ROOT], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0 and "NCCL SELF OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
