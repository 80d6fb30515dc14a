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
