"""World-size-2 (and 3) gloo runs of the vertex-partitioned propagation on CPU ranks: partition,
halo plan, pairwise exchange and ping-pong logic of gnntf.sharded, checked against the
single-process oracle."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, mode, device="cpu"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), mode, device]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "OK " + mode in res.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_matches_single_process_oracle(world):
    launch(world, "slices")


def test_distributed_generator_world2():
    launch(2, "rmat")


@pytest.mark.parametrize("world,grid", [(2, "grid1x2"), (4, "grid2x2")])
def test_feature_sliced_grid(world, grid):
    """pv vertex blocks x pf feature slices: slices need no exchange (columns propagate independently)."""
    launch(world, grid)


def test_choose_grid():
    """Cost model of DESIGN.md section 5: wide features -> slices (no exchange); medium -> vertex blocks."""
    from gnntf.sharded import choose_grid, grid_cost_ms
    for world in (2, 4, 8):
        assert choose_grid(world, 256, 10_000_000 * world, 100_000_000 * world) == (1, world)
    assert choose_grid(8, 128, 80_000_000, 800_000_000) == (1, 8)
    assert choose_grid(8, 64, 80_000_000, 800_000_000) == (8, 1)
    assert choose_grid(8, 7, 80_000_000, 800_000_000) == (8, 1)          # 7 columns cannot be sliced
    assert choose_grid(1, 256) == (1, 1)
    assert choose_grid(8, 256, 80_000_000, 800_000_000, link_GBs=1e6) == (8, 1)   # free links -> keep rows wide
    assert abs(grid_cost_ms(1, 8, 256, 80_000_000, 800_000_000) - 20.0) < 1.0      # measured: 22.3 ms
    assert abs(grid_cost_ms(1, 1, 256, 10_000_000, 100_000_000) - 16.2) < 0.5      # measured: 16.2 ms


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["slices", "rmat", "grid1x2"])
def test_sharded_native_backend_two_ranks_one_gpu(mode):
    """The libgnx.so backend on real shards (rectangular local CSR, halo columns, output view inside
    the ping-pong buffer): two ranks share cuda:0 and exchange halos over gloo (staged through the host;
    RCCL needs one GPU per rank, which the single-GPU box cannot give)."""
    launch(2, mode, "cuda")
