"""World-size-2/3/4 gloo runs of the vertex-partitioned propagation on CPU ranks: partition, pull/push halo
plan (vertex cover), pairwise exchange, interior/boundary split, column-chunk pipelining and the ping-pong
logic of gnntf.sharded, checked against the single-process oracle."""
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


def launch(world, mode, device="cpu", options="cover,split,2"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), mode, device, options]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "OK " + mode in res.stdout


@pytest.mark.parametrize("world,options", [(2, "cover,split,2"), (3, "cover,split,2"), (2, "pull,whole,1"), (3, "pull,split,3"),
                                           (2, "cover@0.25,split,2"), (3, "cover@3,whole,2")])
def test_sharded_matches_single_process_oracle(world, options):
    """Fixed graph with duplicates, uneven blocks: cover (plain and weighted: "cover@w") and pull-only plans, split and whole
    rows, 1-3 chunks."""
    launch(world, "slices", options=options)


def test_weighted_cover_runs_from_the_minimum_cover_to_the_plain_halo():
    """cover_push_mask(push_weight=w): w = 0 is the minimum-rows cover; growing w sums fewer entries on the sender's side for more
    rows on the link, towards the pull-only halo (in the limit only rows whose columns no other row references stay pushed:
    one row on the link instead of several, for the same work).  Any mask is a valid plan: every cross entry is either pushed
    or its column pulled, local entries are never pushed."""
    import numpy as np
    import torch
    from gnntf import sharded
    import graphs
    n, world, rank = 4000, 4, 1
    coo, _, _ = graphs.rmat_symmetric_coo(n, 60000, seed=9)
    bounds = sharded.uniform_bounds(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    mine = np.unique(coo[(coo[:, 0] >= lo) & (coo[:, 0] < hi)], axis=0)
    row, col = torch.from_numpy(mine[:, 0] - lo), torch.from_numpy(mine[:, 1])
    bnd = torch.tensor(bounds[1:], dtype=torch.int64)
    owner = torch.bucketize(col, bnd, right=True)
    remote = owner != rank

    def plan(w):
        push = sharded.cover_push_mask(row, col, owner, rank, hi - lo, bnd, w)
        assert not bool(push[~remote].any())
        pulled_cols = int(torch.unique(col[remote & ~push]).numel())
        pushed_rows = int(torch.unique(owner[push] * n + row[push]).numel())
        return int(push.sum()), pulled_cols + pushed_rows
    halo = int(torch.unique(col[remote]).numel())
    table = [plan(w) for w in (0.0, 0.05, 0.3, 2.0, 1e9)]
    assert table[0] == plan(0) and table[0][1] < 0.8 * halo                  # the plain cover: well below the halo on a power-law graph
    assert table[-1][0] < 0.7 * table[0][0] and table[-1][1] > table[0][1]    # w -> infinity: the push shrinks, part of the halo is back
    pushed = [t[0] for t in table]
    rows = [t[1] for t in table]
    assert all(a >= b for a, b in zip(pushed, pushed[1:])) and pushed[1] < pushed[0]      # fewer entries summed on the sender ...
    assert all(a <= b for a, b in zip(rows, rows[1:]))                                    # ... for more rows on the link
    assert all(r <= halo for r in rows)                                                   # never more than the plain halo


@pytest.mark.parametrize("world,options", [(2, "cover,split,2"), (3, "cover,whole,2")])
def test_sharded_directed_pattern(world, options):
    """Asymmetric pattern: column sums differ from row sums, rows to send need not be boundary rows, and the
    greedy cover can lose to the plain halo for a peer (which then keeps the plain halo)."""
    launch(world, "directed", options=options)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_strong_scaling_generator(world):
    """bench.py's N > 1 workload: ONE global graph cut into `world` vertex blocks (rank 0 generates, broadcast); world 8 is
    the layout of BASELINE config 5 (seven peer regions around the local rows, per-peer pull / push lists)."""
    launch(world, "blocks")


def test_strong_scaling_generator_replicated():
    """The same workload with every rank generating the edge list itself (what bench.py's gloo rehearsals do instead of an 8 GB
    host-staged broadcast): the ranks compare checksums of their lists, and the blocks are the blocks of the broadcast form."""
    launch(3, "blocks_replicated")


def test_distributed_generator_world2():
    launch(2, "rmat")


@pytest.mark.parametrize("world,grid", [(2, "grid1x2"), (4, "grid2x2")])
def test_feature_sliced_grid(world, grid):
    """pv vertex blocks x pf feature slices: slices need no exchange (columns propagate independently)."""
    launch(world, grid)


@pytest.mark.parametrize("world", [2, 3])
def test_training_on_vertex_blocks(world):
    """architecture.train() with one vertex block per rank (symmetric constant adjacency): the backward of the K loop is the
    same sharded propagation applied to the gradient, parameter gradients are summed over the ranks, the task reports global
    losses -- same parameters as single-process dense float64 training."""
    launch(world, "train")


@pytest.mark.parametrize("world,mode", [(1, "dropout"), (2, "dropout"), (3, "dropout"), (4, "dropout"), (3, "dropout_directed")])
def test_edge_dropout_on_vertex_blocks(world, mode):
    """Training-mode propagation with per-iteration edge dropout + re-normalisation across blocks (last bullet):
    masks keyed by global (row, col), global column sums through the reversed halo exchange, backward through A_k^T with the
    halo rows of the gradient returned to their owners -- forward and dH0 equal the single-process oracle for every world size."""
    launch(world, mode)


@pytest.mark.parametrize("world", [2, 3])
def test_training_with_edge_dropout_on_vertex_blocks(world):
    """architecture.train() over vertex blocks with APPNP's default graph_dropout = 0.5: same parameters as single-process
    dense float64 training that rebuilds every iteration's dropped adjacency from the oracle."""
    launch(world, "train_dropout")


@pytest.mark.parametrize("world", [2, 3])
def test_gcn_on_vertex_blocks(world):
    """GCNLayer (gcn.py:77-89) with its aggregation over vertex blocks (one halo exchange per layer, forward and backward):
    a 2-layer GCN trained on blocks ends with the parameters of single-process dense float64 training."""
    launch(world, "gcn")


@pytest.mark.parametrize("world", [2, 3])
def test_gcnii_on_vertex_blocks(world):
    """GCNIILayer (gcn.py:7-27) over vertex blocks: the aggregation (1-a) A_hat H + a H0 with H != H0 is one iteration of the
    block propagation started from the layer's input; three layers + the two dense ends, trained, against dense float64."""
    launch(world, "gcnii")


@pytest.mark.parametrize("world,cover", [(5, "cover"), (7, "pull")])
def test_blocks_as_threads_of_one_process(world, cover):
    """The ranks as threads exchanging through shared memory (tests/thread_comm.py, the harness of the GPU full-size block test):
    odd world sizes, checker backend, against the single-process oracle."""
    import numpy as np
    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd"), os.path.join(ROOT, "tests")]
    import graphs
    from dist_worker import OracleBackend
    from gnntf import sharded
    from oracle import gnntf_oracle as orc
    from thread_comm import run_ranks
    n = 1201
    coo, vals, _ = graphs.rmat_symmetric_coo(n, 11000, seed=3)
    H0 = np.random.default_rng(1).uniform(-1, 1, (n, 12)).astype(np.float32)
    bounds = sharded.uniform_bounds(n, world)

    def body(comm):
        lo, hi = bounds[comm.rank], bounds[comm.rank + 1]
        mine = (coo[:, 0] >= lo) & (coo[:, 0] < hi)
        sg = sharded.ShardedGraph(torch.from_numpy(coo[mine]), torch.from_numpy(vals[mine]), bounds, backend=OracleBackend(), comm=comm,
                                  cover=cover, chunks=3)
        return sg.propagate(sg.make_state(torch.from_numpy(H0[lo:hi].copy())), 0.1, 10).clone().numpy()
    got = np.concatenate(run_ranks(world, body))
    want = orc.appnp_propagate(coo, vals, (n, n), H0, a=0.1, iterations=10)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


def test_choose_grid_and_columns():
    """Planning helpers: the grid cost model takes a MEASURED link rate (no built-in default), and the column
    chunks are whole 128-byte lines wherever the width allows."""
    from gnntf.sharded import choose_grid, grid_cost_ms, split_columns
    big = dict(nodes_total=80_000_000, entries_total=1_000_000_000)
    assert choose_grid(8, 128, link_GBs=1e6, **big) == (8, 1)                  # free links -> keep rows wide
    assert choose_grid(8, 128, link_GBs=1.0, **big) == (1, 8)                  # very slow links -> replicate the graph instead
    assert choose_grid(8, 7, link_GBs=1.0, **big) == (8, 1)                    # 7 columns cannot be sliced
    assert choose_grid(1, 256, link_GBs=50.0, **big) == (1, 1)
    assert abs(grid_cost_ms(1, 1, 256, 10_000_000, 100_000_000, link_GBs=50.0) - 14.5) < 0.5      # measured (round 2): 14.5 ms
    with pytest.raises(TypeError):
        choose_grid(8, 128, 80_000_000, 1_000_000_000)                          # the link rate must be given
    assert split_columns(128, 2) == [(0, 64), (64, 128)]
    assert split_columns(128, 3) == [(0, 32), (32, 64), (64, 128)]
    assert split_columns(12, 2) == [(0, 4), (4, 12)] or split_columns(12, 2) == [(0, 8), (8, 12)]
    assert split_columns(7, 2) == [(0, 3), (3, 7)]
    assert split_columns(5, 9)[-1][1] == 5 and len(split_columns(5, 9)) == 5


def test_overlap_probe_states_why_it_did_not_run(monkeypatch):
    """Comm.tune_overlap records a result even when it bails out: never an exception for the caller to
    swallow, always ``overlap_status`` -- here the branches a CPU box can reach (the probing itself needs RCCL: tests/test_gpu_rccl.py)."""
    import torch
    from gnntf.sharded import Comm
    solo = Comm(solo=True)
    assert solo.tune_overlap(torch.device("cpu")) == [] and solo.overlap_status["ran"] is False
    assert "not applicable" in solo.overlap_status["reason"] and solo.lane_stream is None
    assert solo.tune_overlap(torch.device("cpu")) == []                       # once per communicator
    monkeypatch.setenv("GNX_TUNE_OVERLAP", "0")
    off = Comm(solo=True)
    off.tune_overlap(torch.device("cuda", 0))
    assert off.overlap_status == {"ran": False, "reason": "disabled (GNX_TUNE_OVERLAP=0)", "seconds": 0.0, "groups_tried": 0, "groups_destroyed": 0}


def test_cover_push_mask_properties():
    """The pull/push decision: every pushed entry's (peer, row) is pushed as a whole, a star's leaves are covered by
    its hub from both sides, and a peer whose cover is not smaller keeps the plain halo."""
    import torch
    from gnntf.sharded import cover_push_mask
    bnd = torch.tensor([10, 20])
    # rank 0 owns 0..9; row 0 is a hub with 6 columns on rank 1 (ids 10..15); rows 1..3 all reference column 19 (a remote hub)
    row = torch.tensor([0, 0, 0, 0, 0, 0, 1, 2, 3, 4])
    col = torch.tensor([10, 11, 12, 13, 14, 15, 19, 19, 19, 5])
    owner = torch.bucketize(col, bnd, right=True)
    push = cover_push_mask(row, col, owner, 0, 10, bnd)
    assert push.tolist() == [True] * 6 + [False] * 4          # the hub row is pushed (1 row instead of 6), column 19 pulled, local entry untouched
    # two rows sharing the same two remote columns: the plain halo (2 rows) is not beaten -> all pull
    row2 = torch.tensor([0, 0, 1, 1]); col2 = torch.tensor([10, 11, 10, 11])
    assert not cover_push_mask(row2, col2, torch.bucketize(col2, bnd, right=True), 0, 10, bnd).any()
    assert not cover_push_mask(row, col, torch.zeros_like(owner), 0, 10, torch.tensor([20])).any()   # one block: nothing to cover


@pytest.mark.gpu
@pytest.mark.parametrize("mode,options", [("slices", "cover,split,2"), ("slices", "pull,whole,1"), ("blocks", "cover,split,2"),
                                          ("rmat", "cover,whole,2"), ("grid1x2", "cover,split,2"), ("train", "cover,split,2"),
                                          ("dropout", "-"), ("dropout_directed", "-"), ("train_dropout", "-"), ("gcn", "-"), ("gcnii", "-")])
def test_sharded_native_backend_two_ranks_one_gpu(mode, options):
    """The libgnx.so backend on real shards (rectangular CSR over [regions | local | regions], interior / boundary
    handles with row maps, the send CSR, exchange on its own stream): two ranks share cuda:0 and exchange over gloo
    (staged through the host; RCCL needs one GPU per rank, which the single-GPU box cannot give)."""
    launch(2, mode, "cuda", options)
