"""bench_device.py -- the device-side helpers every part of bench.py shares: the synthetic graph of SURVEY.md section 8(d), event
timing on the launch stream, the in-run stream yardsticks."""
import time


def build_single(args, device):
    """The whole graph on one GPU: R-MAT pairs -> symmetrised unsorted COO -> device CSR (A0) -> normalise once (A2)."""
    import torch
    import gnntf
    from gnntf import sharded
    n, m = args.nodes, args.entries // 2
    t0 = time.time()
    u, v = sharded.rmat_relabelled_pairs(n, m, seed=1, device=device)
    idx = torch.cat([torch.stack([u, v], 1), torch.stack([v, u], 1)])      # symmetrised COO, unsorted
    del u, v
    vals = torch.ones(idx.shape[0], dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    g = gnntf.DeviceGraph(gnntf.SparseCOO(idx, vals, (n, n)), device=device)      # A0: COO -> CSR on the device
    del idx, vals
    adj = gnntf.normalize(g, "symmetric")                                          # A2, once (eval mode)
    torch.cuda.synchronize()
    t_prep = time.time() - t0
    torch.cuda.empty_cache()
    return g, adj, dict(gen_s=round(t_gen, 2), prep_s=round(t_prep, 2))


def timed_steps(step, steps, warmup, barrier):
    """W untimed steps, then exactly K timed ones bracketed by barrier + synchronize; also per-step events on the
    launch stream.  Returns (wall seconds, [ms per step])."""
    import torch
    for _ in range(warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s, e in ev:                     # events sit on the stream the kernels are launched on
        s.record()
        step()
        e.record()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, [s.elapsed_time(e) for s, e in ev]


def median_ms(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def _stream_rate(device, launch, nbytes_moved, reps):
    import torch
    launch()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        launch()
    e.record()
    torch.cuda.synchronize()
    return 1.0 * nbytes_moved * reps / (s.elapsed_time(e) * 1e-3) / 1e9


def stream_read_GBs(device, nbytes=8 << 30, reps=5):
    """Device read-only streaming rate measured in this run (the SpMM is almost all reads)."""
    import torch
    from gnntf import _native as nat
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    sink = torch.zeros(64, dtype=torch.float32, device=device)
    return _stream_rate(device, lambda: nat.check(nat.lib().gnx_stream_read(nat.ptr(src), src.numel(), nat.ptr(sink), nat.current_stream())),
                        nbytes, reps)


def stream_copy_GBs(device, nbytes=4 << 30, reps=5):
    """Device stream-copy rate (read + write bytes per second) measured in this run: the achievable HBM peak."""
    import torch
    from gnntf import _native as nat
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    return _stream_rate(device, lambda: nat.check(nat.lib().gnx_stream_copy(nat.ptr(src), nat.ptr(dst), src.numel(), nat.current_stream())),
                        2 * nbytes, reps)


def kept_entries(g, p, seed, first_stream, n_streams):
    """Stored entries that survive the edge dropout of each of ``n_streams`` consecutive dropout streams (counted from the
    materialised values, gnx_graph_normalize: a dropped entry is an explicit zero there)."""
    import gnntf
    return [int((gnntf.normalize(g, "symmetric", "none", dropout=p, seed=seed, stream_id=first_stream + k).vals != 0).sum())
            for k in range(n_streams)]
