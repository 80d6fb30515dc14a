#!/usr/bin/env python3
"""Same-process A/B of the K loop with and without the two round-4 rules for rows without entries: never written into a work buffer
when nobody references them (bit 1 << 20 switches that off), and their slots of the degree-binned order not launched at all while
they are skipped (bit 1 << 21 switches that off); bit 1 << 22: the rows' entry ranges read through rowptr[row_order[slot]] (as before)
instead of the slot-ordered copies.  Tuning build:  GNX_LIBRARY=gnn-tf_amd/lib/tune/libgnx.so python3 tools/settled_ab.py [--workload config5]"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch
import bench, gnntf
from gnntf import _native as nat
ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="config5"); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--feats", type=int, default=0); a = ap.parse_args()
n, e, C = bench.WORKLOADS[a.workload]
C = a.feats or C
dev = torch.device("cuda:0"); gnntf.set_default_device(dev)
g, adj, _ = bench.build_single(argparse.Namespace(nodes=n, entries=e), dev)
lib = nat.lib(); lib.gnx_debug_set_tune.argtypes = [ctypes.c_int]
H0 = torch.rand(g.n_rows, C, device=dev) * 2 - 1
out, work = torch.empty_like(H0), torch.empty_like(H0)
run = lambda: nat.check(lib.gnx_appnp_propagate(g.handle, nat.ptr(adj.vals), None, nat.ptr(H0), 0.1, 10, C, nat.ptr(out), nat.ptr(work), nat.current_stream()))
res = {"new_rule": [], "old_rule": [], "empty_slots_launched": [], "slot_ptrs": []}
ref = None
for r in range(a.rounds + 1):
    for name, tune in (("new_rule", 0), ("old_rule", 1 << 20), ("empty_slots_launched", 1 << 21), ("slot_ptrs", 1 << 22)):
        lib.gnx_debug_set_tune(tune)
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e_.record(); torch.cuda.synchronize()
        if r: res[name].append(s.elapsed_time(e_))
        if ref is None: ref = out.clone()
        assert torch.equal(out, ref), name
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps({"workload": a.workload, "shipped_ms": med(res["new_rule"]), "settled_rows_written_to_work_buffers_ms": med(res["old_rule"]),
                  "empty_row_slots_launched_ms": med(res["empty_slots_launched"]), "entry_ranges_through_rowptr_ms": med(res["slot_ptrs"]), "C": C, "same_bits": True, "kernel": g.last_kernel()}))
