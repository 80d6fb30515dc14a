"""bench_sharded.select_variant under a fake clock: the selection budget holds INSIDE a variant -- the bare
exchange and kernels are timed first, a full step starts only if K x (exchange + kernels) still fits -- so the selection ends within
its budget plus one step; plans are built in the order cover, pull, weighted; a losing plan is freed before the next is built."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import bench_sharded


class FakeOps:
    """A world in which an exchange takes ``exchange_s`` per iteration, the kernels ``compute_s``, a step K x max of the two plus a
    per-cover penalty, a plan ``plan_s`` to build.  Every operation advances the clock."""

    def __init__(self, K=10, exchange_s=3.0, compute_s=0.05, plan_s=20.0, first_plan="cover", slow=None, broken=()):
        self.now, self.K, self.exchange_s, self.compute_s, self.plan_cost = 0.0, K, exchange_s, compute_s, plan_s
        self.plans, self.built, self.freed, self.steps, self.max_resident = {first_plan}, [first_plan], [], [], 1
        self.slow, self.broken = slow or {}, set(broken)

    def spent(self):
        return self.now

    def has_plan(self, cover):
        return cover in self.plans

    def plan_seconds(self):
        return self.plan_cost

    def build_plan(self, cover):
        self.now += self.plan_cost
        if cover in self.broken:
            return False
        self.plans.add(cover)
        self.built.append(cover)
        self.max_resident = max(self.max_resident, len(self.plans))
        return True

    def keep_only(self, cover):
        for c in list(self.plans):
            if cover is not None and c != cover:
                self.plans.discard(c)
                self.freed.append(c)

    def make_state(self, cover, chunks):
        self.now += 0.5
        return (cover, chunks), ""

    def time_alone(self, cover, state):
        self.now += 2 * (self.exchange_s + self.compute_s)
        return self.exchange_s * 1e3, self.compute_s * 1e3

    def early_options(self, cover):
        return [False, True]

    def run_step(self, cover, state, early):
        t = self.K * max(self.exchange_s, self.compute_s) * self.slow.get(cover, 1.0) * (0.97 if early else 1.0)
        self.now += t
        self.steps.append((cover, state[1], early, t))
        return t * 1e3

    def release(self, state):
        pass


def test_selection_ends_within_its_budget_plus_one_step():
    """Round 5's full-size gloo rehearsal: 176.5 s of selection against a 120 s budget.  Here: exchange 3 s per iteration (a step ~30 s)."""
    ops = FakeOps()
    best, variants, skipped = bench_sharded.select_variant(["cover", "pull", "cover@0.5"], [2, 4, 1], ops, budget=120.0, K=10)
    one_step = 10 * 3.0
    assert ops.now <= 120.0 + one_step
    assert best is not None and best["step_ms"] == min(v["step_ms"] for v in variants if v["step_ms"])
    assert skipped and all("reason" in s for s in skipped)
    assert any("predicted" in s["reason"] or "budget" in s["reason"] for s in skipped)
    # never more steps than the budget allows: what was timed fits, what was skipped says why
    assert sum(t for *_, t in ops.steps) <= 120.0 + one_step


def test_a_step_longer_than_the_budget_is_timed_once_and_only_once():
    ops = FakeOps(exchange_s=20.0)                                                   # one step = 200 s > budget
    best, variants, skipped = bench_sharded.select_variant(["cover", "pull"], [2, 4, 1], ops, budget=120.0, K=10)
    assert len(ops.steps) == 1 and best["cover"] == "cover" and best["chunks"] == 2  # the unconditional first variant, one call
    assert ops.built == ["cover"]                                                    # no second plan
    assert ops.now <= 120.0 + 200.0 + 50.0
    assert {s.get("cover") for s in skipped} == {"cover", "pull"}


def test_plans_are_built_in_order_and_losers_freed_first():
    ops = FakeOps(exchange_s=0.02, compute_s=0.03, plan_s=5.0, slow={"pull": 1.5, "cover@0.5": 0.8})
    args = bench.parse(["--gpus", "8"])
    covers = bench_sharded.cover_order(args, pv=8)
    assert covers == ["cover", "pull", "cover@0.5"]                                  # pull before the weighted covers
    best, variants, skipped = bench_sharded.select_variant(covers, [2, 4, 1], ops, budget=120.0, K=10)
    assert ops.built == covers and not skipped
    assert ops.max_resident <= 2                                                     # the best so far + the one being timed
    assert best["cover"] == "cover@0.5" and ops.plans == {"cover@0.5"}
    assert len(variants) == 3 * 3 * 2
    assert bench_sharded.cover_order(args, pv=1) == ["cover"]
    assert bench_sharded.cover_order(bench.parse(["--gpus", "8", "--cover", "pull"]), pv=8) == ["pull"]


def test_a_plan_that_cannot_be_built_is_recorded_and_the_rest_goes_on():
    ops = FakeOps(exchange_s=0.02, compute_s=0.03, plan_s=5.0, broken={"pull"})
    best, variants, skipped = bench_sharded.select_variant(["cover", "pull", "cover@0.5"], [2], ops, budget=120.0, K=10)
    assert [v["cover"] for v in variants if v["step_ms"] is None] == ["pull"]
    assert "cover@0.5" in ops.built and best is not None


def test_halo_block_of_the_line_is_compact():
    table = [dict(cover="cover", chunks=2, early_pull=True, step_ms=612.3456789, exchange_ms_alone=31.0, compute_ms_alone=12.0)] * 30
    halo = dict(max_halo_rows=10, chosen=table[0], variants_timed_before_the_run=table, variants_skipped=[{}] * 3, overlap_probe={"a": 1}, plan="cover")
    line = bench_sharded.line_halo(halo)
    assert len(line["halo_variants"]) == 18 and line["halo_variants"][0] == ["cover", 2, True, 612.3]
    assert line["n_variants_skipped"] == 3 and "overlap_probe" not in line and "variants_timed_before_the_run" not in line
    assert bench_sharded.line_halo(None) is None
