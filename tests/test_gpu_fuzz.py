"""The seeded fuzz (tests/fuzz_kernels.py) inside the GPU suite: a fixed 200-case slice, and the one miss the long runs ever
produced as a pinned case of its own.

The miss (round 4, seed 303, case 1081): the chained backward of the training loop (gnx_spmm_dropped_back:
trainable.py:70-78's gradient of K dropped iterations, layered.py:47-50) differed from K un-chained launches by 6.07e-5 relative
to the ROW's own largest element.  The row in question is one of a C = 1 problem whose terms of size ~1 cancel to 1.1e-3: what it
carries is the float32 noise of its terms.  The fuzz now scales against at least 1 % of the matrix's largest element; here the
recorded inputs (tests/golden/fuzz_seed303_case1081.npz, written by `python tests/fuzz_kernels.py --dump 1081 303 ...` at the round-5
commit that split the fuzz into draw_case / check_case, BEFORE the row-window draw was added to the graph cases) are judged against
float64 through the oracle's materialised dropped adjacencies, per row, on the scale rounding acts on (the sum of the absolute
values of the terms): the chained result must be as close to float64 as the step-by-step loop is, or within 8 roundings."""
import os

import numpy as np
import pytest
import torch

import fuzz_kernels as fz

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -24                                  # one float32 rounding (half an ulp, relative)


def test_fixed_slice_of_the_fuzz():
    stats = fz.run(200, seed=11, verbose=False)
    assert sum(stats.values()) >= 190 and all(stats[k] >= 20 for k in fz.KINDS), stats


def test_pinned_fixture_is_the_replayed_case(golden_dir):
    """The fixture holds the case the round-4 run reported: kind 1 (training loops), n = 65, C = 1, K = 4, p = 0.9, 1732 unique entries."""
    s = fz.load(os.path.join(golden_dir, "fuzz_seed303_case1081.npz"))
    assert (s["seed"], s["case"], s["kind"], s["n"], s["C"], s["K"], s["p"]) == (303, 1081, 1, 65, 1, 4, 0.9)
    assert s["idx"].shape == (1732, 2) and len(np.unique(s["idx"], axis=0)) == 1732


def test_seed_303_case_1081_chained_backward_against_float64(golden_dir, capsys):
    s = fz.load(os.path.join(golden_dir, "fuzz_seed303_case1081.npz"))
    r = fz.training_loops(s)
    ref, mag = fz.backward_float64(s, r["a"])
    chained = r["b_got"].double().cpu().numpy()
    steps = r["b_want"].double().cpu().numpy()
    e_chained, e_steps = np.abs(chained - ref), np.abs(steps - ref)
    scale = np.maximum(mag, 1e-30)
    with capsys.disabled():
        worst = int((e_chained / scale).argmax())
        print(f"\n[seed 303 / case 1081] error against float64 in roundings of sum|terms|: chained max {float((e_chained / scale).max() / EPS):.2f} "
              f"(row {worst}: value {ref[worst, 0]:.3e}, sum|terms| {mag[worst, 0]:.3e}), step loop max {float((e_steps / scale).max() / EPS):.2f}; "
              f"chained vs steps relative to the row's own value: {float((np.abs(chained - steps) / np.maximum(np.abs(steps), 1e-30)).max()):.2e}; "
              f"rows where the chained result is the closer one: {int((e_chained <= e_steps).sum())} of {len(ref)}")
    allowed = np.maximum(e_steps, 8 * EPS * scale)
    assert (e_chained <= allowed).all(), f"chained backward off by {float((e_chained / scale).max() / EPS):.1f} roundings of sum|terms|"
    # and the forward loop of the same case, against its own step-by-step form on the fuzz's scale
    assert float(fz.rel_rows(r["f_got"], r["f_want"]).max()) < 2e-5 * (1.0 + np.sqrt(np.bincount(s["idx"][:, 0]).max()) / 10.0)
    # the old criterion (relative to the row's own largest element, floor 1e-3) is what the case missed: it must still see the miss,
    # i.e. this fixture keeps exercising the cancelling row
    own = (r["b_got"] - r["b_want"]).abs() / r["b_want"].abs().max(dim=1, keepdim=True).values.clamp_min(1e-3)
    # (6.07e-5 when recorded: float32 noise of terms ~1 on a row of 1.1e-3) -- bounded from BOTH sides: a fixture that stopped
    # hitting the cancelling row would show ~1e-7 here and no longer test what it was pinned for
    assert 1e-5 < float(own.max()) < 1e-3
