"""Vertex orders for ``GNN(reorder=...)`` -- not in the reference (TensorFlow's kernel walks the COO as stored); a legal
preprocessing step of the graph ingest ("vertex reordering"), counted as prep time.

``locality_order``: a numbering in which neighbours in the graph are neighbours in memory, for graphs that HAVE communities
(citation / co-purchase graphs such as the reference's Cora, Citeseer, Pubmed, ogbn-arxiv: experiments/experiment_setup.py:153-181).
Ten rounds of synchronous label propagation guess the communities; vertices are then numbered community by community.  The
library is told so (gnx_graph_set_row_window): it takes the rows in windows of that numbering and gives the workgroups of one XCD
a contiguous stretch of it, so that each L2 holds the rows of H its own communities gather.  What it gives and where it loses
(R-MAT has no communities: the degree order of the default is better there) is in profiles/NOTES.md, round 5.
Pure torch: runs wherever the index tensors live."""
from __future__ import annotations

import torch

LOCALITY_ROUNDS = 10             # rounds of label propagation (measured on the 10M-vertex community graph, share of entries inside a window /
                                 # K = 10 loop at C = 8: 1 round 0.05 / 17.9 ms, 2 0.15 / 17.3, 4 0.38 / 15.1, 6 0.50 / 14.4, 10 0.56 / 14.0,
                                 # 20 0.58 / 13.9; the generator's own communities 0.57 / 14.0; prep 0.08 ... 0.34 s)
LOCALITY_WINDOW = 4096           # rows per window handed to gnx_graph_set_row_window (measured 4096 ... 65536: profiles/NOTES.md round 5)
LOCALITY_MIN_SHARE = 0.1         # the order is kept when at least this share of the entries lies between vertices less than a window apart
LOCALITY_MIN_LIFT = 3.0          # ... AND that is at least this many times what a numbering WITHOUT community knowledge gives: a random one
                                 # (2 W / n) and the plain degree order (hub-to-hub entries are close there on any power-law graph).
                                 # Community graph of 10M vertices: 0.36 against 0.0008 / 0.0008; R-MAT 10M: 0.002; R-MAT 200K / 6M entries:
                                 # 0.35 against 0.04 / ~0.3 (hubs, not communities).  Graphs of fewer than ~6 windows cannot pass -- they
                                 # fit the caches whatever their order


def propagate_labels(rows: torch.Tensor, cols: torch.Tensor, n: int, rounds: int = LOCALITY_ROUNDS) -> torch.Tensor:
    """Synchronous label propagation over the stored entries (row <- col): every vertex takes the label most of its neighbours
    carry (ties: the larger label), ``rounds`` times from singleton labels.  Vertices without entries keep their own label."""
    label = torch.arange(n, device=rows.device)
    rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    for _ in range(rounds):
        pair, counts = torch.unique(rows * n + label[cols], return_counts=True)         # (vertex, neighbour label) -> neighbours carrying it
        best = torch.zeros(n, dtype=torch.int64, device=rows.device)
        best.scatter_reduce_(0, torch.div(pair, n, rounding_mode="floor"), counts * n + pair % n, reduce="amax")
        label = torch.where(best > 0, best % n, label)
    return label


def coarser_labels(rows: torch.Tensor, cols: torch.Tensor, label: torch.Tensor, rounds: int = LOCALITY_ROUNDS) -> torch.Tensor:
    """One level up: the groups of ``label`` become the vertices of the quotient graph (one entry per stored entry between two
    different groups, multiplicities kept as repeated entries) and label propagation runs on THAT -- groups that exchange many
    entries merge.  Returns, per original vertex, the label of its group's group."""
    groups, group_of = torch.unique(label, return_inverse=True)
    gr, gc = group_of[rows.to(torch.int64)], group_of[cols.to(torch.int64)]
    cross = gr != gc
    upper = propagate_labels(gr[cross], gc[cross], int(groups.numel()), rounds)
    return upper[group_of]


def locality_order(indices: torch.Tensor, n: int, rounds: int = LOCALITY_ROUNDS, levels: int = 1) -> torch.Tensor:
    """new id -> old id.  Vertices with entries first, grouped by their propagated label (groups in ascending label order, inside a
    group heaviest first, then by old id); with ``levels`` > 1 the groups themselves are grouped by label propagation on the
    quotient graph, level by level (a large community that the first level leaves in many small groups is put back together);
    vertices without entries last."""
    rows, cols = indices[:, 0], indices[:, 1]
    degree = torch.bincount(rows, minlength=n)
    label = propagate_labels(rows, cols, n, rounds)
    order = torch.argsort(degree, descending=True, stable=True)                       # by degree ...
    order = order[torch.argsort(label[order], stable=True)]                           # ... inside a label ...
    for _ in range(levels - 1):                                                       # ... the labels inside their coarser groups ...
        label = coarser_labels(rows, cols, label, rounds)
        order = order[torch.argsort(label[order], stable=True)]
    return order[torch.argsort((degree[order] == 0).to(torch.int8), stable=True)]     # ... and the empty rows behind everything


def share_within(indices: torch.Tensor, newid: torch.Tensor, window: int) -> float:
    """Share of the stored entries whose two vertices are less than ``window`` positions apart under the numbering ``newid``."""
    if indices.shape[0] == 0:
        return 0.0
    return float(((newid[indices[:, 0]] - newid[indices[:, 1]]).abs() < window).float().mean())


def found_communities(share: float, n: int, window: int, baseline_share: float = 0.0) -> bool:
    """Is a numbering with ``share`` of the entries inside a window a locality order worth handing to the library?
    ``baseline_share``: the same share under the plain degree order of the same graph."""
    chance = min(1.0, 2.0 * window / max(n, 1))
    return share >= LOCALITY_MIN_SHARE and share >= LOCALITY_MIN_LIFT * max(chance, baseline_share)


def degree_order_share(indices: torch.Tensor, n: int, window: int) -> float:
    """share_within under the stable order of descending entry count: what closeness hubs alone produce."""
    order = torch.argsort(torch.bincount(indices[:, 0], minlength=n), descending=True, stable=True)
    newid = torch.empty_like(order)
    newid[order] = torch.arange(n, device=order.device)
    return share_within(indices, newid, window)
