#!/usr/bin/env python3
"""Can a partition-aware relabelling cut the halo of the P-GPU run?  (SURVEY 8(e) "hub-aware partitioning".)

bench.py --gpus P cuts the config-5 graph into P contiguous blocks of its (random) vertex labelling.  This tool computes, for the
SAME graph and several other assignments of vertices to blocks, what the exchange would move per iteration: for every rank the
rows of its pull / push vertex cover (sharded.cover_push_mask -- the plan the bench builds) and of the classic pull-only halo,
the share of entries whose column is remote, and the balance of entries over the blocks (the kernels' balance).

  random        contiguous blocks of the random labelling (today)
  lp            balanced label propagation from the random start: every round a vertex moves to the block holding most of its
                neighbours, net of a per-block price that keeps the entries per block within a few percent (--rounds)
  degree        contiguous blocks of the degree order with EQUAL ENTRIES per block (hub blocks are small)
  hubs_spread   the degree order dealt round-robin over the blocks (hubs spread evenly -- what random does only on average)

    python3 tools/partition_experiment.py --world 8 > gpurun_out/partition_p8.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch

from gnntf import sharded


def lp_labels(u, w, deg, n, P, rounds, log):
    """Balanced label propagation: a vertex wants the block holding most of its neighbours; between every two blocks only as much
    entry mass moves one way as moves the other (the best gains first), so the entries per block stay what they were."""
    dev = u.device
    label = torch.div(torch.arange(n, device=dev) * P, n, rounding_mode="floor")
    degf = deg.double()
    for r in range(rounds):
        key = torch.cat([u * P + label[w], w * P + label[u]])
        cnt = torch.bincount(key, minlength=n * P).view(n, P)
        del key
        here = cnt.gather(1, label[:, None]).reshape(-1)
        best, want = cnt.max(dim=1)
        del cnt
        movers = torch.nonzero((best > here) & (want != label)).reshape(-1)
        gain = (best - here)[movers].double() / deg[movers].double()
        pair = label[movers] * P + want[movers]
        o1 = torch.argsort(gain, descending=True, stable=True)
        o = o1[torch.argsort(pair[o1], stable=True)]                     # by pair, best gain first
        movers, pair, mass = movers[o], pair[o], degf[movers[o]]
        total = torch.bincount(pair, weights=mass, minlength=P * P)
        limit = torch.minimum(total.view(P, P), total.view(P, P).t()).reshape(-1)
        before = torch.cumsum(total, 0) - total                          # mass of the pairs sorted ahead
        within = torch.cumsum(mass, 0) - before[pair]
        ok = within <= limit[pair]
        label = label.clone()
        label[movers[ok]] = want[movers[ok]]
        m = torch.bincount(label, weights=degf, minlength=P)
        cut = float((label[u] != label[w]).float().mean())
        log.append(dict(round=r, wanted_to_move=int(movers.numel()), moved=int(ok.sum()), cut_share=cut,
                        entries_max_over_mean=float(m.max() / m.mean())))
        del here, best, want, movers, gain, pair, o1, o, mass, total, limit, before, within, ok
    return label


def evaluate(name, label, u, w, deg, n, P, extra=None):
    """Per-rank plan sizes for the blocks ``label`` defines (vertices renumbered block by block, stable)."""
    dev = u.device
    t0 = time.time()
    order = torch.argsort(label, stable=True)
    newid = torch.empty_like(order)
    newid[order] = torch.arange(n, device=dev)
    counts = torch.bincount(label, minlength=P)
    bounds = [0] + torch.cumsum(counts, 0).tolist()
    bnd = torch.tensor(bounds[1:], dtype=torch.int64, device=dev)
    uu, ww = newid[u], newid[w]
    lu, lw = label[u], label[w]
    mass = torch.bincount(label, weights=deg.double(), minlength=P)
    ranks, pair_rows = [], []
    for r in range(P):
        mu, mw = lu == r, lw == r
        row = torch.cat([uu[mu], ww[mw]]) - bounds[r]
        col = torch.cat([ww[mu], uu[mw]])
        owner = torch.cat([lw[mu], lu[mw]])
        n_local = bounds[r + 1] - bounds[r]
        remote = owner != r
        push = sharded.cover_push_mask(row, col, owner, r, n_local, bnd)
        pulled = int(torch.unique(col[remote & ~push]).numel())
        pushed = int(torch.unique(owner[push] * n_local + row[push]).numel())
        pull_only = int(torch.unique(col[remote]).numel())
        # rows arriving from every peer q (one xGMI link per pair): the columns pulled from q + the partial sums q pushes
        from_peer = torch.bincount(torch.bucketize(torch.unique(col[remote & ~push]), bnd, right=True), minlength=P) \
            + torch.bincount(torch.div(torch.unique(owner[push] * max(n_local, 1) + row[push]), max(n_local, 1), rounding_mode="floor"), minlength=P)
        pair_rows.append([int(x) for x in from_peer.tolist()])
        ranks.append(dict(rank=r, rows=n_local, entries=int(row.numel()), remote_entry_share=float(remote.float().mean()) if row.numel() else 0.0,
                          cover_rows=pulled + pushed, pulled=pulled, pushed=pushed, pull_only_rows=pull_only,
                          entries_pushed=int(push.sum())))
        del mu, mw, row, col, owner, remote, push
    ent = [x["entries"] for x in ranks]
    cov = [x["cover_rows"] for x in ranks]
    out = dict(partition=name, world=P, cut_share=float((lu != lw).float().mean()),
               entries_max_over_mean=max(ent) / (sum(ent) / P), cover_rows_max=max(cov), cover_rows_mean=sum(cov) / P,
               pull_only_rows_max=max(x["pull_only_rows"] for x in ranks),
               # the links are point to point: the exchange lasts as long as the busiest (receiver, sender) pair
               busiest_link_rows=max(max(row) for row in pair_rows), rows_from_peer=pair_rows,
               # what the slowest rank pays per iteration: its kernels scale with its entries, its exchange with its cover rows
               ranks=ranks, seconds=round(time.time() - t0, 1))
    if extra:
        out.update(extra)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=80_000_000)
    ap.add_argument("--entries", type=int, default=1_000_000_000)
    ap.add_argument("--feats", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--device", type=str, default="cuda:0")
    a = ap.parse_args()
    dev = torch.device(a.device)
    n, P = a.nodes, a.world
    u, w = sharded.rmat_relabelled_pairs(n, a.entries // 2, seed=1, device=dev)
    deg = torch.bincount(u, minlength=n) + torch.bincount(w, minlength=n)
    names = a.only.split(",") if a.only else ["random", "lp", "degree", "hubs_spread"]
    results = []
    for name in names:
        extra = None
        if name == "random":
            label = torch.div(torch.arange(n, device=dev) * P, n, rounding_mode="floor")
        elif name == "lp":
            log = []
            label = lp_labels(u, w, deg, n, P, a.rounds, log)
            extra = {"lp_rounds": log}
        elif name == "degree":
            order = torch.argsort(deg, descending=True, stable=True)
            cum = torch.cumsum(deg[order].double(), 0)
            blk = torch.clamp((cum / (float(cum[-1]) / P)).floor().long(), max=P - 1)
            label = torch.empty(n, dtype=torch.int64, device=dev)
            label[order] = blk
            del order, cum, blk
        elif name == "hubs_spread":
            order = torch.argsort(deg, descending=True, stable=True)
            label = torch.empty(n, dtype=torch.int64, device=dev)
            label[order] = torch.arange(n, device=dev) % P
            del order
        else:
            raise SystemExit("unknown partition " + name)
        res = evaluate(name, label, u, w, deg, n, P, extra)
        res["halo_bytes_per_iteration_max"] = res["cover_rows_max"] * a.feats * 4
        res["busiest_link_bytes_per_iteration"] = res["busiest_link_rows"] * a.feats * 4
        results.append(res)
        sys.stderr.write(json.dumps({k: v for k, v in res.items() if k not in ("ranks", "rows_from_peer")}) + "\n")
        sys.stderr.flush()
        del label
        torch.cuda.empty_cache()
    print(json.dumps(dict(graph=dict(nodes=n, entries=a.entries, features=a.feats), results=results), indent=1))


if __name__ == "__main__":
    main()
