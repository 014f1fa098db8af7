#!/usr/bin/env python3
"""CPU design study for VERDICT r4 item 2: does the ORDER of the live points in the mask-mode operand shorten the
first range of the two-range sweep?  (MLFriends.inside only needs "any hit", mlfriends.pyx:1186-1211; the first-index
contract binds find_nearby alone, :176-183.)

Builds the bench's C5 region (N = 4000, d = 50, 30 bootstrap rounds) through the CPU oracle stand-in (test
infrastructure, as in tests/), draws a sample of set E / set N proposals, computes all pair distances with a matrix
product and reports, per ordering of the live points, the share of proposals decided after the first c tiles of 32 and
the modelled sweep cost  c * P + (T - c) * undecided(c)  (tile visits per proposal) at the best c.  No GPU.

    python scripts/order_study.py [--sample 20000] > profiles/r05_order_study.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def build_region(n, d, seed=1, nboot=30):
    import oracle_backend
    import ultranest_amd.kernels as K
    import ultranest_amd.mlfriends as M
    for name in oracle_backend.PATCHED:
        setattr(K, name, getattr(oracle_backend, name))
        if hasattr(M, name):
            setattr(M, name, getattr(oracle_backend, name))
    rs = np.random.RandomState(seed)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=nboot, rng=rs)
    region.create_ellipsoid(minvol=0.)
    return u, layer, region


def set_e(region, p, seed):
    rs = np.random.RandomState(seed)
    d = len(region.ellipsoid_center)
    z = rs.normal(size=(p, d))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    z *= region.enlarge ** 0.5 * rs.uniform(size=(p, 1)) ** (1.0 / d)
    return region.ellipsoid_center + z @ region.ellipsoid_axes_T


def pair_d2(a, b):
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T


def greedy_cover(hits_live):
    """greedy set cover with the live points themselves as the proxy for the proposals: next = the live point whose ball
    holds the most live points no chosen ball holds yet (ties: lowest index)"""
    n = len(hits_live)
    covered = np.zeros(n, dtype=bool)
    chosen = np.zeros(n, dtype=bool)
    order = []
    h = hits_live.astype(np.int32)
    gain = h.sum(1)
    for _ in range(n):
        g = np.where(chosen, -1, gain)
        i = int(np.argmax(g))
        if g[i] <= 0:
            break
        order.append(i)
        chosen[i] = True
        newly = hits_live[i] & ~covered
        covered |= newly
        gain -= h[:, newly].sum(1)
    rest = [i for i in range(n) if not chosen[i]]
    return np.array(order + rest)


def evaluate(hits, order, ntiles):
    """hits: (N, P) bool in storage order.  Returns decided share after c tiles (c = 0 .. ntiles) and the tile-visit cost model"""
    h = hits[order]
    n, p = h.shape
    pad = ntiles * 32 - n
    if pad:
        h = np.vstack([h, np.zeros((pad, p), dtype=bool)])
    tile_hit = h.reshape(ntiles, 32, p).any(1)            # (T, P)
    first = np.where(tile_hit.any(0), tile_hit.argmax(0), ntiles)     # first tile with a hit, T = none
    decided = np.array([(first < c).mean() for c in range(ntiles + 1)])
    cost = np.array([c + (ntiles - c) * (1.0 - decided[c]) for c in range(ntiles + 1)])
    return decided, cost


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample", type=int, default=20000)
    ap.add_argument("--n", type=int, default=4000)
    ap.add_argument("--d", type=int, default=50)
    args = ap.parse_args()
    u, layer, region = build_region(args.n, args.d)
    t = region.unormed
    r2 = region.maxradiussq
    ntiles = (args.n + 31) // 32
    ctr = t.mean(0)
    dist_c = ((t - ctr) ** 2).sum(1)
    hl = pair_d2(t, t) <= r2
    np.fill_diagonal(hl, True)
    ncount = hl.sum(1)
    orders = {
        "storage": np.arange(args.n),
        "far_first": np.argsort(-dist_c, kind="stable"),
        "near_first": np.argsort(dist_c, kind="stable"),
        "many_neighbours_first": np.argsort(-ncount, kind="stable"),
        "few_neighbours_first": np.argsort(ncount, kind="stable"),
        "greedy_cover_on_live_points": greedy_cover(hl),
    }
    out = {"what": "share of proposals with a hit inside the first c tiles of 32 live points, per ordering of the operand; cost = tile "
                   "visits per proposal of a two-range sweep cut at c (c + (T - c) * undecided(c)); exact binary64 distances by matrix product",
           "n": args.n, "d": args.d, "r2": r2, "enlarge": region.enlarge, "ntiles": ntiles,
           "live_neighbour_count": {"min": int(ncount.min()), "median": float(np.median(ncount)), "max": int(ncount.max())},
           "sets": {}}
    for name, scale in (("E", 1.0), ("N_radius_over_sqrt5", 0.2)):
        pts = set_e(region, args.sample, 77)
        tq = layer.transform(pts)
        hits = pair_d2(t, tq) <= r2 * scale
        k = hits.sum(0)
        rec = {"accepted": float((k > 0).mean()), "neighbours_per_accepted_proposal": {"median": float(np.median(k[k > 0])) if (k > 0).any() else 0.0,
               "p10": float(np.percentile(k[k > 0], 10)) if (k > 0).any() else 0.0}, "orders": {}}
        for oname, order in orders.items():
            decided, cost = evaluate(hits, order, ntiles)
            best = int(np.argmin(cost))
            rec["orders"][oname] = {"decided_after_pct_of_tiles": {str(pc): float(decided[int(round(ntiles * pc / 100))]) for pc in (10, 20, 30, 40, 50, 60)},
                                    "best_cut_tiles": best, "best_cut_pct": 100.0 * best / ntiles, "cost_at_best_cut": float(cost[best]),
                                    "cost_at_50pct": float(cost[ntiles // 2])}
        # the oracle bound: greedy cover on the proposals themselves (not available to the product: it has no proposals at region_set)
        rec["full_sweep_cost"] = float(ntiles)
        out["sets"][name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
