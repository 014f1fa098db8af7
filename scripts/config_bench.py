#!/usr/bin/env python3
"""Numbers for the BASELINE.json configurations other than the headline (GPU box): proposal sets
E (inside the wrapping ellipsoid), U (uniform cube: ellipsoid test only), F (no neighbour within
reach, radius below the pre-filter's range: exact full scan), N (radius / sqrt(5): almost no neighbour, pre-filter) through MLFriends.inside, region rebuild, bootstrap sharding unit, likelihood
kernels.  One JSON object on stdout; scripts/collect_profiles.py stores it under profiles/."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ultranest_amd.mlfriends as M  # noqa: E402
from ultranest_amd import _lib, kernels as K  # noqa: E402
from ultranest_amd.harness import RegionUpdater  # noqa: E402

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
out = {"device": _lib.device_name()}


def region_for(n, d):
    rs = np.random.RandomState(1)
    u = 0.5 + 0.05 * rs.normal(size=(n, d))
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    # the first call of a process pays context creation, code-object loading of the kernels of this dimensionality and
    # the first allocations (round 2 reported 9.0 ms for C2 -- the first configuration of the loop -- against 1.7 ms at
    # C5 for exactly that reason): one untimed call on another stream position, then the median of three
    region.compute_enlargement(nbootstraps=30, rng=np.random.RandomState(99))
    ts = []
    for rep in range(3):
        rs_t = np.random.RandomState(1000 + rep)
        t0 = time.perf_counter()
        region.compute_enlargement(nbootstraps=30, rng=rs_t)
        ts.append((time.perf_counter() - t0) * 1e3)
    region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30, rng=rs)
    boot_ms = float(np.median(ts))
    region.create_ellipsoid()
    return u, region, boot_ms


def timed_inside(handle, pts, reps=5):
    p = pts.shape[0]
    mask = torch.empty(p, dtype=torch.uint8, device=dev)
    handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return p / dt, float(mask.float().mean().item())


def prep_kernel(d):
    return ("k_prep4's arithmetic (split-binary16 matrix cores, bounded): inside the first sweep launch (k_prep_sweep) for phased batches of even d <= 56, "
            "in its own launch (k_prep4) otherwise" if d <= 64 else
            "k_prep_mfma64 (bounded quadratic form + whitening chain on the FP64 matrix cores, matrices streamed from L2)" if d <= 128 else
            "k_prep_wide (run-time dimensionality, mlf_wide.hip)")


# the reference's only PUBLISHED workload is a 100-d Gaussian with 400 live points (docs/performance.rst:221-223);
# the d = 100 rows take the routing of 65 ... 128 dimensions (measured nowhere before); d = 256: mlf_wide.hip
CONFIGS = [("C2", 2000, 20, 100000), ("C5", 4000, 50, 1000000), ("C1-size", 400, 5, 100000),
           ("published-shape", 400, 100, 100000), ("C5-at-d100", 4000, 100, 1000000), ("wide", 1000, 256, 100000)]
if len(sys.argv) > 1:
    CONFIGS = [c for c in CONFIGS if c[0] in sys.argv[1:]]
for name, n, d, p in CONFIGS:
    u, region, boot_ms = region_for(n, d)
    handle = region._dev.sync(region, True)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    z = torch.randn(p, d, dtype=torch.float64, device=dev, generator=g)
    z /= z.norm(dim=1, keepdim=True)
    z *= region.enlarge ** 0.5 * torch.rand(p, 1, dtype=torch.float64, device=dev, generator=g) ** (1.0 / d)
    E = (torch.as_tensor(region.ellipsoid_center, device=dev) + z @ torch.as_tensor(region.ellipsoid_axes_T.copy(), device=dev)).contiguous()
    U = torch.rand(p, d, dtype=torch.float64, device=dev, generator=g)
    res = {"bootstrap30_ms": boot_ms}
    # N: (almost) no proposal has a neighbour, but the radius is in the pre-filter's range: every query sweeps all live
    # points through the matrix cores and leaves undecided -- the realistic "full scan" (F falls back to the exact kernel)
    for label, pts, r2 in (("E", E, region.maxradiussq), ("U", U, region.maxradiussq), ("F", E, 1e-300),
                           ("N", E, region.maxradiussq * 0.2)):
        handle.set_thresholds(region.enlarge, r2)
        rate, acc = timed_inside(handle, pts)
        on, kdim, _ = handle.filter_info(p)
        res["per_proposal_stage"] = prep_kernel(d)
        res[label] = {"proposals_per_s": rate, "accept": acc,
                      # what the reference's loop would do on this batch at most: 3 d flop per (proposal, live point) pair
                      "equivalent_allpairs_TFLOPs": 3.0 * d * n * rate / 1e12, "prefilter_k_columns": kdim if on else None,
                      "scan_kernel": "MFMA pre-filter + exact re-check" if on else
                                     "k_scan: the exact FP64 scan alone (the radius is outside the pre-filter's range -- set F's r2 = 1e-300 "
                                     "is SURVEY 8d's no-early-exit worst case, not a radius a run produces)"}
    upd = RegionUpdater(d, region_class=M.MLFriends, transform_layer_class=M.LocalAffineLayer, freeze_gc=True)
    np.random.seed(11)
    try:
        upd.update(u, nbootstraps=30, minvol=0.)
    except Exception as e:      # e.g. fewer selected points than dimensions in a bootstrap round (n = 400, d = 100 is fine; recorded otherwise)
        res["rebuild_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
        out[name + " N=%d d=%d P=%d" % (n, d, p)] = res
        continue
    rs = np.random.RandomState(7)
    ts = []
    for rep in range(3):
        u2 = u.copy()
        u2[:n // 10] = 0.5 + 0.045 * rs.normal(size=(n // 10, d))
        t0 = time.perf_counter()
        upd.update(u2, nbootstraps=30, minvol=0.)
        ts.append((time.perf_counter() - t0) * 1e3)
    res["rebuild_ms"] = float(np.median(ts))
    out[name + " N=%d d=%d P=%d" % (n, d, p)] = res

# likelihood kernels on device-resident batches (hipEvents through torch on the launch stream)
lib = _lib.lib()
import ctypes  # noqa: E402
for kind, name, d in ((0, "gauss", 50), (1, "eggbox", 10), (1, "eggbox", 50), (3, "rosenbrock", 50)):
    n = 1000000
    x = torch.rand(n, d, dtype=torch.float64, device=dev)
    aux = torch.full((d,), 0.5, dtype=torch.float64, device=dev)
    like = torch.empty(n, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(2):
        e0.record()
        _lib.check(lib.mlf_loglike_dev(kind, ctypes.c_void_p(x.data_ptr()), d, n, ctypes.c_void_p(aux.data_ptr()), 0.1,
                                       ctypes.c_void_p(like.data_ptr()), ctypes.c_void_p(stream)))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out["loglike %s d=%d n=1e6" % (name, d)] = {"ms": ms, "points_per_s": n / (ms * 1e-3), "GBps": n * (8 * d + 8) / (ms * 1e-3) / 1e9}
print(json.dumps(out, indent=1))
