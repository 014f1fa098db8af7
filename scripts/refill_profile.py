"""mlf_region_refill at C5 (N = 4000, d = 50, 2^20 draws inside the wrapping ellipsoid, Gaussian likelihood evaluated in place,
only the points above the threshold travel back): wall time per batch; under rocprofv3 --kernel-trace --stats its kernels.
    python scripts/refill_profile.py [reps]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ultranest_amd import likelihoods  # noqa: E402
from ultranest_amd.regions import DeviceRNG  # noqa: E402

import gc  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
u, region = bench.build_region(None)
region.device_rng = DeviceRNG(3)
loglike = likelihoods.GaussLikelihood(0.5, 0.05, bench.NDIM)
Ls = loglike(u)
Lmin = float(np.sort(Ls)[-40])          # only ~1 % of the live-point-like proposals pass: the copy back is small
n = 1 << 20
out = {}
gc.collect()
gc.freeze()      # a full pass of the cyclic collector over torch's objects (35 ms) landed inside the 10 timed calls of the second
gc.disable()     # method in every earlier run of this script (3.7-4.4 ms "per batch" for 0.3 ms of work)
methods = [(1, "wrapping_ellipsoid"), (0, "boundingbox")]
if "--all" in sys.argv:      # the two methods whose proposals are born in the whitened space
    methods += [(2, "transformed_boundingbox"), (3, "points")]
    region.bbox_lo = region.unormed.min(axis=0)     # what MLFriends.sample_from_transformed_boundingbox computes on first use
    region.bbox_hi = region.unormed.max(axis=0)
for method, name in methods:
    for _ in range(3):
        region._dev.refill(region, True, method, n, Lmin, likelihoods.identity_transform.device_spec, loglike.device_spec)
    t0 = time.perf_counter()
    kept = nev = 0
    for _ in range(reps):
        uu, pp, LL, ne = region._dev.refill(region, True, method, n, Lmin, likelihoods.identity_transform.device_spec, loglike.device_spec)
        kept += len(LL)
        nev += ne
    dt = (time.perf_counter() - t0) / reps
    out[name] = dict(ms_per_batch=dt * 1e3, draws=n, evaluated_per_batch=nev / reps, kept_per_batch=kept / reps,
                     draws_per_s=n / dt)
print(json.dumps(out))
