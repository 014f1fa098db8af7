"""Where a multi-round sampler call spends its time (C3 shape: eggbox d = 10, 1000 live points, popsize 1024, nsteps 40):
wall time per __next__, rounds per call, the C call alone, the live-point update alone.  Under rocprofv3 --stats: the kernels.
    python scripts/walk_rounds_profile.py [calls]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ultranest_amd.mlfriends as m  # noqa: E402
import ultranest_amd.popstepsampler as pop  # noqa: E402
from ultranest_amd import likelihoods  # noqa: E402
from ultranest_amd.regions import DeviceRNG  # noqa: E402

ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
d, nlive = 10, 1000
rs = np.random.RandomState(4)
us = rs.uniform(size=(nlive, d))
Ls = likelihoods.eggbox_loglike(likelihoods.eggbox_transform(us))
layer = m.AffineLayer()
layer.optimize(us, us)
region = m.MLFriends(us, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
region.create_ellipsoid()
out = {}
for mr in (256, 1):
    sampler = pop.PopulationSliceSampler(popsize=1024, nsteps=40, generate_direction=pop.generate_mixture_random_direction, scale=1.0,
                                         device_rng=DeviceRNG(7))
    sampler.max_rounds = mr
    u2, L2 = us.copy(), Ls.copy()
    rounds, t_calls = [], []
    found = 0
    n = ncalls if mr > 1 else ncalls * 4
    t_begin = time.perf_counter()
    for it in range(n):
        worst = int(np.argmin(L2))
        t0 = time.perf_counter()
        unew, pnew, Lnew, nc = sampler.__next__(region, L2[worst], u2, L2, likelihoods.eggbox_transform, likelihoods.eggbox_loglike)
        t_calls.append(time.perf_counter() - t0)
        rounds.append(getattr(sampler, "rounds_last_call", 1))
        if unew is not None:
            found += 1
            u2[worst], L2[worst] = unew, Lnew
    total = time.perf_counter() - t_begin
    t = np.array(t_calls[n // 5:]) * 1e6
    out["max_rounds_%d" % mr] = dict(calls=n, found=found, us_per_call_median=float(np.median(t)), us_per_call_mean=float(t.mean()),
                                     rounds_mean=float(np.mean(rounds[n // 5:])), rounds_max=int(np.max(rounds)),
                                     us_per_round=float(t.mean() / np.mean(rounds[n // 5:])), total_s=total)
# the two C calls alone (steady state of the multi-round sampler)
w = sampler._walkers
t0 = time.perf_counter()
for _ in range(300):
    w.update_live(np.array([3]), u2[[3]], L2[[3]])
out["update_live_us"] = (time.perf_counter() - t0) / 300 * 1e6
print(json.dumps(out))
