"""One configuration of scripts/walk_bench.py (Philox, resident likelihood, rounds until a walker is harvested) at a large population,
for rocprofv3 --kernel-trace --stats.   python scripts/walk_large_profile.py [popsize] [calls]"""
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

popsize = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 25
d, nlive, nsteps = 50, 400, 10
rs = np.random.RandomState(1)
u = 0.55 + 0.01 * rs.normal(size=(nlive, d))
theta = u * 20 - 10
Ls = -2 * (100 * (theta[:, 1:] - theta[:, :-1]**2)**2 + (1 - theta[:, :-1])**2).sum(axis=1)
Lmin = np.sort(Ls)[nlive // 10]
layer = m.AffineLayer()
layer.optimize(u, u)
region = m.MLFriends(u, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
region.create_ellipsoid()
np.random.seed(3)
sampler = pop.PopulationSliceSampler(popsize=popsize, nsteps=nsteps, generate_direction=pop.generate_mixture_random_direction,
                                     scale=0.1, device_rng=DeviceRNG(7))
sampler.max_rounds = 256
for _ in range(5):
    sampler.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
t0 = time.perf_counter()
rounds = []
for _ in range(ncalls):
    sampler.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
    rounds.append(sampler.rounds_last_call)
dt = (time.perf_counter() - t0) / ncalls
print(json.dumps(dict(popsize=popsize, ms_per_call=dt * 1e3, rounds=rounds)))
