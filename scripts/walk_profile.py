"""Per-phase wall time of one PopulationSliceSampler.__next__ (resident likelihood)."""
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, ".")
import ultranest_amd.mlfriends as m
import ultranest_amd.popstepsampler as pop
from ultranest_amd import likelihoods
from ultranest_amd.regions import DeviceRNG

d, nlive = 50, 400
rs = np.random.RandomState(1)
u = 0.55 + 0.01 * rs.normal(size=(nlive, d))
Ls = likelihoods.rosenbrock_loglike(likelihoods.rosenbrock_transform(u))
Lmin = np.sort(Ls)[nlive // 10]
layer = m.AffineLayer()
layer.optimize(u, u)
region = m.MLFriends(u, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
region.create_ellipsoid()

acc = defaultdict(float)


def timed(cls, name):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t0
        return r
    setattr(cls, name, wrap)


for name in ("begin", "start", "points", "brackets", "brackets_philox", "set_direction_data", "set_layer", "propose",
             "finish", "finish_dev"):
    timed(pop._Walkers, name)
timed(pop.PopulationSliceSampler, "_sync_region")

for popsize in (100, 10000):
    for seed in (None, 7):
        acc.clear()
        np.random.seed(3)
        s = pop.PopulationSliceSampler(popsize=popsize, nsteps=10, generate_direction=pop.generate_mixture_random_direction,
                                       scale=0.1, device_rng=None if seed is None else DeviceRNG(seed))
        for _ in range(5):
            s.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
        acc.clear()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            s.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
        tot = (time.perf_counter() - t0) / n
        print("popsize %d rng %s: %.1f us/call;" % (popsize, "philox" if seed else "numpy", tot * 1e6),
              ", ".join("%s %.1f" % (k, v / n * 1e6) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
