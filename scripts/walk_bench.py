"""Population slice sampler throughput (row f1): walker-steps per second of one __next__ call,
C5 wiring (Rosenbrock d=50, transform u*20-10, mixture directions; reference
examples/test_PopSliceSampler.py:103-114).

    python scripts/walk_bench.py device      # this package on the GPU (resident state)
    python scripts/walk_bench.py reference   # the real reference (dev container only)
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
impl = sys.argv[1] if len(sys.argv) > 1 else "device"
d, nlive, nsteps = 50, 400, 10


def host_transform(x):
    return x * 20 - 10


def host_loglike(theta):
    a, b = theta[:, :-1], theta[:, 1:]
    return -2 * (100 * (b - a**2)**2 + (1 - a)**2).sum(axis=1)


rs = np.random.RandomState(1)
u = 0.55 + 0.01 * rs.normal(size=(nlive, d))
Ls = host_loglike(host_transform(u))
Lmin = np.sort(Ls)[nlive // 10]

if impl == "reference":
    sys.path.insert(0, os.environ.get("ULTRANEST_REFBUILD") or os.path.join(tempfile.gettempdir(), "ultranest_refbuild"))
    import ultranest.mlfriends as m
    import ultranest.popstepsampler as pop
    configs = [("numpy stream, numpy likelihood", host_transform, host_loglike, None)]
else:
    import ultranest_amd.mlfriends as m
    import ultranest_amd.popstepsampler as pop
    from ultranest_amd import likelihoods
    from ultranest_amd.regions import DeviceRNG
    configs = [("numpy stream, numpy likelihood", host_transform, host_loglike, None),
               ("numpy stream, resident likelihood", likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike, None),
               ("philox, resident likelihood, one round per call", likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike, 7),
               ("philox, resident likelihood, rounds until a walker is harvested", likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike, 7)]

layer = m.AffineLayer()
layer.optimize(u, u)
region = m.MLFriends(u, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
region.create_ellipsoid()

out = []
for popsize in (100, 1000, 10000, 100000):
    for label, transform, loglike, seed in configs:
        np.random.seed(3)
        kw = {} if impl == "reference" or seed is None else dict(device_rng=DeviceRNG(seed))
        sampler = pop.PopulationSliceSampler(popsize=popsize, nsteps=nsteps,
                                             generate_direction=pop.generate_mixture_random_direction, scale=0.1, **kw)
        if hasattr(sampler, "max_rounds"):
            sampler.max_rounds = 256 if label.endswith("harvested") else 1
        ncalls = 60 if popsize <= 10000 else 25
        for _ in range(5):
            sampler.__next__(region, Lmin, u, Ls, transform, loglike)
        t0 = time.perf_counter()
        nc = nfound = 0
        for _ in range(ncalls):
            r = sampler.__next__(region, Lmin, u, Ls, transform, loglike)
            nc += r[3]
            nfound += r[0] is not None
        dt = (time.perf_counter() - t0) / ncalls
        row = dict(impl=impl, mode=label, popsize=popsize, d=d, ms_per_call=dt * 1e3,
                   walker_steps_per_s=(popsize / dt) if not label.endswith("harvested") else None,
                   likelihood_evals_per_call=nc / ncalls, likelihood_evals_per_s=nc / ncalls / dt, found=nfound)
        out.append(row)
        print(json.dumps(row), flush=True)
json.dump(out, open(os.path.join("gpurun_out" if os.path.isdir("gpurun_out") else ".", "walk_bench_%s.json" % impl), "w"), indent=1)
