import sys, time
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
layer = m.AffineLayer(); layer.optimize(u, u)
region = m.MLFriends(u, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=10, rng=np.random.RandomState(2))
region.create_ellipsoid()
acc = defaultdict(float)
def timed(cls, name):
    fn = getattr(cls, name)
    def wrap(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t0; return r
    setattr(cls, name, wrap)
for name in ("set_live", "step_dev", "set_direction_data", "set_layer"):
    timed(pop._Walkers, name)
timed(pop.PopulationSliceSampler, "_sync_region")
timed(pop.PopulationSliceSampler, "_next_on_device")
for popsize in (100, 10000):
    for graph in (True, False):
        s = pop.PopulationSliceSampler(popsize=popsize, nsteps=10, generate_direction=pop.generate_mixture_random_direction, scale=0.1, device_rng=DeviceRNG(7))
        s.use_graph = graph
        for _ in range(5): s.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
        acc.clear(); n = 200
        t0 = time.perf_counter()
        for _ in range(n): s.__next__(region, Lmin, u, Ls, likelihoods.rosenbrock_transform, likelihoods.rosenbrock_loglike)
        tot = (time.perf_counter() - t0) / n
        print("popsize %d graph %s: %.1f us/call;" % (popsize, graph, tot * 1e6), ", ".join("%s %.1f" % (k, v / n * 1e6) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
