import cProfile, io, pstats, sys, time
sys.path.insert(0, ".")
from ultranest_amd import likelihoods
from ultranest_amd.harness import StaticNestedSampler
from ultranest_amd.regions import DeviceRNG
s = StaticNestedSampler(10, likelihoods.eggbox_loglike, transform=likelihoods.eggbox_transform, num_live_points=1000,
                        ndraw=262144, seed=1, device_rng=DeviceRNG(3))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = s.run(dlogz=0.5, max_iters=int(sys.argv[1]))
pr.disable()
print(res, time.perf_counter() - t0)
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
print("\n".join(l[:160] for l in out.getvalue().splitlines()[:55]))
