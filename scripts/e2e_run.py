"""End-to-end nested sampling runs on the device-resident path (region proposals, prior transform,
likelihood and threshold cut on the GPU; StaticNestedSampler drives).  C3-like: eggbox likelihood as a
HIP kernel, N = 1000 live points."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
t_import = time.perf_counter()
from ultranest_amd import likelihoods  # noqa: E402
from ultranest_amd.harness import StaticNestedSampler  # noqa: E402
from ultranest_amd.regions import DeviceRNG  # noqa: E402

out = []
IMPORT_S = time.perf_counter() - t_import     # package import incl. loading libmlfriends_hip.so (the first device call comes later)


TRUTH = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g15_eggbox_logz.json")))["cases"]


def laplace_logz(d):
    """Eggbox evidence by Laplace's method: 5^d / 2 peaks (boundary peaks count half per axis) of height 243,
    each a Gaussian of variance 8 / 810 per axis, prior volume (10 pi)^d.  d = 2: 235.85 (quadrature: 235.88)."""
    return np.log(5.0**d / 2) + 243.0 + 0.5 * d * np.log(2 * np.pi * 8.0 / 810.0) - d * np.log(10 * np.pi)


CASES = [(2, 1000, 65536, 200000)]
ARGS = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
if "region10" in ARGS:      # the 10-d eggbox (5^10 modes) by region sampling is optional: it needs many minutes
    CASES.append((10, 1000, 262144, int(ARGS["region10"])))
POP = [(2, 1000, 1024, 10), (10, 1000, 1024, 40)]
if "nsteps10" in ARGS:      # nsteps10=40,80,160: the d = 10 run once per value (how the evidence moves with the chain length)
    POP = [(2, 1000, 1024, 10)] + [(10, 1000, 1024, int(n)) for n in ARGS["nsteps10"].split(",")]
MAX_ROUNDS = int(ARGS.get("max_rounds", "256"))
for d, nlive, ndraw, max_iters in CASES:
    s = StaticNestedSampler(d, likelihoods.eggbox_loglike, transform=likelihoods.eggbox_transform, num_live_points=nlive,
                            ndraw=ndraw, seed=1, device_rng=DeviceRNG(3))
    t0 = time.perf_counter()
    res = s.run(dlogz=0.5, max_iters=max_iters)
    dt = time.perf_counter() - t0
    res.update(d=d, nlive=nlive, ndraw=ndraw, laplace_logz=float(laplace_logz(d)), truth_logz=TRUTH[str(d)]["logz"], seconds=dt, iterations_per_s=res["niter"] / dt,
               likelihood_evaluations_per_s=res["ncall"] / dt, proposals_per_s=res["ncall_region"] / dt,
               finished=res["niter"] < max_iters,
               # where the wall time goes (StaticNestedSampler.phases)
               phases=dict(s.phases, import_s=IMPORT_S))
    out.append(res)
    print(json.dumps(res), flush=True)
# C3 proper: 10-d eggbox (5^10 modes).  Region rejection sampling cannot follow the volume there; the
# population slice sampler (device-resident walkers, Philox, likelihood evaluated in place) can.
import ultranest_amd.popstepsampler as pop  # noqa: E402
for d, nlive, popsize, nsteps in POP:
    step = pop.PopulationSliceSampler(popsize=popsize, nsteps=nsteps, generate_direction=pop.generate_mixture_random_direction,
                                      scale=1.0, device_rng=DeviceRNG(7))
    step.max_rounds = MAX_ROUNDS
    s = StaticNestedSampler(d, likelihoods.eggbox_loglike, transform=likelihoods.eggbox_transform, num_live_points=nlive,
                            seed=1, stepsampler=step)
    t0 = time.perf_counter()
    res = s.run(dlogz=0.5, max_iters=400000)
    dt = time.perf_counter() - t0
    res.update(d=d, nlive=nlive, laplace_logz=float(laplace_logz(d)), sampler="PopulationSliceSampler(popsize=%d, nsteps=%d, mixture directions)" % (popsize, nsteps),
               seconds=dt, iterations_per_s=res["niter"] / dt, likelihood_evaluations_per_s=res["ncall"] / dt,
               far_enough_fraction=float(step.far_enough_fraction), finished=res["niter"] < 400000, phases=dict(s.phases),
               max_rounds=MAX_ROUNDS, truth_logz=TRUTH[str(d)]["logz"],
               sigmas_from_truth=(res["logz"] - TRUTH[str(d)]["logz"]) / res["logzerr"])
    out.append(res)
    print(json.dumps(res), flush=True)
json.dump(out, open("gpurun_out/e2e_run.json", "w"), indent=1)
