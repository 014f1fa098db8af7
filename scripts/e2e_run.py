"""End-to-end nested sampling runs on the device-resident path (region proposals, prior transform,
likelihood and threshold cut on the GPU; StaticNestedSampler drives).  C3-like: eggbox likelihood as a
HIP kernel, N = 1000 live points."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from ultranest_amd import likelihoods  # noqa: E402
from ultranest_amd.harness import StaticNestedSampler  # noqa: E402
from ultranest_amd.regions import DeviceRNG  # noqa: E402

out = []
CASES = [(2, 1000, 65536, 200000)]
if len(sys.argv) > 1:      # the 10-d eggbox (5^10 modes) is optional: it needs minutes
    CASES.append((10, 1000, 262144, int(sys.argv[1])))
for d, nlive, ndraw, max_iters in CASES:
    s = StaticNestedSampler(d, likelihoods.eggbox_loglike, transform=likelihoods.eggbox_transform, num_live_points=nlive,
                            ndraw=ndraw, seed=1, device_rng=DeviceRNG(3))
    t0 = time.perf_counter()
    res = s.run(dlogz=0.5, max_iters=max_iters)
    dt = time.perf_counter() - t0
    res.update(d=d, nlive=nlive, ndraw=ndraw, seconds=dt, iterations_per_s=res["niter"] / dt,
               likelihood_evaluations_per_s=res["ncall"] / dt, proposals_per_s=res["ncall_region"] / dt,
               finished=res["niter"] < max_iters)
    out.append(res)
    print(json.dumps(res), flush=True)
json.dump(out, open("gpurun_out/e2e_run.json", "w"), indent=1)
