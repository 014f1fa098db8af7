"""A/B of pre-filter options on the C5 membership pipeline: per setting, wall time per step, the event time of every
matrix-kernel launch, the batch counters, and the mask against the exact scan's.  The settings are cycled `rounds` times
after a warm-up (the chip's clock settles over the first few hundred launches).
    python scripts/sweep_ab.py [steps] name=v[,name=v...] [name=v...] ...      (one argument per setting)
Environment: MLF_AB_P proposals per batch, MLF_AB_N / MLF_AB_D live points and dimensionality of the region (default: the
headline's), MLF_AB_R2_SCALE factor on the squared radius (0.2: set N of scripts/config_bench.py, almost no neighbour within
reach), MLF_AB_ROUNDS passes over the settings."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
settings = sys.argv[2:] or ["filter_sweep=0", "filter_sweep=1", "filter_sweep=2"]
rounds = int(os.environ.get("MLF_AB_ROUNDS", "2"))
dev = torch.device("cuda", 0)
bench.N_LIVE = int(os.environ.get("MLF_AB_N", bench.N_LIVE))
bench.NDIM = int(os.environ.get("MLF_AB_D", bench.NDIM))
u, region = bench.build_region(None)
region.maxradiussq *= float(os.environ.get("MLF_AB_R2_SCALE", "1"))
handle = region._dev.sync(region, True)
P = int(os.environ.get("MLF_AB_P", bench.NPROPOSALS))
pts = bench.proposals_in_ellipsoid(region, P, 1000, dev)
mask = torch.empty(P, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream

_lib.set_option("filter", 0)
handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
torch.cuda.synchronize()
exact = mask.clone()
_lib.set_option("filter", 1)
_lib.set_option("time_filter_launches", 1)
for _ in range(100):
    handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
torch.cuda.synchronize()
for rnd in range(rounds):
    for setting in settings:
        pairs = [kv.split("=") for kv in setting.split(",")]
        for k, v in pairs:
            _lib.set_option(k, int(v))
        for _ in range(5):
            handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
        torch.cuda.synchronize()
        handle.timing_filter_launches()
        t0 = time.perf_counter()
        for _ in range(steps):
            handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms = handle.timing_filter_launch_ms()
        per = len(ms) // steps if steps else 0
        launches = [float(np.mean(ms[i::per])) for i in range(per)] if per else []
        print(json.dumps(dict(setting=setting, ms_per_step=round(dt * 1e3, 4), filter_launch_ms=[round(x, 4) for x in launches],
                              mask_equals_exact=bool((mask == exact).all().item()), accept=round(float(mask.float().mean().item()), 5),
                              stats=handle.debug_stats())), flush=True)
