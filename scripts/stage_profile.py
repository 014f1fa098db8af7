"""C5 membership pipeline only (no rebuilds, no comparison passes): counters of the batch and a clean kernel list for
rocprofv3.  python scripts/stage_profile.py [steps] [option=value ...]"""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for opt in sys.argv[2:]:   # name=value pairs for mlf_set_option
    from ultranest_amd import _lib as _k
    _k.set_option(opt.split("=")[0], int(opt.split("=")[1]))
dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
pts = bench.proposals_in_ellipsoid(region, bench.NPROPOSALS, 1000, dev)
mask = torch.empty(bench.NPROPOSALS, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    handle.inside_dev(pts.data_ptr(), bench.NPROPOSALS, mask.data_ptr(), stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    handle.inside_dev(pts.data_ptr(), bench.NPROPOSALS, mask.data_ptr(), stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps(dict(ms_per_step_untimed=dt * 1e3, accept=float(mask.float().mean().item()), stats=handle.debug_stats())))
