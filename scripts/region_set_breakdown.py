"""mlf_region_set with the live points already on the device (what the device-resident rebuild calls last): median wall
time per call at C5; under rocprofv3 --kernel-trace --stats the kernels inside it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import kernels  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
layer = region.transformLayer
n, d = u.shape
live = torch.from_numpy(np.ascontiguousarray(u)).to(dev)
amax = float(np.abs(u - layer.ctr).max())
h = kernels.DeviceRegion()


def call(hint):
    h.set_from_device(live, n, d, 0, layer.ctr, layer.T, None, region.ellipsoid_center, region.ellipsoid_invcov,
                      region.enlarge, region.maxradiussq, use_scan=True, live_space=1, live_amax=amax if hint else None)


out = {}
for hint in (True, False):
    for _ in range(20):
        call(hint)
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        call(hint)
        ts.append(time.perf_counter() - t0)
    out["set_from_device, extent %s" % ("given" if hint else "fetched back")] = round(float(np.median(ts)) * 1e6, 1)
# the host-pointer flavour (default rebuild): the same call with numpy live points
h2 = kernels.DeviceRegion()
ts = []
for i in range(60):
    t0 = time.perf_counter()
    h2.set(u, 0, layer.ctr, layer.T, None, region.ellipsoid_center, region.ellipsoid_invcov, region.enlarge, region.maxradiussq,
           use_scan=True, live_space=1)
    ts.append(time.perf_counter() - t0)
out["set, host live points"] = round(float(np.median(ts[10:])) * 1e6, 1)
print(json.dumps(out, indent=1))
