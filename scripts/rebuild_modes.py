"""Steady-state rebuild at C5 (N = 4000, d = 50, LocalAffineLayer, 30 bootstraps): the default (reference-bit-preserving)
path against the opt-in device-resident path.  python scripts/rebuild_modes.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ultranest_amd.mlfriends as M  # noqa: E402
from ultranest_amd.harness import RegionUpdater  # noqa: E402

u, region = bench.build_region(None)
out = {}
for name, dev in (("default", False), ("device_resident", True), ("default_again", False), ("device_resident_again", True)):
    rs = np.random.RandomState(7)
    upd = RegionUpdater(bench.NDIM, region_class=M.MLFriends, transform_layer_class=M.LocalAffineLayer, device_resident=dev, freeze_gc=True)
    np.random.seed(11)
    upd.update(u, nbootstraps=bench.NBOOT, minvol=0.)
    ts = []
    for rep in range(12):
        u2 = u.copy()
        u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
        t0 = time.perf_counter()
        changed = upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[name] = {"median_ms": float(np.median(ts[2:])), "each_ms": [round(t, 3) for t in ts], "accepted_last": bool(changed),
                 "r2": upd.region.maxradiussq, "enlarge": upd.region.enlarge}
print(json.dumps(out, indent=1))

# where the device-resident path spends its time (each checkpoint behind a device synchronisation)
upd = RegionUpdater(bench.NDIM, region_class=M.MLFriends, transform_layer_class=M.LocalAffineLayer, device_resident=True, freeze_gc=True)
np.random.seed(11)
upd.update(u, nbootstraps=bench.NBOOT, minvol=0.)
rs = np.random.RandomState(7)
acc = {}
for rep in range(8):
    u2 = u.copy()
    u2[:bench.N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(bench.N_LIVE // 10, bench.NDIM))
    if upd._device_rebuild is not None:
        upd._device_rebuild.trace = []
    t0 = time.perf_counter()
    upd.update(u2, nbootstraps=bench.NBOOT, minvol=0.)
    t1 = time.perf_counter()
    tr = upd._device_rebuild.trace
    if rep >= 2:
        acc.setdefault("before next_region", []).append((tr[0][1] - t0) * 1e3)
        for (la, ta), (lb, tb) in zip(tr[:-1], tr[1:]):
            acc.setdefault(lb, []).append((tb - ta) * 1e3)
        acc.setdefault("after next_region", []).append((t1 - tr[-1][1]) * 1e3)
print(json.dumps({k: round(float(np.median(v)), 3) for k, v in acc.items()}, indent=1))
