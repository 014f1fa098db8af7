"""Throughput of the path the reference calls: MLFriends.inside(host numpy) -> host bool mask
(reference mlfriends.pyx:1186-1211, call site integrator.py:1776-1804), C5: N = 4000, d = 50, P = 10^6.
Pageable and pinned source buffers; the PCIe bound is 8 d bytes per proposal at the measured H2D rate."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
import ultranest_amd.mlfriends as M  # noqa: E402

N, D, P = 4000, 50, 1000000
rs = np.random.RandomState(1)
u = 0.5 + 0.05 * rs.normal(size=(N, D))
layer = M.AffineLayer()
layer.optimize(u, u)
region = M.MLFriends(u, layer)
region.maxradiussq, region.enlarge = region.compute_enlargement(nbootstraps=30, rng=rs)
region.create_ellipsoid()
z = rs.normal(size=(P, D))
z /= np.linalg.norm(z, axis=1, keepdims=True)
z *= region.enlarge ** 0.5 * rs.uniform(size=(P, 1)) ** (1.0 / D)
pts = region.ellipsoid_center + z @ region.ellipsoid_axes_T
pinned_t = torch.empty((P, D), dtype=torch.float64).pin_memory()
pinned = pinned_t.numpy()
pinned[:] = pts
res = {}
for name, arr in (("pageable", pts), ("pinned", pinned)):
    region.inside(arr)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        m = region.inside(arr)
        ts.append(time.perf_counter() - t0)
    best, med = min(ts), float(np.median(ts))
    res[name] = dict(ms_median=med * 1e3, ms_best=best * 1e3, proposals_per_s=P / med, accept=float(m.mean()))
# raw H2D rate of the same 400 MB for the PCIe bound
dev = torch.device("cuda", 0)
dst = torch.empty((P, D), dtype=torch.float64, device=dev)
for name, src in (("pageable", torch.from_numpy(pts)), ("pinned", pinned_t)):
    dst.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    res[name]["h2d_GBps"] = P * D * 8 / dt / 1e9
    res[name]["pcie_bound_proposals_per_s"] = P / dt
    res[name]["fraction_of_pcie_bound"] = res[name]["proposals_per_s"] * dt / P
print(json.dumps(res))
