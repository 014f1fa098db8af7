"""Size check beyond the headline batch: P = 8e6 proposals x d = 50 (3.2 GB) through MLFriends.inside on
the device, MFMA pre-filter (phased) against the exact scan, plus a 2e5-live-point region at small P."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
out = {}
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
P = 8000000
pts = bench.proposals_in_ellipsoid(region, P, 5, dev)
stream = torch.cuda.current_stream().cuda_stream
masks = {}
for filt in (1, 0):
    _lib.set_option("filter", filt)
    mask = torch.empty(P, dtype=torch.uint8, device=dev)
    handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    handle.inside_dev(pts.data_ptr(), P, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    out["P8e6_filter%d_ms" % filt] = (time.perf_counter() - t0) * 1e3
    masks[filt] = mask
_lib.set_option("filter", 1)
out["P8e6_masks_equal"] = bool((masks[0] == masks[1]).all().item())
out["P8e6_accept"] = float(masks[1].float().mean().item())
out["P8e6_proposals_per_s"] = P / (out["P8e6_filter1_ms"] * 1e-3)
del pts, masks

# many live points: N = 200000, d = 10
import ultranest_amd.mlfriends as M  # noqa: E402
rs = np.random.RandomState(2)
N, d = 200000, 10
ub = 0.5 + 0.05 * rs.normal(size=(N, d))
layer = M.AffineLayer()
layer.optimize(ub, ub)
big = M.MLFriends(ub, layer)
big.maxradiussq, big.enlarge = 0.02, 1.2          # fixed radius: the bootstrap is not what is checked here
big.create_ellipsoid()
q = np.clip(ub[:20000] + 0.01 * rs.normal(size=(20000, d)), 1e-6, 1 - 1e-6)
res = {}
for filt in (1, 0):
    _lib.set_option("filter", filt)
    t0 = time.perf_counter()
    res[filt] = big.inside(q)
    out["N2e5_filter%d_ms" % filt] = (time.perf_counter() - t0) * 1e3
_lib.set_option("filter", 1)
out["N2e5_masks_equal"] = bool(np.array_equal(res[0], res[1]))
out["N2e5_accept"] = float(res[1].mean())
print(json.dumps(out, indent=1))
