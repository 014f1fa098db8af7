"""Where the time of `region.inside(active_u)` (N = 4000 host points, C5) goes: the C-ABI call on a host pointer, the same
batch device-resident, and the copies alone (pageable / pinned, torch as the stopwatch's plumbing)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
pts = np.ascontiguousarray(np.asarray(region.u, dtype=float))
n = len(pts)


def med(f, reps=300):
    for _ in range(30):
        f()
    each = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        each.append(time.perf_counter() - t0)
    return round(float(np.median(each)) * 1e6, 2)


out = {}
out["region.inside(host numpy)"] = med(lambda: region.inside(pts))
out["handle.inside(host numpy)"] = med(lambda: handle.inside(pts))
d_pts = torch.from_numpy(pts).to(dev)
d_mask = torch.empty(n, dtype=torch.uint8, device=dev)


def dev_call():
    handle.inside_dev(d_pts.data_ptr(), n, d_mask.data_ptr(), stream)
    torch.cuda.synchronize()


out["inside_dev + synchronize"] = med(dev_call)
h_mask = torch.empty(n, dtype=torch.uint8).pin_memory()


def dev_call_mask():
    handle.inside_dev(d_pts.data_ptr(), n, d_mask.data_ptr(), stream)
    h_mask.copy_(d_mask, non_blocking=True)
    torch.cuda.synchronize()


out["inside_dev + mask to pinned host + synchronize"] = med(dev_call_mask)
t_page = torch.from_numpy(pts)
t_pin = torch.from_numpy(pts).pin_memory()


def up_page():
    d_pts.copy_(t_page, non_blocking=True)
    torch.cuda.synchronize()


def up_pin():
    d_pts.copy_(t_pin, non_blocking=True)
    torch.cuda.synchronize()


out["H2D 1.6 MB pageable + synchronize"] = med(up_page)
out["H2D 1.6 MB pinned + synchronize"] = med(up_pin)
out["host memcpy 1.6 MB into pinned"] = med(lambda: t_pin.copy_(t_page))
print(json.dumps(out, indent=1))
if "--save" in sys.argv:
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/active_u_breakdown.json", "w"), indent=1)
