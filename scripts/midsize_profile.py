"""Device-resident MLFriends.inside at batch sizes between the tiny-call path and the headline batch (C5 region).
    python scripts/midsize_profile.py [batch ...] [name=value ...]      (options for mlf_set_option)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or (300, 1024, 4096, 16384, 65536, 131072, 262144)
for a in sys.argv[1:]:
    if "=" in a:
        _lib.set_option(a.split("=")[0], int(a.split("=")[1]))
warm = bench.proposals_in_ellipsoid(region, 262144, 999, dev)
wmask = torch.empty(262144, dtype=torch.uint8, device=dev)
for _ in range(200):     # the chip's clock settles over the first tens of milliseconds of load
    handle.inside_dev(warm.data_ptr(), 262144, wmask.data_ptr(), stream)
# one throw-away pass over the first size: in some processes the first measurement after the warm-up carries a one-time ~60 ms
# (263-300 us per call at 300 proposals instead of 27; the same size measured again in the same process: 27-28 -- profiles/README)
first_pass = True
for p in [sizes[0]] + list(sizes):
    pts = bench.proposals_in_ellipsoid(region, p, 1000, dev)
    mask = torch.empty(p, dtype=torch.uint8, device=dev)
    for _ in range(20):
        handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    st = handle.debug_stats()
    if first_pass:
        first_pass = False
        continue
    print(json.dumps(dict(batch=p, us_per_call=round(dt * 1e6, 2), proposals_per_s=round(p / dt), accept=float(mask.float().mean().item()),
                          stage_cycles=st.get("uncertain_stage_cycles"), stamp7=st.get("stamp7"))), flush=True)
