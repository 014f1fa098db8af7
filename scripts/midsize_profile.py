"""Device-resident MLFriends.inside at batch sizes between the tiny-call path and the headline batch (C5 region)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
for p in [int(a) for a in sys.argv[1:]] or (1024, 4096, 16384, 65536, 262144):
    pts = bench.proposals_in_ellipsoid(region, p, 1000, dev)
    mask = torch.empty(p, dtype=torch.uint8, device=dev)
    for _ in range(3):
        handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        handle.inside_dev(pts.data_ptr(), p, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(json.dumps(dict(batch=p, us_per_call=dt * 1e6, proposals_per_s=p / dt, accept=float(mask.float().mean().item()))), flush=True)
