"""Stage stamps of single k_sweep_min workgroups at the headline size (wave 0, shader cycles from the workgroup's start):
0 start, 1 operands and first two tiles asked for, 2 first tile done, 3 tile loop done, 4 fates known, 5 / 6 past the two barriers,
7 compaction stores issued.   python scripts/sweep_stamps.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
batch = bench.proposals_in_ellipsoid(region, bench.NPROPOSALS, 1000, dev)
mask = torch.empty(bench.NPROPOSALS, dtype=torch.uint8, device=dev)


def run(n):
    for i in range(n):
        handle.inside_dev(batch.data_ptr(), bench.NPROPOSALS, mask.data_ptr(), stream)
    torch.cuda.synchronize()


run(50)
for launch, blocks in ((1, (0, 100, 255, 256, 400, 511, 512, 700, 1019)), (2, (0, 100, 255, 256, 400, 509))):
    for b in blocks:
        handle.fused_stamps(launch * 1000000 + b)
        run(4)
        st = handle.fused_stamps(None)
        rt = None
        if st and st[8] is not None and st[10] is not None:      # 100 MHz ticks -> microseconds from workgroup 0's start
            rt = {"wg0_end_us": (st[9] - st[8]) / 100.0, "start_us": (st[10] - st[8]) / 100.0, "end_us": (st[11] - st[8]) / 100.0}
        print(json.dumps({"launch": "range %d" % (launch + 1), "block": b, "cycles": st[:8] if st else None, "realtime": rt}), flush=True)
