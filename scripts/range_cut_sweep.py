"""Tile-range cuts of the three-range min-only sweep at the headline size, all settings inside ONE process (C5 batch, set E).
    python scripts/range_cut_sweep.py [steps]
filter_second_range_pct = where the last range starts (share of the tiles), filter_first_range_pct = share of THAT taken by the first
range (inside k_prep_sweep)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
batches = [bench.proposals_in_ellipsoid(region, bench.NPROPOSALS, 1000 + 7919 * k, dev) for k in range(3)]
masks = [torch.empty(bench.NPROPOSALS, dtype=torch.uint8, device=dev) for _ in range(3)]


def run(n):
    for i in range(n):
        handle.inside_dev(batches[i % 3].data_ptr(), bench.NPROPOSALS, masks[i % 3].data_ptr(), stream)
    torch.cuda.synchronize()


_lib.set_option("filter", 0)
run(3)
exact = [m.clone() for m in masks]
_lib.set_option("filter", 1)
run(200)
rows = []
for rep in range(2):
    for second in (40, 45, 50, 55, 60, 65):
        for first in (20, 25, 30, 35, 40):
            _lib.set_option("filter_first_range_pct", first)
            _lib.set_option("filter_second_range_pct", second)
            run(30)
            t0 = time.perf_counter()
            run(steps)
            dt = (time.perf_counter() - t0) / steps
            ok = all(bool((a == b).all().item()) for a, b in zip(exact, masks))
            rows.append(dict(rep=rep, first=first, second=second, ms_per_step=round(dt * 1e3, 5), masks_equal_exact=ok))
            print(json.dumps(rows[-1]), flush=True)
best = sorted(rows, key=lambda r: r["ms_per_step"])[:6]
print(json.dumps({"best": best}))
