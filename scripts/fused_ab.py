"""k_prep_sweep variants against each other INSIDE one process (boxes of the pool differ by 8 % in this kernel), C5 batch;
stage stamps of single waves.
    python scripts/fused_ab.py [steps] [name:waves:variant ...]      variant = fused_variant option value"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
configs = [a.split(":") for a in sys.argv[2:]] or [["base", "8", "0"], ["waves4", "4", "0"], ["dma", "8", "1"]]
dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
if os.environ.get("MLF_AB_SET", "E") == "U":      # uniform draws in the unit cube: (nearly) everything fails the ellipsoid test
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    batches = [torch.rand(bench.NPROPOSALS, bench.NDIM, dtype=torch.float64, device=dev, generator=gen) for k in range(3)]
else:
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
run(300)
for rep in range(3):
    for name, waves, variant in configs:
        _lib.set_option("fused_waves", int(waves))
        _lib.set_option("fused_variant", int(variant))
        run(60)
        t0 = time.perf_counter()
        run(steps)
        dt = (time.perf_counter() - t0) / steps
        ok = all(bool((a == b).all().item()) for a, b in zip(exact, masks))
        _lib.set_option("time_filter_launches", 1)
        handle.timing_filter_launches()
        run(30)
        lm = handle.timing_filter_launch_ms()
        _lib.set_option("time_filter_launches", 0)
        nl = max(1, len(lm) // 30)
        per = [round(float(lm[i::nl].mean()), 5) for i in range(nl)] if len(lm) else []
        row = dict(name=name, fused_waves=int(waves), variant=int(variant), ms_per_step=round(dt * 1e3, 5), masks_equal_exact=ok, launch_ms=per)
        if rep == 2:
            stamps = {}
            for blk in (3, 600) if int(waves) == 8 else (3, 1200):
                handle.fused_stamps(blk)
                run(4)
                stamps[blk] = handle.fused_stamps(None)
            row["stamps_cycles"] = stamps
        print(json.dumps(row), flush=True)
_lib.set_option("fused_waves", 8)
_lib.set_option("fused_variant", 0)
