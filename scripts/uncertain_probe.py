"""Raw diagnostic words of k_uncertain's workgroup 0 after a headline batch (what they hold depends on the build: the production
build writes stage stamps, the probe builds of scripts/gpu_r06_n.sh per-wave or per-tile durations)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from ultranest_amd import _lib  # noqa: E402
from ultranest_amd._lib import check, ptr  # noqa: E402

dev = torch.device("cuda", 0)
u, region = bench.build_region(None)
handle = region._dev.sync(region, True)
stream = torch.cuda.current_stream().cuda_stream
batch = bench.proposals_in_ellipsoid(region, bench.NPROPOSALS, 1000, dev)
mask = torch.empty(bench.NPROPOSALS, dtype=torch.uint8, device=dev)
if os.environ.get("MLF_VARIANT"):
    _lib.set_option("fused_variant", int(os.environ["MLF_VARIANT"]))
for rep in range(4):
    for i in range(20):
        handle.inside_dev(batch.data_ptr(), bench.NPROPOSALS, mask.data_ptr(), stream)
    torch.cuda.synchronize()
    out = np.zeros(18, dtype=np.uint64)
    check(_lib.lib().mlf_region_debug_stats(handle._h, ptr(out), 18))
    print(json.dumps({"words": [int(v) for v in out[8:16]], "uncertain_queries": int(out[6]), "uncertain_pairs": int(out[2])}), flush=True)
