#!/usr/bin/env python3
"""Build ultranest_amd/libmlfriends_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python ultranest_amd/csrc/build.py [--force]

-ffp-contract=off is part of the arithmetic contract (no FMA may be formed in the distance
accumulation; the reference's x86-64 build has none).  The library is built IN-TREE so that it
travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libmlfriends_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["mlf_scan.hip", "mlf_boot.hip", "mlf_prep.hip", "mlf_misc.hip", "mlf_filter.hip", "mlf_sweep.hip", "mlf_sweepmin.hip", "mlf_mid.hip", "mlf_fused.hip", "mlf_prep3.hip", "mlf_prep4.hip", "mlf_prep64.hip", "mlf_sample.hip", "mlf_walk.hip", "mlf_walk_api.hip", "mlf_netiter.hip", "mlf_comm.hip", "mlf_small.hip", "mlf_wide.hip", "mlf_api.hip"]
HEADERS = ["mlf_common.hpp", "mlf_dpp_dev.hpp", "mlf_recheck_dev.hpp", "mlf_ell_exact.hpp", "mlf_misc.hpp", "mlf_filter.hpp", "mlf_filter_dev.hpp", "mlf_prep3.hpp", "mlf_prep4.hpp", "mlf_prep64.hpp", "mlf_small.hpp", "mlf_sample.hpp", "mlf_walk.hpp", "mlf_ctx.hpp", "mlf_philox_dev.hpp", "mlf_loglike_dev.hpp", os.path.join("..", "..", "include", "mlfriends_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-result"]


# keep MFMA accumulators in VGPRs: the epilogue reads every accumulator, AGPR form costs one
# v_accvgpr_read per value (40 % of the filter kernel's VALU instructions)
EXTRA_FLAGS = {"mlf_filter.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "mlf_sweep.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "mlf_sweepmin.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "mlf_mid.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "mlf_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def source_hash():
    """16 hex digits over the kernel sources and headers (what the library is built from): profiles record it next to their
    counters so that a figure measured on another tree can be told apart (bench.py: roofline.traffic)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + [x for x in HEADERS if not x.startswith("..")]):
        with open(os.path.join(HERE, name), "rb") as fh:
            h.update(name.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            extra = EXTRA_FLAGS.get(src, [])
            jobs.append([hipcc] + FLAGS + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
