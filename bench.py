#!/usr/bin/env python3
"""Headline benchmark of the MLFriends hot path on MI355X (BASELINE.json metric):
proposal points filtered per second through MLFriends.inside + region-rebuild ms, N=4000 live
points, d=50.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Launched plainly with --gpus N > 1 the script starts its N ranks itself (torch.distributed.run on
127.0.0.1); on a box with fewer than N devices the ranks share the devices and use gloo collectives,
so that the N-rank control flow can be exercised anywhere.

One "step" = one pass of MLFriends.inside (wrapping-ellipsoid test -> whitening -> neighbour
scan) over one batch of P = 10^6 synthetic proposals that is already resident in HBM when the
timed region starts.  Weak scaling: every rank (one process per GPU) filters its own batch of P
proposals against the replicated region; there is no collective in the data path.  The region
itself is built with the sharded bootstrap (RCCL max-all-reduce of 3 doubles when N > 1).

Workload (configuration C5): live points u = 0.5 + 0.05*normal (RandomState(1)),
AffineLayer, 30 bootstrap rounds, proposal set "E" = uniform draws inside the wrapping
ellipsoid (every proposal passes the ellipsoid test and reaches the neighbour scan).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_LIVE, NDIM, NPROPOSALS, NBOOT = 4000, 50, 1000000, 30
FP64_VALU_PEAK_TFLOPS = 39.3     # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz, one flop per non-fused v_*_f64
F16_MFMA_PEAK_TFLOPS = 2500.0    # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
PROBE_F16_MFMA_TFLOPS = 1620.0   # what v_mfma_f32_32x32x16_f16 sustains ALONE on this part on random binary16 operands (4 chains,
# operands in registers, 2 waves per SIMD, pipe 98 % occupied, clock 1.59-1.60 GHz measured inside the kernel:
# scripts/probes/sweep_src_probe.hip, profiles/r04_sweep_src_probe.json): the part's power limit, not the issue slots, sets it
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_FLOPS = 2 * 32 * 32 * 16   # one v_mfma_f32_32x32x16_f16
CLOCK_SETTLE_STEPS = 100         # untimed steps in front of the W warm-up steps of a timed loop (see run_steps)
NBATCHES = 3                     # distinct 400 MB proposal batches a timed loop visits in turn (1.2 GB > the 256 MB Infinity Cache)
PREP_MFMA_PER_GROUP = 42         # v_mfma_f32_32x32x16_f16 of the per-proposal stage per 32 proposals at d = 50: 3 partial products
# (hi hi, hi lo, lo hi) x [6 k-steps of the triangular L^T (ellipsoid form) + 8 of T^T (whitening)]
PREP_MFMA_PER_GROUP_SAME_FORM = 24   # ... where the ellipsoid's matrix is T T^T of the layer (AffineLayer, one cluster -- this workload):
# k_prep_sweep<.., SQ> reads the ellipsoid form off the whitening chain, the L^T chain is not executed (debug_stats: same_quadratic_form)
SET_BYTES_PER_SLOT = lambda ks: ks * 32 + 16   # noqa: E731 -- a compacted proposal: K = 16 ks binary16 columns + T_lo, T_hi, index, minimum


def build_region(group):
    import ultranest_amd.mlfriends as M
    from ultranest_amd import distributed
    rs = np.random.RandomState(1)
    u = 0.5 + 0.05 * rs.normal(size=(N_LIVE, NDIM))
    assert np.logical_and(u > 0, u < 1).all()
    layer = M.AffineLayer()
    layer.optimize(u, u)
    region = M.MLFriends(u, layer)
    distributed.update_region_bootstrap(region, NBOOT, minvol=0., group=group, rng=rs)
    region.create_ellipsoid(minvol=0.)
    return u, region


def proposals_in_ellipsoid(region, p, seed, dev):
    """set E on the device: center + (z/|z| * sqrt(enlarge) * U^(1/d)) @ axes^T"""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    d = len(region.ellipsoid_center)
    z = torch.randn(p, d, dtype=torch.float64, device=dev, generator=g)
    z /= z.norm(dim=1, keepdim=True)
    z *= region.enlarge ** 0.5 * torch.rand(p, 1, dtype=torch.float64, device=dev, generator=g) ** (1.0 / d)
    pts = torch.as_tensor(region.ellipsoid_center, device=dev) + z @ torch.as_tensor(region.ellipsoid_axes_T.copy(), device=dev)
    return pts.contiguous()


def time_rebuild(u, group, device_resident=False):
    """Steady-state region rebuild (harness counterpart of the driver's _update_region: LocalAffineLayer
    create_new incl. clustering + subtract_nearby, region ctor, 30 bootstrap rounds, create_ellipsoid,
    inside(live points), shrink test) on the SURVEY 8d rebuild input."""
    import ultranest_amd.mlfriends as M
    from ultranest_amd.harness import RegionUpdater
    rs = np.random.RandomState(7)
    upd = RegionUpdater(NDIM, region_class=M.MLFriends, transform_layer_class=M.LocalAffineLayer, group=group,
                        device_resident=device_resident, freeze_gc=True)
    np.random.seed(11)
    t0 = time.perf_counter()
    upd.update(u, nbootstraps=NBOOT, minvol=0.)
    first_ms = (time.perf_counter() - t0) * 1e3
    steady = []
    for rep in range(7):     # median of 7 (all seven are reported as rebuild_ms_each)
        u2 = u.copy()
        u2[:N_LIVE // 10] = 0.5 + 0.045 * rs.normal(size=(N_LIVE // 10, NDIM))
        t0 = time.perf_counter()
        upd.update(u2, nbootstraps=NBOOT, minvol=0.)
        steady.append((time.perf_counter() - t0) * 1e3)
    return first_ms, float(np.median(steady)), [float(x) for x in steady]


def hbm_probe_ceiling(achieved_gbps):
    """What hand-written streaming kernels reach on this pool's boxes (scripts/probes/hbm_stream_probe.hip, a committed profile of
    an earlier run of this round -- imported, NOT measured in this run): the 8 TB/s of `peak` is the spec."""
    path = os.path.join(ROOT, "profiles", "r05_hbm_stream_probe.json")
    try:
        rows = json.load(open(path))["rows"]
    except (OSError, ValueError, KeyError):
        return None
    best = {}
    for r in rows:
        kind = ("read_only" if r["variant"].startswith("read,") or r["variant"].startswith("read through") else
                "copy" if r["variant"].startswith("copy") else "read_plus_11_32_write")
        best[kind] = max(best.get(kind, 0.0), float(r["GBps"]))
    return {"source": "profiles/r05_hbm_stream_probe.json (float4 / LDS-DMA streaming kernels over the same 400 MB, best of the grid sizes)",
            "GBps": best, "this_kernel_over_measured_copy": achieved_gbps / best["copy"] if best.get("copy") else None,
            "this_kernel_over_the_probe_with_its_own_mix": (achieved_gbps / best["read_plus_11_32_write"])
            if best.get("read_plus_11_32_write") else None}


def host_api(region, pts_dev):
    """The path the reference itself calls (mlfriends.pyx:1186-1211 from integrator.py:1776-1804): host numpy in,
    host bool mask out, P = 10^6; pageable source buffer.  Bound: 8 d bytes per proposal over PCIe."""
    import torch
    pts = pts_dev.cpu().numpy()
    region.inside(pts[:1000])
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        m = region.inside(pts)
        ts.append(time.perf_counter() - t0)
    dst = torch.empty_like(pts_dev)
    src = torch.from_numpy(pts)
    dst.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    h2d = (time.perf_counter() - t0) / 3
    med = float(np.median(ts))
    return {"call": "MLFriends.inside(host numpy (P, d) float64) -> host bool mask, pageable memory",
            "ms_per_batch": med * 1e3, "proposals_per_s": len(pts) / med, "accept_fraction": float(m.mean()),
            "h2d_GBps_same_buffer": pts.nbytes / h2d / 1e9, "pcie_bound_proposals_per_s": len(pts) / h2d,
            "fraction_of_pcie_bound": h2d / med}


def cpu_baseline(region, pts_host, u):
    """The CPU oracle (single-threaded C restatement of the reference's Cython loops,
    oracle/mlfriends_oracle.c) on a bounded sample of the same proposals, on this box's host."""
    from oracle import oracle as orc
    layer = region.transformLayer
    sample = pts_host
    t0 = time.perf_counter()
    mask = orc.region_inside(sample, region.unormed, layer.ctr, layer.T, region.ellipsoid_center,
                             region.ellipsoid_invcov, region.enlarge, region.maxradiussq)
    dt = time.perf_counter() - t0
    rs = np.random.RandomState(3)
    masks = orc.draw_bootstrap_masks(rs, N_LIVE, 3)
    t0 = time.perf_counter()
    orc.maxradiussq_bootstrap(region.unormed, masks)
    boot_ms = (time.perf_counter() - t0) * 1e3 / 3 * NBOOT
    parallel = cpu_baseline_parallel(region, sample, min(32, os.cpu_count() or 1))
    # how the port's speed relates to the real Cython path: timed side by side in the dev container by
    # tests/golden/make_golden.py, a fixture, not a measurement of this run
    cython_over_port = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "META.json")) as fh:
            tm = json.load(fh).get("timing", {})
        cython_over_port = {k: round(v["cython_over_port"], 3) for k, v in tm.items()
                            if isinstance(v, dict) and "cython_over_port" in v}
        cython_over_port["note"] = ("time of the reference's Cython build / time of this port, same inputs, one core of the "
                                    "dev container (tests/golden/META.json): > 1 = the port is faster, the baseline conservative")
    except (OSError, ValueError):
        pass
    return mask, dict(value=len(sample) / dt, unit="proposals/s", cores=1, kind="port", parallel=parallel,
                      cython_over_port=cython_over_port,
                      sample="first %d proposals of the timed batch (%.1f s); oracle/mlfriends_oracle.c, gcc -O3 "
                             "-ffp-contract=off, 1 thread" % (len(sample), dt),
                      bootstrap30_ms=boot_ms, host_cores_available=os.cpu_count())


_WORKER = """
import sys, time, numpy as np
sys.path.insert(0, %r)
from oracle import oracle as orc
z = np.load(sys.argv[1])
orc.lib()
t0 = time.perf_counter()
orc.region_inside(z['pts'], z['unormed'], z['ctr'], z['T'], z['ectr'], z['einv'], float(z['enlarge']), float(z['r2']))
print(time.perf_counter() - t0)
"""


def cpu_baseline_parallel(region, sample, nproc):
    """The same oracle pass in `nproc` independent processes at once (the reference's `mpiexec -np k`
    mode, docs/performance.rst:355-371, has no shared state in this path): host throughput when all
    of them run concurrently, each on the same sample."""
    import tempfile
    if nproc <= 1:
        return None
    layer = region.transformLayer
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sample.npz")
        np.savez(path, pts=sample, unormed=region.unormed, ctr=layer.ctr, T=layer.T, ectr=region.ellipsoid_center,
                 einv=region.ellipsoid_invcov, enlarge=region.enlarge, r2=region.maxradiussq)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", _WORKER % ROOT, path], stdout=subprocess.PIPE, text=True)
                 for _ in range(nproc)]
        inner = []
        for p in procs:
            out, _ = p.communicate(timeout=300)
            if p.returncode == 0:
                inner.append(float(out.strip().splitlines()[-1]))
        wall = time.perf_counter() - t0
    if len(inner) != nproc:
        return None
    return dict(cores=nproc, value=nproc * len(sample) / max(inner), unit="proposals/s",
                sample="%d processes, each the %d-proposal sample; slowest pass %.1f s, wall incl. start-up %.1f s"
                       % (nproc, len(sample), max(inner), wall))


def imported_traffic():
    """HBM bytes per kernel launch from the PMC counters.  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE need passes of their own (they
    cannot be collected inside this run): the figures of the last profiled tree are IMPORTED from profiles/pmc_scan_traffic.json
    (scripts/collect_pmc.py) and carry the hash of the kernel sources they were measured on; if the library built from THIS tree
    has another hash the figures are marked stale (a kernel change must not ship with old traffic figures)."""
    pmc_file = os.path.join(ROOT, "profiles", "pmc_scan_traffic.json")
    if not os.path.exists(pmc_file):
        return None
    pmc = json.load(open(pmc_file))
    try:
        from ultranest_amd.csrc import build as _build
        now = _build.source_hash()
    except Exception:       # noqa: BLE001
        now = None
    then = pmc.get("source_hash")
    pmc["fresh"] = bool(now and then and now == then)
    pmc["source_hash_now"] = now
    pmc["status"] = ("measured on this tree's kernel sources" if pmc["fresh"] else
                     "STALE: measured on kernel sources with hash %s, this tree has %s" % (then, now))
    return pmc


def step_roofline(handle, stats, kdim, ntiles32, filter_launches, launch_ms_list, steps, fused_first, step_ms, launch_pass_ms,
                  alg_bytes, pmc, hbm, scan_flops):
    """The roofline object of the JSON line: every matrix-kernel launch of a step with its time (hipEvents of the launch-event
    pass), executed matrix instructions, bytes and both fractions; the kernel that takes most of the step named at the top."""
    ks = kdim // 16
    nlaunch, ms_kernels = filter_launches
    per_step = max(1, round(nlaunch / max(steps, 1)))
    ngroups1 = (NPROPOSALS + 31) // 32
    by_phase = [float(np.mean(launch_ms_list[i::per_step])) for i in range(per_step)] if len(launch_ms_list) else []
    cut, cut2 = stats["range_cuts"]       # the tile ranges as the library cut them for this batch
    slot = SET_BYTES_PER_SLOT(ks)
    prep_per_group = PREP_MFMA_PER_GROUP_SAME_FORM if stats.get("same_quadratic_form") else PREP_MFMA_PER_GROUP
    per_kernel_pmc = (pmc or {}).get("per_kernel_all") or {}

    def counter_bytes(prefix):
        for k, v in per_kernel_pmc.items():
            if k.startswith(prefix):
                return v
        return None

    launches = []
    # k_prep_sweep<DP, waves per workgroup>: 4 (two workgroups per CU) up to d = 50 by default, else 8 ("fused_waves")
    # ..., same quadratic form>: the ellipsoid form read off the whitening chain (this workload: AffineLayer, one cluster)
    fused_name = "k_prep_sweep<%d, %d%s>" % (NDIM, 4 if (handle.get_option("fused_waves") == 4 and NDIM <= 50) else 8,
                                            ", true" if stats.get("same_quadratic_form") else "")
    if per_step in (3, 4) and cut and len(by_phase) == per_step:
        # min-only sweep (mlf_sweepmin.hip): first range over every group, the later ones over the groups left after the
        # compaction in front of them, then the uncertain proposals (sets of 4 groups) once more over all tiles
        nranges = per_step - 1
        assert (cut2 > 0) == (nranges == 3), (per_step, stats)
        tiles = [cut, cut2 - cut, ntiles32 - cut2] if cut2 else [cut, ntiles32 - cut]
        groups = [ngroups1, stats["second_range_groups"], stats.get("third_range_groups", 0)][:nranges]
        nunc = stats["uncertain_queries"]
        usets = -(-(-(-nunc // 32)) // 4)
        for i in range(nranges):
            first = i == 0
            name = fused_name if (first and fused_first) else "k_sweep_min<4, 4, 2>"
            sweep_mfma = groups[i] * tiles[i] * ks
            prep_mfma = ngroups1 * prep_per_group if (first and fused_first) else 0
            set_in = 0 if (first and fused_first) else groups[i] * 32 * slot
            set_out = (groups[i + 1] * 32 * slot) if i + 1 < nranges else nunc * slot
            alg_in = NPROPOSALS * (8 * NDIM + 1) if (first and fused_first) else 0
            launches.append({
                "kernel": name,
                "what": ("per-proposal stage (ellipsoid bound + whitening on the matrix cores, split binary16) AND the first live-point range "
                         "in one launch; running minima only; compacts the proposals without a certain hit, with their minima" if (first and fused_first)
                         else "range %d of %d over the proposals left; running minima only; compacts %s" %
                         (i + 1, nranges, "the proposals whose minimum ended in the band" if i + 1 == nranges else "again, with the minima")),
                "ms": by_phase[i], "tiles": tiles[i], "groups_of_32_proposals": groups[i],
                "executed_mfma": prep_mfma + sweep_mfma, "executed_mfma_whitening": prep_mfma, "executed_mfma_sweep": sweep_mfma,
                "algorithmic_bytes_in": alg_in, "compacted_set_bytes_in": set_in, "compacted_set_bytes_out": set_out,
                "bytes": alg_in + set_in + set_out, "counter_bytes": counter_bytes(name.split("<")[0])})
        launches.append({
            "kernel": "k_uncertain<4, 4>",
            "what": "the proposals whose minimum ended in the band: all tiles again with their band pairs listed, binary64 whitening, "
                    "the pairs in the reference's arithmetic; trailing workgroups decide the ellipsoid band",
            "ms": by_phase[nranges], "tiles": ntiles32, "groups_of_32_proposals": usets * 4,
            "executed_mfma": usets * 4 * ntiles32 * ks, "executed_mfma_whitening": 0, "executed_mfma_sweep": usets * 4 * ntiles32 * ks,
            "algorithmic_bytes_in": 0, "compacted_set_bytes_in": nunc * slot, "compacted_set_bytes_out": 0,
            "bytes": nunc * (slot + 8 * NDIM), "counter_bytes": counter_bytes("k_uncertain")})
    else:
        tiles = [ntiles32 * (i + 1) // per_step - ntiles32 * i // per_step for i in range(per_step)]
        if per_step == 2 and cut:
            tiles = [cut, ntiles32 - cut]
        groups = [ngroups1] + [stats["second_range_groups"]] * (per_step - 1)
        for i in range(min(per_step, len(by_phase))):
            launches.append({"kernel": "k_sweep<4, %d, ...>" % (4 if i == 0 else 2), "what": "range %d of %d, per-tile band test" % (i + 1, per_step),
                             "ms": by_phase[i], "tiles": tiles[i], "groups_of_32_proposals": groups[i], "executed_mfma": groups[i] * tiles[i] * ks,
                             "executed_mfma_whitening": 0, "executed_mfma_sweep": groups[i] * tiles[i] * ks, "algorithmic_bytes_in": 0,
                             "compacted_set_bytes_in": 0, "compacted_set_bytes_out": 0, "bytes": 0, "counter_bytes": None})
    for e in launches:
        sec = e["ms"] * 1e-3
        e["mfma_TFLOPs"] = e["executed_mfma"] * MFMA_F16_FLOPS / sec / 1e12
        e["mfma_frac_of_2500"] = e["mfma_TFLOPs"] / F16_MFMA_PEAK_TFLOPS
        e["GBps"] = e["bytes"] / sec / 1e9
        e["hbm_frac_of_8000"] = e["GBps"] / HBM_PEAK_GBPS
        e["share_of_step_time"] = e["ms"] / step_ms
    # the kernel that takes most of the step (summed over its launches)
    by_kernel = {}
    for e in launches:
        by_kernel[e["kernel"]] = by_kernel.get(e["kernel"], 0.0) + e["ms"]
    dominant = max(by_kernel, key=by_kernel.get) if by_kernel else None
    dom = [e for e in launches if e["kernel"] == dominant]
    dom_ms = float(np.mean([e["ms"] for e in dom])) if dom else None
    dom_bytes = float(np.mean([e["bytes"] for e in dom])) if dom else 0.0
    dom_mfma = float(np.mean([e["executed_mfma"] for e in dom])) if dom else 0.0
    hbm_bound = bool(dominant and dominant.startswith("k_prep_sweep"))
    ach_hbm = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else None
    ach_mfma = dom_mfma * MFMA_F16_FLOPS / (dom_ms * 1e-3) / 1e12 if dom_ms else None
    sweeps = [e for e in launches if e["kernel"].startswith("k_sweep_min")]
    sweep_entry = None
    if sweeps:
        sm_ms = float(np.mean([e["ms"] for e in sweeps]))
        sm_flops = float(np.mean([e["executed_mfma"] for e in sweeps])) * MFMA_F16_FLOPS
        sweep_entry = {"kernel": "k_sweep_min<4, 4, 2> (average over its %d launches of a step)" % len(sweeps), "bound": "mfma",
                       "achieved": sm_flops / (sm_ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": sm_flops / (sm_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, "ms_per_launch": sm_ms,
                       "frac_of_what_the_instruction_sustains_alone_on_random_operands": sm_flops / (sm_ms * 1e-3) / 1e12 / PROBE_F16_MFMA_TFLOPS,
                       "share_of_step_time": sum(e["ms"] for e in sweeps) / step_ms}
    mfma_step = float(sum(e["executed_mfma"] for e in launches))
    step_counter = (pmc or {}).get("hbm_bytes_per_step_all_kernels")
    fp64 = None
    if scan_flops:
        eq = scan_flops / (step_ms * 1e-3) / 1e12
        fp64 = {"reference_loop_flops_of_this_batch": scan_flops, "flop_per_proposal": scan_flops / NPROPOSALS,
                "equivalent_TFLOPs_over_the_step": eq, "peak": FP64_VALU_PEAK_TFLOPS, "ratio_to_that_peak": eq / FP64_VALU_PEAK_TFLOPS,
                "why_above_1": "SURVEY.md 8d prices the path as 3 d binary64 flops per (proposal, live point visited) on the vector ALUs "
                               "(39.3 T non-fused op/s).  That roofline no longer binds: 99.99 % of the pair decisions are made by a PROVEN "
                               "binary16 / binary32 matrix-core bound on the pair distance (DESIGN 4b, tests/test_filter_bounds.py); only the "
                               "pairs inside the bound's band (uncertain_pairs) are evaluated in the reference's binary64 arithmetic, and the "
                               "masks are asserted equal to the exact binary64 scan's in this run.  No work is skipped that the answer depends "
                               "on; the binding roofs are HBM (first launch) and the f16 matrix pipe (the sweeps)"}
    return {
        "kernel": dominant, "dominant_by": "time: %.4f of the %.4f ms step" % (by_kernel.get(dominant, 0.0), step_ms),
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": ach_hbm if hbm_bound else ach_mfma,
        "peak": HBM_PEAK_GBPS if hbm_bound else F16_MFMA_PEAK_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": (ach_hbm / HBM_PEAK_GBPS if hbm_bound else ach_mfma / F16_MFMA_PEAK_TFLOPS) if dom_ms else None,
        "traffic": dom[0]["counter_bytes"] if (dom and pmc and pmc.get("fresh")) else None,
        "traffic_status": (pmc or {}).get("status", "no PMC file"),
        "traffic_imported_value": dom[0]["counter_bytes"] if dom else None,
        "ms_per_launch": dom_ms,
        "bytes_per_launch": dom_bytes,
        "bytes_what": "proposals in (8 d + 1 B each) + the compacted set out (%d B per proposal without a certain hit)" % slot if hbm_bound else "operand sets in / out",
        "other_roof": ({"bound": "mfma", "achieved": ach_mfma, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_mfma / F16_MFMA_PEAK_TFLOPS,
                        "executed_mfma_per_launch": dom_mfma} if hbm_bound else
                       {"bound": "hbm", "achieved": ach_hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (ach_hbm or 0.0) / HBM_PEAK_GBPS}),
        "hbm_peak_measured": hbm_probe_ceiling(ach_hbm) if ach_hbm else None,
        "mfma_peak_at_measured_clock": {
            "instruction_alone_random_operands_TFLOPs": PROBE_F16_MFMA_TFLOPS,
            "what": "the part is power-limited under v_mfma_f32_32x32x16_f16: the instruction alone (4 independent chains, operands in "
                    "registers, matrix pipe 98 % occupied) runs at 2.2 GHz = 2.23-2.33 PFLOP/s on ZERO operands and at 1.50-1.60 GHz = "
                    "1.48-1.63 PFLOP/s on random binary16 operands (scripts/probes/sweep_src_probe.hip, profiles/r04_sweep_src_probe.json)"},
        "how_measured": "ms: hipEvent pairs around every matrix-kernel launch on the library's stream, in a pass of its own over the same "
                        "%d steps (wall %.4f ms per step with the events, %.4f without: the headline loop carries none); executed_mfma: "
                        "group counts read back from the device x tiles x K / 16 (= SQ_INSTS_MFMA of profiles/); bytes: see bytes_what; "
                        "counter_bytes: (2 FETCH_SIZE + WRITE_SIZE) x 1024 of the rocprofv3 --pmc passes (profiles/), imported" % (steps, launch_pass_ms, step_ms),
        "launches": launches,
        "launches_per_step": per_step,
        "k_sweep_min": sweep_entry,
        "step": {"ms": step_ms, "sum_of_the_matrix_launches_ms": float(sum(e["ms"] for e in launches)),
                 "executed_mfma": mfma_step, "mfma_TFLOPs": mfma_step * MFMA_F16_FLOPS / (step_ms * 1e-3) / 1e12,
                 "mfma_frac_of_2500": mfma_step * MFMA_F16_FLOPS / (step_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS,
                 "algorithmic_bytes": alg_bytes, "algorithmic_GBps": hbm["achieved_GBps"], "algorithmic_hbm_frac": hbm["frac"],
                 "counter_bytes": step_counter, "counter_bytes_status": (pmc or {}).get("status", "no PMC file"),
                 "counter_GBps": (step_counter / (step_ms * 1e-3) / 1e9) if step_counter else None,
                 "counter_bytes_over_algorithmic": (step_counter / alg_bytes) if step_counter else None},
        "fp64_valu_roofline_of_survey_8d": fp64,
        "range_cuts_tiles": [cut, cut2], "tiles": ntiles32, "executed_k_columns": kdim,
        "first_range_pct": handle.get_option("filter_first_range_pct"), "second_range_pct": handle.get_option("filter_second_range_pct"),
        "mask_operand_order": "nearest to the centre first" if handle.get_option("filter_order") else "storage order",
        "uncertain_queries": stats.get("uncertain_queries"), "uncertain_pairs": stats.get("uncertain_pairs"),
        "hbm": hbm}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run ourselves."""
    import socket
    import torch
    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    if ndev < args.gpus:      # fewer devices than ranks (a 1-GPU box): ranks share devices, collectives over gloo
        env["MLF_BENCH_BACKEND"] = "gloo"
        env["MLF_BENCH_NDEV"] = str(max(ndev, 1))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=150000, help="proposals timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="which leg is the headline `value`: weak = P proposals per rank and step (default), strong = ONE "
                         "batch of P proposals per step sharded over the ranks; the other leg is reported beside it")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline path (no single-sweep / filter-off / host-API comparison passes): used for "
                         "the rocprofv3 runs, so that the per-kernel averages of the summary are those of the timed steps")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch one process per GPU (torch.distributed.run)"
    # hooks for boxes with fewer devices than ranks: MLF_BENCH_DEVICE pins every rank to one device, MLF_BENCH_NDEV
    # wraps the local rank around the devices present, MLF_BENCH_BACKEND replaces "nccl"
    if "MLF_BENCH_DEVICE" in os.environ:
        device_index = int(os.environ["MLF_BENCH_DEVICE"])
    else:
        device_index = local_rank % int(os.environ.get("MLF_BENCH_NDEV", str(max(world, 1))))
    backend = os.environ.get("MLF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    from ultranest_amd import _lib, kernels
    _lib.set_device(device_index)
    group = None
    # under torch.distributed.run the process group is RCCL, also for a single rank (so that the 1-GPU
    # box exercises the same init / barrier / all-reduce calls the N > 1 runs make)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        # device work of THIS rank first (the untimed settle / warm-up steps are still queued when the host gets here; ranks
        # that share a device -- the 1-GPU boxes of the pool -- would otherwise start their clocks while another rank's
        # queue is still draining), then the ranks meet, then the device once more
        torch.cuda.synchronize()
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    u, region = build_region(group)
    handle = region._dev.sync(region, True)          # device-resident region state
    # THREE distinct batches, visited in turn by every timed loop: one 400 MB batch filtered again and again could sit in the
    # 256 MB Infinity Cache between steps (MI355X_MICROARCH.md: FETCH_SIZE counts MALL hits too) -- 1.2 GB cannot
    batches = [proposals_in_ellipsoid(region, NPROPOSALS, 1000 + rank + 7919 * k, dev) for k in range(NBATCHES)]
    masks = [torch.empty(NPROPOSALS, dtype=torch.uint8, device=dev) for _ in range(NBATCHES)]
    pts, mask = batches[0], masks[0]
    stream = torch.cuda.current_stream().cuda_stream

    from ultranest_amd import _lib as lib_mod

    def run_steps(nsteps, stage_events, settle=0):
        """nsteps passes between two barriers, batch (step mod 3); returns (max-over-ranks seconds, this rank's seconds).
        settle: untimed steps in front of the W warm-up steps (CLOCK_SETTLE_STEPS for the headline: the device's clock
        management needs tens of milliseconds of load before the chip runs at its sustained clock; the first 10 ms
        after an idle period -- such as the host-side region build in front of this loop -- run ~8 % slower).
        stage_events: also record the per-stage hipEvents (only outside the headline loop: six extra events per pass
        cost ~7 % of a 0.6 ms step)."""
        call = handle.inside_dev_timed if stage_events else handle.inside_dev
        for i in range(settle + args.warmup):
            handle.inside_dev(batches[i % NBATCHES].data_ptr(), NPROPOSALS, masks[i % NBATCHES].data_ptr(), stream)
        handle.timing_collect()
        handle.timing_filter_launches()
        barrier()
        t0 = time.perf_counter()
        for i in range(nsteps):
            call(batches[i % NBATCHES].data_ptr(), NPROPOSALS, masks[i % NBATCHES].data_ptr(), stream)
        barrier()
        mine = time.perf_counter() - t0
        dt = mine
        if use_dist:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        run_steps.launch_ms = handle.timing_filter_launch_ms()
        run_steps.filter_launches = handle.timing_filter_launches()
        return dt, mine

    def batch0_once():
        handle.inside_dev(pts.data_ptr(), NPROPOSALS, mask.data_ptr(), stream)
        torch.cuda.synchronize()

    lib_mod.set_option("time_filter_launches", 0)
    exact_elapsed = ncalls_x = ms_scan_x = scan_flops = ell_pass = None
    nsteps_x = max(3, args.steps // 4)
    masks_exact = None
    if not args.headline_only:
        # ---- the reference answer first: the exact FP64 scan kernel alone (MFMA pre-filter switched off) on the same
        # batches.  Its masks are what the timed path is compared with below; its time is `roofline_exact_scan`.
        lib_mod.set_option("filter", 0)
        exact_elapsed, _ = run_steps(max(nsteps_x, NBATCHES), True)
        ncalls_x, ms_prep_x, ms_scan_x, ms_rest_x = handle.timing_collect()
        lib_mod.set_option("filter", 1)
        masks_exact = [m.clone() for m in masks]
    # ---- the headline: W warm-up steps, then exactly K steps; NO events inside the loop (the event pairs around the
    # matrix launches cost 7 %; the per-launch times come from the separate pass below) -------------------------------------
    elapsed, elapsed_mine = run_steps(args.steps, False, settle=CLOCK_SETTLE_STEPS)
    for i in range(NBATCHES):      # every batch's mask is the timed path's (a short loop may not have reached all three)
        handle.inside_dev(batches[i].data_ptr(), NPROPOSALS, masks[i].data_ptr(), stream)
    batch0_once()
    stats = handle.debug_stats()
    accept = float(mask.float().mean().item())
    filter_on, kdim, ntiles32 = handle.filter_info(NPROPOSALS)
    masks_filter = [m.clone() for m in masks]
    mask_filter = masks_filter[0]
    if masks_exact is not None:
        for me, mf in zip(masks_exact, masks_filter):
            assert bool((me == mf).all().item()), "filter and exact scan disagree"
    # ---- per-launch times of the matrix kernels: the same K steps once more with an event pair around every such launch ----
    lib_mod.set_option("time_filter_launches", 1)
    launch_pass_s, _ = run_steps(args.steps, False, settle=args.warmup)
    filter_launches = run_steps.filter_launches
    launch_ms_list = run_steps.launch_ms
    lib_mod.set_option("time_filter_launches", 0)
    # ---- per-stage breakdown (separate pass with stage events) ---------------------------------------------
    nsteps_b = max(3, args.steps // 2)
    stage_pass_s, _ = run_steps(nsteps_b, True)
    ncalls, ms_prep, ms_scan, ms_rest = handle.timing_collect()
    batch0_once()
    assert bool((mask == mask_filter).all().item())
    # the per-proposal stage rides in the first sweep launch by default (fused_first_range): its own kernel (k_prep4) is
    # timed in one more pass with the option off, for `roofline_prep` and `kernel_ms.unfused`
    fused_first = bool(handle.get_option("fused_first_range")) and NDIM % 2 == 0 and NDIM <= 56
    unfused = None
    if fused_first and not args.headline_only:
        lib_mod.set_option("fused_first_range", 0)
        unfused_pass_s, _ = run_steps(nsteps_b, True)
        ncalls_u, ms_prep_u, ms_scan_u, ms_rest_u = handle.timing_collect()
        batch0_once()
        lib_mod.set_option("fused_first_range", 1)
        assert bool((mask == mask_filter).all().item()), "the per-proposal stage inside and in front of the first sweep launch disagree"
        unfused = {"prep": ms_prep_u / max(ncalls_u, 1), "scan": ms_scan_u / max(ncalls_u, 1), "tail": ms_rest_u / max(ncalls_u, 1),
                   "wall_ms_per_step": unfused_pass_s / nsteps_b * 1e3}

    # single-sweep variant of the pre-filter (no compaction between live-point ranges), for the record
    ms_scan_single = None
    hostapi = None
    if not args.headline_only:
        lib_mod.set_option("filter_phases", 0)
        run_steps(nsteps_x, True)
        ncalls_single, _, ms_scan_single, _ = handle.timing_collect()
        ms_scan_single /= max(ncalls_single, 1)
        batch0_once()
        lib_mod.set_option("filter_phases", 1)
        assert bool((mask == mask_filter).all().item()), "single sweep and phased sweep disagree"
        # ... and the FP64 per-proposal stage (k_prep3) in front of the filter instead of the bounded FP32 one
        lib_mod.set_option("prep_bounded", 0)
        run_steps(nsteps_x, True)
        ncalls_p3, ms_prep_p3, _, ms_rest_p3 = handle.timing_collect()
        batch0_once()
        lib_mod.set_option("prep_bounded", 1)
        assert bool((mask == mask_filter).all().item()), "bounded and exact per-proposal stage disagree"

        # algorithmic work of the neighbour scan on this batch: the reference's loop stops at the first
        # hit, so a proposal costs 3*d flops per live point visited = (first index + 1), or N if none
        idx = torch.empty(NPROPOSALS, dtype=torch.int64, device=dev)
        handle.first_index_dev(pts.data_ptr(), NPROPOSALS, idx.data_ptr(), stream)
        torch.cuda.synchronize()
        visited = torch.where(idx >= 0, idx + 1, torch.full_like(idx, N_LIVE))
        visited = torch.where(idx == -2, torch.zeros_like(idx), visited)
        scan_flops = 3.0 * NDIM * float(visited.sum().item())
        ell_pass = float((idx != -2).float().mean().item())
        assert bool(((idx >= 0) == (mask != 0)).all().item()), "index and mask pipelines disagree"
        del idx, visited
        if rank == 0:
            hostapi = host_api(region, pts)

    # ---- strong scaling: ONE batch of P proposals per step, rows sharded over the ranks (distributed.shard_bounds; no
    # collective in the data path, integrator.py:1916-1928 gathers only accepted rows), and the 30-round bootstrap of
    # the rebuild sharded over the ranks INCLUDING its all-reduce (integrator.py:375-415) -------------------------
    from ultranest_amd import distributed

    def maxed(x):
        if not use_dist:
            return float(x)
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    lo, hi = distributed.shard_bounds(NPROPOSALS, rank, world)
    ranks_agree = None
    if world > 1:
        full = proposals_in_ellipsoid(region, NPROPOSALS, 1000, dev)     # the SAME batch on every rank; each takes its rows
        shard = full[lo:hi].contiguous()
        # every rank also filters the SAME first 65536 rows of that batch: a digest of the mask must be the same number on
        # every rank (MIN == MAX over the ranks) -- a rank whose region, library or device disagrees fails the run loudly
        nshared = 65536
        shared_mask = torch.empty(nshared, dtype=torch.uint8, device=dev)
        handle.inside_dev(full.data_ptr(), nshared, shared_mask.data_ptr(), stream)
        torch.cuda.synchronize()
        digest = int((shared_mask.long() * (torch.arange(nshared, device=dev, dtype=torch.int64) % 1000003 + 1)).sum().item())
        import torch.distributed as dist
        dg = torch.tensor([digest, -digest], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(dg, op=dist.ReduceOp.MAX)
        ranks_agree = bool(int(dg[0].item()) == -int(dg[1].item()))
        assert ranks_agree, "rank %d: the ranks' masks on the shared rows differ (digest %d, max %d, min %d)" % (
            rank, digest, int(dg[0].item()), -int(dg[1].item()))
        del full, shared_mask
    else:
        shard = pts
    smask = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
    for _ in range(CLOCK_SETTLE_STEPS + args.warmup):
        handle.inside_dev(shard.data_ptr(), hi - lo, smask.data_ptr(), stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        handle.inside_dev(shard.data_ptr(), hi - lo, smask.data_ptr(), stream)
    barrier()
    strong_mine = time.perf_counter() - t0
    strong_elapsed = maxed(strong_mine)
    strong_accept = float(smask.float().mean().item())
    # sharded bootstrap: masks drawn once, broadcast, 30 rounds over the ranks, ONE all-reduce(MAX) of 3 doubles
    boot_reps = max(3, min(10, args.steps))
    r2_keep, f_keep = region.maxradiussq, region.enlarge
    rs_b = np.random.RandomState(5)
    distributed.update_region_bootstrap(region, NBOOT, minvol=0., group=group, rng=rs_b)
    barrier()
    t0 = time.perf_counter()
    for _ in range(boot_reps):
        r_b, f_b = distributed.update_region_bootstrap(region, NBOOT, minvol=0., group=group, rng=rs_b)
    barrier()
    boot_mine = (time.perf_counter() - t0) / boot_reps
    boot_elapsed = maxed(boot_mine)
    region.maxradiussq, region.enlarge = r2_keep, f_keep     # the timed batches (and the CPU baseline below) belong to THIS region
    ranks_seen = 1
    collective_library = None
    if use_dist:
        import torch.distributed as dist
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        if backend == "nccl":
            try:
                collective_library = "RCCL %s (torch.cuda.nccl.version(); backend 'nccl' IS RCCL on ROCm)" % ".".join(
                    str(x) for x in torch.cuda.nccl.version())
            except Exception as e:       # noqa: BLE001 -- the version string is a report field, not a requirement
                collective_library = "RCCL (version unavailable: %s)" % e
        else:
            collective_library = "gloo (ranks share this box's device(s): rehearsal, not RCCL)"
    strong = {
        "what": "ONE batch of %d proposals per step, rows sharded over %d rank(s) by shard_bounds (region replicated, no "
                "collective in the data path); value = P x steps / max-over-ranks seconds between two barriers" % (NPROPOSALS, world),
        "value": NPROPOSALS * args.steps / strong_elapsed, "unit": "proposals/s", "ms_per_step": strong_elapsed / args.steps * 1e3,
        "rows_this_rank": hi - lo, "accept_fraction_this_rank": strong_accept,
        "bootstrap30_sharded_ms": boot_elapsed * 1e3,
        "bootstrap30_sharded_what": "distributed.update_region_bootstrap: 30 selection masks drawn on every rank, rank 0's "
                                    "broadcast, ceil(30 / ranks) rounds per rank on the device, ONE all-reduce(MAX) of (r2, f, "
                                    "error flag), INCLUDED in the time; max over ranks",
        "bootstrap_result": [r_b, f_b],
        "collective_backend": (backend if use_dist else None), "collective_library": collective_library,
        "ranks_seen_by_allreduce": ranks_seen,
        "ranks_agree_on_the_shared_rows": ranks_agree,
    }

    first_ms, rebuild_ms, rebuild_all = time_rebuild(u, group)
    _, rebuild_dev_ms, rebuild_dev_all = time_rebuild(u, group, device_resident=True) if world == 1 else (None, None, None)

    per_rank = [elapsed_mine]
    per_rank_strong = [(strong_mine, boot_mine)]
    if use_dist:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, elapsed_mine)
        per_rank = gathered
        gathered = [None] * world
        dist.all_gather_object(gathered, (strong_mine, boot_mine))
        per_rank_strong = gathered
    if rank != 0:
        if use_dist:
            import torch.distributed as dist
            dist.barrier()          # rank 0 reaches this after assembling the JSON line
            dist.destroy_process_group()
        return

    scan_ms = ms_scan / max(ncalls, 1)
    prep_ms = ms_prep / max(ncalls, 1)
    rest_ms = ms_rest / max(ncalls, 1)
    value = NPROPOSALS * world * args.steps / elapsed
    strong["per_rank_ms_per_step"] = [a / args.steps * 1e3 for a, _ in per_rank_strong]
    strong["per_rank_bootstrap30_ms"] = [b * 1e3 for _, b in per_rank_strong]
    weak = {"what": "%d proposals per rank and step (the per-GPU work is fixed as ranks are added)" % NPROPOSALS,
            "value": value, "unit": "proposals/s", "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank]}
    headline = strong if args.scaling == "strong" else weak
    step_ms = elapsed / args.steps * 1e3
    alg_bytes = NPROPOSALS * (8 * NDIM + 1) + 8 * N_LIVE * NDIM + 2 * 8 * NDIM * NDIM
    traffic_file = imported_traffic()
    exact_roof = scan_x_ms = None
    if not args.headline_only:
        scan_x_ms = ms_scan_x / max(ncalls_x, 1)
        exact_tflops = scan_flops / (scan_x_ms * 1e-3) / 1e12
        exact_roof = {
            "kernel": "k_scan<50> (exact FP64 neighbour scan; the whole scan when the pre-filter is off, "
                      "the re-check arithmetic when it is on)",
            "bound": "valu_fp64", "achieved": exact_tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": exact_tflops / FP64_VALU_PEAK_TFLOPS, "ms_per_launch": scan_x_ms,
            "algorithmic_flops_per_launch": scan_flops,
            "measured_valu_probe_tflops": kernels.bench_fp64_valu(),
            "proposals_per_s_filter_off": NPROPOSALS * world * max(nsteps_x, NBATCHES) / exact_elapsed,
            "note": "bit-exactness forbids FMA/MFMA in the distance itself: peak = non-fused FP64 vector issue rate"}
    hbm = {"algorithmic_bytes_per_step": alg_bytes, "achieved_GBps": alg_bytes / (step_ms * 1e-3) / 1e9,
           "peak_GBps": HBM_PEAK_GBPS, "frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "over": "ms_per_step of the timed loop (this rank's batch)"}
    if filter_on:
        roofline = step_roofline(handle, stats, kdim, ntiles32, filter_launches, launch_ms_list, args.steps, fused_first, step_ms,
                                 launch_pass_s / args.steps * 1e3, alg_bytes, traffic_file, hbm, scan_flops)
    else:
        roofline = dict(exact_roof or {})
        roofline["traffic"] = None
        roofline["hbm"] = hbm
    # per-proposal stage: row in (8 d), binary16 operand + thresholds + route / slot / best words out
    prep4_ms = (unfused["prep"] if unfused else None) if fused_first else prep_ms      # k_prep4 in its own launch
    prep_bytes = NPROPOSALS * (8 * NDIM + 2 * kdim + 8 + 1 + 4 + 4 + 1)
    prep_mfma_flops = NPROPOSALS / 32 * PREP_MFMA_PER_GROUP * MFMA_F16_FLOPS   # k_prep4 in its own launch runs both chains
    out = {
        "metric": "proposal-points filtered/sec (MLFriends.inside) + region-rebuild ms, N=4000 d=50",
        "value": headline["value"], "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": headline["ms_per_step"], "higher_is_better": True, "scaling": args.scaling,
        "clock_settle_steps": CLOCK_SETTLE_STEPS,
        "weak_scaling": weak, "strong_scaling": strong,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C5: N=4000 live points, d=50, P=1e6 proposals/step/GPU drawn uniformly inside the "
                               "wrapping ellipsoid (set E), AffineLayer, 30 bootstraps; one step = MLFriends.inside "
                               "on a batch resident in HBM; the timed loop visits %d distinct batches in turn" % NBATCHES,
                   "n_live": N_LIVE, "d": NDIM, "proposals_per_step_per_gpu": NPROPOSALS, "bootstraps": NBOOT,
                   "distinct_batches_in_the_timed_loop": NBATCHES,
                   "parallelism": "proposal rows sharded over " + str(world) + " GPU(s), region replicated",
                   "mfma_prefilter": bool(filter_on),
                   "arithmetic": "inputs, thresholds and every decision that depends on the reference's rounding: binary64 "
                                 "(non-fused); deciding bounds for the other 99.99 %% of the pairs: binary16 operands / "
                                 "binary32 accumulate on the matrix cores, bounded whitening with split binary16 operands; masks bit-identical "
                                 "to the exact FP64 scan (asserted in this run, all %d batches)" % NBATCHES},
        "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
        "rebuild_ms": rebuild_ms, "rebuild_first_ms": first_ms, "rebuild_ms_each": rebuild_all,
        "rebuild_what": "steady-state rebuild through harness.RegionUpdater (LocalAffineLayer create_new incl. clustering + "
                        "subtract_nearby, region, 30 bootstrap rounds, wrapping ellipsoid, membership of the live points): "
                        "DEFAULT path -- every numpy / LAPACK call of the reference kept on the host, T and unormed bit-identical "
                        "to the reference's",
        "rebuild_device_resident_ms": rebuild_dev_ms, "rebuild_device_resident_ms_each": rebuild_dev_all,
        "rebuild_device_resident_what": "the same rebuild with RegionUpdater(device_resident=True) (ultranest_amd.device_rebuild, "
                                        "opt-in): live points uploaded once, whitening / covariances / clustering / bootstrap on "
                                        "the device, d x d LAPACK on the host; tolerance class 1e-10 on T, radius, enlargement "
                                        "instead of bit parity (tests/test_device_rebuild.py); N = 1 only",
        "accept_fraction": accept, "ellipsoid_pass_fraction": ell_pass,
        "kernel_ms": {"prep": prep_ms, "scan": scan_ms, "tail": rest_ms, "unfused": unfused, "scan_single_sweep": ms_scan_single,
                      "prep_fp64": (ms_prep_p3 / max(ncalls_p3, 1)) if not args.headline_only else None,
                      "keys": {"prep": ("nothing of its own: the per-proposal stage runs inside the first sweep launch (k_prep_sweep); the stage event pair brackets an empty interval"
                                        if fused_first else
                                        "per-proposal stage (k_prep4: bounded ellipsoid test + whitening on the matrix cores with split "
                                        "binary16 operands -> f16 operand; the ellipsoid band is decided by trailing workgroups of the k_uncertain launch)"),
                               "scan": ((("k_prep_sweep (per-proposal stage + first live-point range) + k_sweep_min over the later ranges" if fused_first else
                                          "k_sweep_min over the live-point ranges") +
                                         " + k_uncertain (the proposals whose minimum ended in the band, incl. their exact whitening)") if filter_on else "k_scan"),
                               "unfused": "the same three stages with fused_first_range = 0: k_prep4 in its own launch (prep), k_sweep_min over all ranges + k_uncertain (scan)",
                               "tail": "k_scan tail: what the filter could not take, routing, finalise",
                               "scan_single_sweep": "the scan stage as a single sweep over all live points (filter_phases = 0)",
                               "prep_fp64": "per-proposal stage with the FP64 kernel instead (k_prep3, prep_bounded = 0)"},
                      "measured_in": "a separate pass of %d steps with stage events (the headline loop carries no events at all)" % nsteps_b,
                      # the stages add up to the wall time of THAT pass, not of the headline loop: every stage boundary is a
                      # hipEventRecord between two dependent kernels, i.e. a bubble in the queue and a time stamp taken after
                      # the device has drained; the three figures below make the difference visible (+14 %)
                      "sum_of_the_stages": prep_ms + scan_ms + rest_ms,
                      "wall_ms_per_step_of_the_stage_event_pass": stage_pass_s / nsteps_b * 1e3,
                      "wall_ms_per_step_of_the_launch_event_pass": launch_pass_s / args.steps * 1e3,
                      "wall_ms_per_step_of_the_headline_loop": headline["ms_per_step"]},
        "batch_counters": stats,
        "roofline": roofline,
        "roofline_exact_scan": exact_roof,
        "roofline_prep": None if prep4_ms is None else {"kernel": "k_prep4<50> (split-binary16 v_mfma_f32_32x32x16_f16 + global_load_lds staging)", "bound": "hbm",
                          "unit": "GB/s", "achieved": prep_bytes / (prep4_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                          "frac": prep_bytes / (prep4_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                          "algorithmic_bytes_per_launch": prep_bytes,
                          "mfma_f16": {"executed_flops_per_launch": prep_mfma_flops,
                                       "achieved_TFLOPs": prep_mfma_flops / (prep4_ms * 1e-3) / 1e12,
                                       "peak_TFLOPs": F16_MFMA_PEAK_TFLOPS},
                          "peak_measured": hbm_probe_ceiling(prep_bytes / (prep4_ms * 1e-3) / 1e9),
                          "measured_with": "fused_first_range = 0 (k_prep4 in its own launch: a separate pass; by default the stage rides in k_prep_sweep, roofline.launches[0])" if fused_first else "the default routing",
                          "note": "the stage time includes the launch gap in front of the filter; the matrix work is 13 % "
                                  "of the SIMD time (profiles/); DESIGN.md section 4c"},
        "host_api": hostapi,
    }
    if world == 1 and not args.no_cpu:
        sample = pts[:args.cpu_sample].cpu().numpy()
        cmask, base = cpu_baseline(region, sample, u)
        gmask = mask_filter[:args.cpu_sample].cpu().numpy().astype(bool)
        base["gpu_mask_equals_cpu_mask_on_sample"] = bool(np.array_equal(cmask, gmask))
        base["speedup_1gpu_vs_1core"] = value / base["value"]
        out["cpu_baseline"] = base
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
