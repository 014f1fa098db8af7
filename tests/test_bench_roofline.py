"""bench.py's `roofline` object is arithmetic on measured launch times and device counters: checked here (CPU) on the figures of
profiles/r05_bench.json, whose fractions VERDICT r5 recomputed by hand (k_prep_sweep 0.32 of 8 TB/s and 0.25 of 2.5 PFLOP/s;
k_sweep_min 0.47; the step 755 TFLOP/s = 0.30; the FP64-VALU roofline of SURVEY 8d exceeded 17x)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Handle(object):
    def get_option(self, name):
        return {"filter_first_range_pct": 30, "filter_second_range_pct": 50, "filter_order": 1, "fused_waves": 8}[name]


def test_step_roofline_reproduces_the_hand_computation():
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    stats = {"range_cuts": [18, 62], "second_range_groups": 16320, "third_range_groups": 8160, "uncertain_queries": 23100,
             "uncertain_pairs": 24880}
    per_launch = [0.18598, 0.0812, 0.0574, 0.0424]
    launch_ms = np.array(per_launch * 20)
    step_ms = 0.3844
    alg = bench.NPROPOSALS * (8 * bench.NDIM + 1) + 8 * bench.N_LIVE * bench.NDIM + 2 * 8 * bench.NDIM**2
    hbm = {"achieved_GBps": alg / (step_ms * 1e-3) / 1e9, "frac": alg / (step_ms * 1e-3) / 1e9 / 8000.0}
    pmc = {"per_kernel_all": {"k_prep_sweep<50, 8>": 493616128.0, "k_sweep_min<4, 4, 2>": 81673216.0, "k_uncertain<4, 4>": 68191232.0},
           "hbm_bytes_per_step_all_kernels": 731218944.0, "fresh": False, "status": "STALE: test"}
    r = bench.step_roofline(_Handle(), stats, 64, 125, (80, sum(per_launch) * 20), launch_ms, 20, True, step_ms, 0.41, alg, pmc, hbm,
                            2.63e11)
    assert r["kernel"] == "k_prep_sweep<50, 8>" and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert abs(r["frac"] - 0.32) < 0.005, r["frac"]                       # 476 MB / 185.98 us = 2.56 TB/s
    assert abs(r["other_roof"]["frac"] - 0.25) < 0.005                    # 3.56e6 MFMA x 32768 / 185.98 us = 628 TF
    assert r["launches"][0]["executed_mfma"] == 31250 * 42 + 31250 * 18 * 4
    assert r["traffic"] is None and r["traffic_imported_value"] == 493616128.0 and r["traffic_status"].startswith("STALE")
    assert abs(r["k_sweep_min"]["frac"] - 0.47) < 0.01
    assert abs(r["step"]["mfma_frac_of_2500"] - 0.30) < 0.005 and abs(r["step"]["counter_bytes_over_algorithmic"] - 1.82) < 0.01
    assert 17.0 < r["fp64_valu_roofline_of_survey_8d"]["ratio_to_that_peak"] < 17.8
    assert len(r["launches"]) == 4 and abs(sum(e["share_of_step_time"] for e in r["launches"]) - sum(per_launch) / step_ms) < 1e-12
    # the first launch with the ellipsoid form read off the whitening chain: 24 instead of 42 matrix instructions per group
    r3 = bench.step_roofline(_Handle(), dict(stats, same_quadratic_form=1), 64, 125, (80, sum(per_launch) * 20), launch_ms, 20, True,
                             step_ms, 0.41, alg, pmc, hbm, 2.63e11)
    assert r3["kernel"] == "k_prep_sweep<50, 8, true>" and r3["launches"][0]["executed_mfma"] == 31250 * 24 + 31250 * 18 * 4
    pmc["fresh"] = True
    r2 = bench.step_roofline(_Handle(), stats, 64, 125, (80, 1.0), launch_ms, 20, True, step_ms, 0.41, alg, pmc, hbm, None)
    assert r2["traffic"] == 493616128.0 and r2["fp64_valu_roofline_of_survey_8d"] is None


def test_traffic_file_is_marked_stale_when_the_kernel_sources_changed():
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    from ultranest_amd.csrc import build
    t = bench.imported_traffic()
    if t is None:
        return
    assert t["source_hash_now"] == build.source_hash()
    assert t["fresh"] == (t.get("source_hash") == build.source_hash())
    assert t["status"].startswith("STALE") != t["fresh"]
