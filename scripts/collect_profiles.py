#!/usr/bin/env python3
"""Copy the judged summaries of a GPU session from gpurun_out/ (scratch) into profiles/ (tracked):
rocprofv3 kernel stats, per-kernel PMC averages for the headline launches, the bench line, test logs.

    python scripts/collect_profiles.py r01
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
os.makedirs(P, exist_ok=True)

# kernels of the headline step (1e6 proposals per launch in the bench's --headline-only pass) and the smallest grid
# (threads) such a launch has: the same kernels also run on the 4000 live points during the region build
HEADLINE_GRID = {"k_prep4": 100000, "k_prep3": 100000, "k_sweep<4, 4, true": 400000,
                 "k_sweep<4, 2, false": 400000, "k_recheck_whiten": 400000, "k_scan": 400000,
                 "k_phase_finish": 64}


def headline(name, grid):
    for key, least in HEADLINE_GRID.items():
        if key in name and grid >= least:
            return key
    return None


def cp(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        print("copied", dst)


cp("prof_stats/bench_kernel_stats.csv", "%s_rocprofv3_kernel_stats.csv" % tag)
cp("bench.json", "%s_bench.json" % tag)
cp("pytest_gpu.log", "%s_pytest_gpu.log" % tag)
cp("smoke.log", "%s_smoke.log" % tag)
cp("microbench.log", "%s_microbench.log" % tag)
cp("config_bench.json", "%s_config_bench.json" % tag)
cp("walk_bench_device.json", "%s_walk_bench.json" % tag)
cp("sample_bench.json", "%s_sample_bench.json" % tag)
cp("big_batch.json", "%s_big_batch.json" % tag)
cp("e2e_run.json", "%s_e2e_run.json" % tag)
cp("bench_torchrun.json", "%s_bench_torchrun.json" % tag)
cp("bench_2rank_gloo.json", "%s_bench_2rank_gloo.json" % tag)
cp("bench_2rank_selfspawn.json", "%s_bench_2rank_selfspawn_gloo.json" % tag)
cp("small_batch.json", "%s_small_batch.json" % tag)
cp("rebuild_modes.json", "%s_rebuild_modes.json" % tag)
cp("midsize.json", "%s_midsize_batches.json" % tag)
cp("host_api.json", "%s_host_api.json" % tag)
cp("loglike_bench.json", "%s_loglike_bench.json" % tag)
cp("active_u_breakdown.json", "%s_active_u_breakdown.json" % tag)
cp("fp64_issue_probe.json", "%s_fp64_issue_probe.json" % tag)
cp("minphase_probe.json", "%s_minphase_probe.json" % tag)

# kernel-trace: average duration of the headline launches only (the stats CSV mixes them with the
# small scans of the region rebuild)
trace = os.path.join(G, "prof_stats", "bench_kernel_trace.csv")
summary = {}
if os.path.exists(trace):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = headline(r["Kernel_Name"], int(r["Grid_Size_X"]))
        if key:
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for key, v in dur.items():
        summary[key] = dict(headline_launches=len(v), avg_ms=sum(v) / len(v), min_ms=min(v), max_ms=max(v))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ", "pmc_SQ2", "pmc_SQ3"):
    f = os.path.join(G, sub, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        key = headline(r["Kernel_Name"], int(r["Grid_Size"]))
        if key:
            pmc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc[key]["VGPR_Count"] = [float(r["VGPR_Count"])]
            pmc[key]["LDS_Block_Size"] = [float(r["LDS_Block_Size"])]
for key in pmc:
    summary.setdefault(key, {})["pmc_avg_per_launch"] = {c: sum(v) / len(v) for c, v in pmc[key].items()}

traffic_total = 0.0
for key in pmc:
    if "FETCH_SIZE" in pmc[key]:
        f_kb = sum(pmc[key]["FETCH_SIZE"]) / len(pmc[key]["FETCH_SIZE"])
        w = pmc[key].get("WRITE_SIZE", [0.0])
        w_kb = sum(w) / max(1, len(w))
        summary[key]["hbm_traffic"] = dict(FETCH_SIZE_KiB=f_kb, WRITE_SIZE_KiB=w_kb, bytes_raw=(f_kb + w_kb) * 1024.0,
                                           bytes_gfx950_corrected=(2.0 * f_kb + w_kb) * 1024.0)
# the roofline entry of bench.py is per k_filter launch, averaged over the two launches of a step
mainkeys = [k for k in ("k_sweep<4, 4, true", "k_sweep<4, 2, false") if k in pmc and "FETCH_SIZE" in pmc[k]]
if not mainkeys and "k_scan" in pmc and "FETCH_SIZE" in pmc["k_scan"]:
    mainkeys = ["k_scan"]
if mainkeys:
    # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # reports half the bytes of a wide coalesced read -> doubled as the guide prescribes
    per_kernel = dict((k, summary[k]["hbm_traffic"]["bytes_gfx950_corrected"]) for k in mainkeys)
    traffic = sum(per_kernel.values()) / len(per_kernel)
    per_kernel_all = dict((k, summary[k]["hbm_traffic"]["bytes_gfx950_corrected"]) for k in pmc if "hbm_traffic" in summary.get(k, {}))
    json.dump(dict(hbm_bytes_per_launch=traffic, per_kernel=per_kernel, per_kernel_all=per_kernel_all,
                   hbm_bytes_per_step_all_kernels=sum(per_kernel_all.values()), source="%s_pmc_summary.json" % tag,
                   kernel=" + ".join(mainkeys),
                   note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --headline-only, "
                        "launches of 1e6 proposals; (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md; "
                        "mean over the launches of a step, like roofline.achieved"),
              open(os.path.join(P, "pmc_scan_traffic.json"), "w"), indent=1)
    # the bench line copied above was produced before these counters existed on this box: give it the traffic
    # of the same session (bench.py reads profiles/pmc_scan_traffic.json from now on)
    bj = os.path.join(P, "%s_bench.json" % tag)
    if os.path.exists(bj):
        line = json.load(open(bj))
        if isinstance(line.get("roofline"), dict) and "k_sweep" in str(line["roofline"].get("kernel", "")):
            line["roofline"]["traffic"] = traffic
            json.dump(line, open(bj, "w"))
json.dump(summary, open(os.path.join(P, "%s_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
