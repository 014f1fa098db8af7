#!/usr/bin/env python3
"""Summary of the rocprofv3 passes over the REBUILD kernels (scripts/gpu_round.sh: scripts/rebuild_modes.py under
--pmc SQ counters, FETCH_SIZE, WRITE_SIZE and --stats) -> profiles/<tag>_rebuild_pmc_summary.json.
Per kernel: launches, average duration, instruction mix per launch, wave-time share spent waiting, HBM bytes per launch
(gfx950 correction of MI355X_MICROARCH.md: 2 x FETCH_SIZE + WRITE_SIZE, KiB), and for the distance kernels the
fraction of the non-fused FP64 vector roofline (39.3 T op/s at 2.4 GHz) of their algorithmic work."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
N, D = 4000, 50
KEYS = ("k_boot<", "k_boot_final", "k_boot_index", "k_boot_mean", "k_boot_cov", "k_boot_chol", "k_boot_solvemax", "k_scan<",
        "k_subtract_accum", "k_pack_selection", "k_build_layouts", "k_prep<", "k_quant_refs", "k_ref_")


def key_of(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def table(sub):
    f = glob.glob(os.path.join(G, sub, "*counter_collection.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            k = key_of(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


out = {}
sq = table("pmc_rebuild_SQ")
fe = table("pmc_rebuild_FETCH_SIZE")
wr = table("pmc_rebuild_WRITE_SIZE")
act = table("pmc_rebuild_ACT")
dur = collections.defaultdict(list)
tr = glob.glob(os.path.join(G, "prof_rebuild", "*kernel_trace.csv"))
if tr:
    for r in csv.DictReader(open(tr[0])):
        k = key_of(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in KEYS:
    if k not in sq and k not in dur:
        continue
    e = {}
    if dur[k]:
        e["launches_in_trace"] = len(dur[k])
        e["avg_us"] = sum(dur[k]) / len(dur[k])
    if k in sq:
        c = {n: sum(v) / len(v) for n, v in sq[k].items()}
        e["per_launch"] = {n: c[n] for n in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM") if n in c}
        if c.get("SQ_WAVE_CYCLES"):
            e["wave_time_waiting_to_issue"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("GRBM_GUI_ACTIVE") and e.get("avg_us", 0.0) >= 50.0:   # shorter launches: the counter's idle lead-in dominates
            e["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (e["avg_us"] * 1e3)
    if k in act:   # second SQ pass: SQ_WAIT_ANY counts every wait (s_waitcnt included), SQ_WAIT_INST_ANY the wait for an issue slot
        c = {n: sum(v) / len(v) for n, v in act[k].items()}
        if c.get("SQ_WAVE_CYCLES"):
            e["activity_pass"] = {
                "wave_time_waiting_for_anything": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                "wave_time_waiting_to_issue": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                "wave_time_issuing_vector_instructions": c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"],
                "vector_instructions_per_launch": c.get("SQ_INSTS_VALU"),
                "vector_memory_instructions_per_launch": c.get("SQ_INSTS_VMEM"),
                "waves": c.get("SQ_WAVES")}
    if k in fe:
        f_kb = sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"])
        w_kb = sum(wr[k]["WRITE_SIZE"]) / len(wr[k]["WRITE_SIZE"]) if k in wr and wr[k].get("WRITE_SIZE") else 0.0
        e["hbm_bytes_per_launch"] = (2.0 * f_kb + w_kb) * 1024.0
    out[k] = e
flops = {"k_boot<": 3.0 * D * N * N, "k_scan<": 3.0 * D * N * N}
for k, fl in flops.items():
    if k in out and out[k].get("avg_us"):
        tf = fl / (out[k]["avg_us"] * 1e-6) / 1e12
        out[k]["algorithmic_TFLOPs"] = tf
        out[k]["frac_of_fp64_valu_roofline"] = tf / 39.3
        out[k]["algorithmic_flops_per_launch"] = fl
out["_note"] = ("kernels of scripts/rebuild_modes.py (C5: N = 4000, d = 50, 30 bootstrap rounds; default and device-resident "
                "rebuilds); k_scan< rows are the all-pairs FLAGS passes (clustering, subtract_nearby) mixed with the small "
                "membership scans of the rebuild; counters are averages over all launches of the kernel in the pass")
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_rebuild_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1))
