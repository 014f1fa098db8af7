"""PMC summary: per kernel of the C5 membership pipeline (scripts/stage_profile.py under rocprofv3 --pmc, separate passes:
matrix / instruction counters, wait split, FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_pmc_summary.json + profiles/pmc_scan_traffic.json.
    python scripts/collect_pmc.py gpurun_out/r05z profiles/r05
Round 5: `clock_GHz` and `matrix_pipe_busy` are derived from GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 =
"launch duration in shader cycles" -- true only for a launch that keeps every XCD busy from start to end.  Round 4 printed them
for every kernel and got 3.0 / 3.5 / 5.4 GHz for k_prep4 / k_uncertain / k_scan (short launches and launches that leave XCDs idle:
the counter keeps running on the busy ones while the wall time is that of the longest).  They are reported for the sweep launches
only (which fill the chip for their whole duration: 1.77-1.9 GHz, consistent with the in-kernel clock of the probes); the other
kernels carry null and the instruction / wait counters, which do not depend on that assumption."""
import csv
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]


def table(path):
    out, name = {}, None
    for line in open(path):
        if line.startswith(" "):
            k, v = line.split()[:2]
            out[name][k] = float(v)
        elif line.strip():
            name = line.strip()
            out[name] = {}
    return out


sq, wait = table(src + "_pmc_sq.txt"), table(src + "_pmc_wait.txt")
fetch, write = table(src + "_pmc_FETCH_SIZE.txt"), table(src + "_pmc_WRITE_SIZE.txt")
stats = {}
import glob
for f in glob.glob(src + "_stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = float(r["AverageNs"]) * 1e-6
summary = {}
for k in sq:
    if not k.startswith(("void mlf::k_sweep_min", "void mlf::k_uncertain", "void mlf::k_prep4", "void mlf::k_prep_sweep", "void mlf::k_scan",
                         "mlf::k_recheck")):
        continue
    a, w = sq[k], wait.get(k, {})
    cycles = a["GRBM_GUI_ACTIVE"] / 8.0                      # per XCD = launch duration in shader cycles
    ms = stats.get(k)
    if ms is None:      # the counter tables may carry a shortened kernel name
        ms = next((v for name, v in stats.items() if name.startswith(k) or k.startswith(name[:44])), None)
    fills_the_chip = k.startswith(("void mlf::k_sweep_min", "void mlf::k_prep_sweep"))   # see the module docstring
    e = dict(avg_ms_kernel_stats=ms, launch_cycles=cycles if fills_the_chip else None,
             clock_GHz=(cycles / (ms * 1e6)) if (ms and fills_the_chip) else None,
             SQ_INSTS_MFMA=a["SQ_INSTS_MFMA"], SQ_INSTS_VALU=a["SQ_INSTS_VALU"], SQ_INSTS_SALU=a["SQ_INSTS_SALU"],
             matrix_pipe_busy=(a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles) if (cycles and fills_the_chip) else None,
             non_matrix_instructions_per_matrix_instruction=((a["SQ_INSTS_VALU"] + a["SQ_INSTS_SALU"]) / a["SQ_INSTS_MFMA"]) if a["SQ_INSTS_MFMA"] else None,
             executed_TFLOPs=(a["SQ_INSTS_MFMA"] * 32768.0 / (ms * 1e-3) / 1e12) if ms else None)
    if w:
        e.update(wave_cycles_quads=w["SQ_WAVE_CYCLES"], parked_in_s_waitcnt=w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"],
                 issue_stalled=w["SQ_WAIT_INST_ANY"] / w["SQ_WAVE_CYCLES"], issuing=w["SQ_ACTIVE_INST_ANY"] / w["SQ_WAVE_CYCLES"],
                 waves=w["SQ_WAVES"])
    if k in fetch and k in write:
        # MI355X_MICROARCH.md, HBM / rocprofv3: FETCH_SIZE and WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 32-byte units of
        # 64-byte requests once: bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024
        e["hbm_bytes_gfx950_corrected"] = (2.0 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024.0
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
    summary[k] = e
tot = sum(v.get("hbm_bytes_gfx950_corrected", 0.0) * (2 if k.startswith("void mlf::k_sweep_min") and len([x for x in summary if x.startswith("void mlf::k_sweep_min")]) == 1 else 1)
          for k, v in summary.items())
summary["_per_step"] = dict(hbm_bytes_all_kernels=tot, algorithmic_bytes=402640000, ratio=tot / 402640000.0,
                            kernel_ms_sum=sum((v["avg_ms_kernel_stats"] or 0.0) * (2 if k.startswith("void mlf::k_sweep_min") and
                                              len([x for x in summary if x.startswith("void mlf::k_sweep_min")]) == 1 else 1)
                                              for k, v in summary.items() if not k.startswith("_")))
json.dump(summary, open(dst + "_pmc_summary.json", "w"), indent=1)
# what bench.py imports as roofline.traffic (per launch of the dominant kernel, like roofline.achieved: the k_sweep_min launches)
import os
sweeps = {k: v["hbm_bytes_gfx950_corrected"] for k, v in summary.items() if k.startswith("void mlf::k_sweep_min") and "hbm_bytes_gfx950_corrected" in v}
stats_calls = {}
for f in glob.glob(src + "_stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats_calls[r["Name"]] = int(r["Calls"])
if sweeps:
    # the ranges run the same instance: its counters are the mean over its launches already
    per_launch = sum(sweeps.values()) / len(sweeps)
    per_kernel_all = {k.replace("void mlf::", "").split("(")[0]: v["hbm_bytes_gfx950_corrected"] for k, v in summary.items()
                      if not k.startswith("_") and "hbm_bytes_gfx950_corrected" in v}
    launches = {k.replace("void mlf::", "").split("(")[0]: (2 if k.startswith("void mlf::k_sweep_min") else 1) for k in summary if not k.startswith("_")}
    step_sum = sum(per_kernel_all[k] * launches.get(k, 1) for k in per_kernel_all)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ultranest_amd.csrc import build as _build
    json.dump({"source_hash": _build.source_hash(), "hbm_bytes_per_launch": per_launch, "per_kernel_all": per_kernel_all, "launches_per_step": launches,
               "hbm_bytes_per_step_all_kernels": step_sum, "source": os.path.basename(dst) + "_pmc_summary.json",
               "kernel": "k_sweep_min<4, 4, 2> (its launches of a step -- the ranges behind the first, which rides in k_prep_sweep by default: mean over them)",
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over scripts/stage_profile.py (the bench workload, "
                       "1e6 proposals per step); (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md; per launch, like roofline.achieved"},
              open(os.path.join(os.path.dirname(dst), "pmc_scan_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
