#!/bin/bash
# round 6, final GPU call: PMC passes on the membership pipeline (traffic file first: bench.py imports it and checks its source
# hash), the whole GPU suite, smoke, bench (+ rocprofv3 kernel statistics of the same command), mid-size / small batches, rebuild,
# the other configurations, end-to-end runs (NOT under the profiler: 2e5 graph launches), f1 / f2 benches, 8-rank rehearsal, soak
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
P=r06z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== pmc"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/${P}_pmc_sq -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_sq.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_sq -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/${P}_pmc_wait -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_wait.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_wait -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_wait.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${P}_pmc_$C -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_$C -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_$C.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_stats -o st -- python $R/scripts/stage_profile.py 20 > $O/${P}_stats.log 2>&1
head -8 $(find $O/${P}_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
cp $(find $O/${P}_stats -name "*kernel_stats.csv" | head -1) $O/r06_rocprofv3_kernel_stats.csv
cd $R
python scripts/collect_pmc.py $O/$P profiles/r06 > $O/${P}_collect.log 2>&1; tail -3 $O/${P}_collect.log | cut -c1-200
cp profiles/pmc_scan_traffic.json profiles/r06_pmc_summary.json $O/ 2>/dev/null
for f in sq wait FETCH_SIZE WRITE_SIZE; do cp $O/${P}_pmc_$f.txt $O/r06_pmc_$f.txt; done
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/r06_pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/r06_pytest_gpu.log | tail -5; grep -B5 -A25 "^E " $O/r06_pytest_gpu.log | head -50
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -2 $O/r06_smoke.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/r06_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'), d.get('rebuild_device_resident_ms'), d['strong_scaling']['ms_per_step'])
r=d['roofline']; print({k:r.get(k) for k in ('kernel','bound','achieved','frac','traffic','traffic_status','ms_per_launch')}); print(r['other_roof'])
print([(e['kernel'], round(e['ms'],4), round(e['mfma_frac_of_2500'],3), round(e['hbm_frac_of_8000'],3), e['counter_bytes']) for e in r['launches']])
print(r['step']); print(r['k_sweep_min']['frac'], r['fp64_valu_roofline_of_survey_8d'] and r['fp64_valu_roofline_of_survey_8d']['ratio_to_that_peak'])
print(d['kernel_ms']['wall_ms_per_step_of_the_launch_event_pass'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_mask_equals_cpu_mask_on_sample'], d['host_api'] and d['host_api']['proposals_per_s'])
PY
tail -3 $O/r06_bench.err
cd /tmp
echo "== bench under rocprofv3 (kernel statistics of the same command)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_bench_stats -o st -- python $R/bench.py --steps 20 --warmup 5 --headline-only --no-cpu > $O/${P}_bench_stats.log 2>&1
head -8 $(find $O/${P}_bench_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
cp $(find $O/${P}_bench_stats -name "*kernel_stats.csv" | head -1) $O/r06_bench_rocprofv3_kernel_stats.csv
cd $R
echo "== mid-size"; timeout 200 python scripts/midsize_profile.py > $O/r06_midsize_batches.json 2> $O/r06_midsize.err; cat $O/r06_midsize_batches.json | cut -c1-120
bash scripts/gpu_trace_mid.sh r06z 16384 65536 131072 2>&1 | grep "k_prep4\|k_sweep\|k_scan" | head -12 > $O/r06_midsize_traces.txt; cat $O/r06_midsize_traces.txt
echo "== small batches through the reference API"; timeout 200 python scripts/small_batch_latency.py --save > $O/r06_small_batch.log 2>&1; tail -3 $O/r06_small_batch.log | cut -c1-300
echo "== rebuild"; timeout 200 python scripts/rebuild_calls.py > $O/r06_rebuild_calls.log 2>&1; head -12 $O/r06_rebuild_calls.log | cut -c1-150
echo "== config bench"; timeout 600 python scripts/config_bench.py > $O/r06_config_bench.json 2> $O/r06_config_bench.err; tail -2 $O/r06_config_bench.err | cut -c1-200
echo "== end-to-end runs"; timeout 300 python scripts/e2e_run.py nsteps10=40,80 > $O/r06_e2e_run.log 2>&1; tail -4 $O/r06_e2e_run.log | cut -c1-330; cp $O/e2e_run.json $O/r06_e2e_run.json
echo "== f1 / f2 benches"
timeout 200 python scripts/walk_rounds_profile.py 1500 > $O/r06_walk_rounds.json 2> $O/r06_walk_rounds.err; cat $O/r06_walk_rounds.json
timeout 300 python scripts/walk_bench.py > $O/r06_walk_bench.json 2> $O/r06_walk_bench.err; tail -c 600 $O/r06_walk_bench.json
timeout 300 python scripts/sample_bench.py > $O/r06_sample_bench.json 2> $O/r06_sample_bench.err; tail -c 700 $O/r06_sample_bench.json
timeout 200 python scripts/refill_profile.py 10 --all > $O/r06_refill.json 2> $O/r06_refill.err; cat $O/r06_refill.json
echo "== 8 ranks on this box's single device over gloo (rehearsal of the driver's multi-GPU pass)"
timeout 600 python bench.py --gpus 8 --scaling strong --steps 5 --warmup 2 --no-cpu > $O/r06_bench_8rank_selfspawn_gloo.json 2> $O/r06_bench_8rank.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_bench_8rank_selfspawn_gloo.json').read().strip().splitlines()[-1])
    s=d['strong_scaling']; print(d['n_gpus'], d['scaling'], d['value'], s['ranks_seen_by_allreduce'], s['collective_backend'], s['collective_library'], s['ranks_agree_on_the_shared_rows'], len(s['per_rank_ms_per_step']))
except Exception as e:
    print("8-rank:", e)
PY
tail -3 $O/r06_bench_8rank.err | cut -c1-300
echo "== soak"; SOAK_A="701 702 703 704" SOAK_B="71" bash scripts/gpu_soak.sh 2>&1 | tail -6; cp $O/soak.log $O/r06_soak.log
find $O -name "*.csv" -size +4M -delete
