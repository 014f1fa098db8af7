#!/bin/bash
# round 5, final GPU call: PMC passes on the membership pipeline (traffic file first: bench.py imports it), the whole GPU suite,
# smoke, bench (+ rocprofv3 kernel statistics of the same command), reference grid, mid-size and small batches, rebuild, the other
# configurations (d = 100, d = 256 included), end-to-end runs with their phase breakdown, the 8-rank rehearsal over gloo, soak
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
P=r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "== pmc"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/${P}_pmc_sq -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_sq.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_sq -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_sq.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $O/${P}_pmc_wait -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_wait.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_wait -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_wait.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${P}_pmc_$C -o st -- python $R/scripts/stage_profile.py 3 > $O/${P}_pmc_$C.log 2>&1
python $R/scripts/pmc_table.py $(find $O/${P}_pmc_$C -name "*counter_collection.csv" | head -1) > $O/${P}_pmc_$C.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_stats -o st -- python $R/scripts/stage_profile.py 20 > $O/${P}_stats.log 2>&1
head -8 $(find $O/${P}_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
cd $R
python scripts/collect_pmc.py $O/$P profiles/r05 > $O/${P}_collect.log 2>&1; tail -3 $O/${P}_collect.log | cut -c1-200
cp profiles/pmc_scan_traffic.json profiles/r05_pmc_summary.json $O/ 2>/dev/null
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -5; grep -B5 -A25 "^E " $O/pytest_gpu.log | head -50
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('rebuild_ms'), d.get('rebuild_device_resident_ms'))
r=d['roofline']; print({k:r.get(k) for k in ('achieved','frac','traffic','ms_per_launch_by_phase','achieved_by_phase')})
print(d.get('kernel_ms'))
print(d['cpu_baseline'])
PY
tail -3 $O/bench.err
cd /tmp
echo "== bench under rocprofv3 (kernel statistics of the same command)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_bench_stats -o st -- python $R/bench.py --steps 20 --warmup 5 --headline-only --no-cpu > $O/${P}_bench_stats.log 2>&1
head -12 $(find $O/${P}_bench_stats -name "*kernel_stats.csv" | head -1) | cut -c1-160
cd $R
echo "== reference grid"; timeout 600 python scripts/reference_grid_bench.py > $O/reference_grid.json 2> $O/reference_grid.err; tail -3 $O/reference_grid.err | cut -c1-300
echo "== mid-size"; timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json | cut -c1-160
echo "== small batches through the reference API"; timeout 300 python scripts/small_batch_latency.py --save > $O/small_batch.log 2>&1; tail -3 $O/small_batch.log | cut -c1-300
echo "== rebuild"; timeout 300 python scripts/rebuild_calls.py > $O/rebuild_calls.log 2>&1; head -14 $O/rebuild_calls.log | cut -c1-150
timeout 300 python scripts/rebuild_modes.py > $O/rebuild_modes.json 2> $O/rebuild_modes.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/rebuild_modes.json'))
print({k: v.get('median_ms') for k, v in d.items() if isinstance(v, dict)})
PY
echo "== config bench"; timeout 1200 python scripts/config_bench.py > $O/config_bench.json 2> $O/config_bench.err; tail -2 $O/config_bench.err | cut -c1-200
echo "== kernel trace of the d = 100 configuration"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_d100_stats -o st -- python $R/scripts/config_bench.py C5-at-d100 > $O/r05_d100.log 2>&1
cp $(find $O/r05_d100_stats -name "*kernel_stats.csv" | head -1) $O/r05_d100_kernel_stats.csv; head -8 $O/r05_d100_kernel_stats.csv | cut -c1-160
cd $R
echo "== end-to-end runs"; timeout 900 python scripts/e2e_run.py > $O/e2e_run.log 2>&1; tail -4 $O/e2e_run.log | cut -c1-700
echo "== 8 ranks on this box's single device over gloo (rehearsal of the driver's multi-GPU pass: rendezvous, barriers, collectives)"
timeout 900 python bench.py --gpus 8 --scaling strong --steps 5 --warmup 2 --no-cpu > $O/bench_8rank_selfspawn_gloo.json 2> $O/bench_8rank.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_8rank_selfspawn_gloo.json').read().strip().splitlines()[-1])
    s=d['strong_scaling']; print(d['n_gpus'], d['scaling'], d['value'], s['ranks_seen_by_allreduce'], s['collective_backend'], len(s['per_rank_ms_per_step']), s['bootstrap_result'])
except Exception as e:
    print("8-rank:", e)
PY
tail -3 $O/bench_8rank.err | cut -c1-300
echo "== soak"; SOAK_A="501 502 503 504 505 506" SOAK_B="51 52" bash scripts/gpu_soak.sh 2>&1 | tail -10
find $O -name "*.csv" -size +4M -delete
