#!/bin/bash
# One GPU-box session: parity tests, smoke, headline bench, row benches, rocprofv3 kernel stats, PMC passes.
# Everything lands under gpurun_out/ (merged back by gpurun); scripts/collect_profiles.py r03 then copies the summaries
# that are to be judged into profiles/.    usage: scripts/gpu_round.sh [noprof] [quick]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
echo "== rebuild modes"; timeout 300 python scripts/rebuild_modes.py > $O/rebuild_modes.json 2> $O/rebuild_modes.err; grep -n "median" $O/rebuild_modes.json
echo "== mid-size batches"; timeout 300 python scripts/midsize_profile.py > $O/midsize.json 2> $O/midsize.err; cat $O/midsize.json
echo "== host API"; timeout 300 python scripts/host_api_bench.py > $O/host_api.json 2> $O/host_api.err; tail -c 600 $O/host_api.json
echo "== likelihood kernels"; timeout 300 python scripts/loglike_bench.py > $O/loglike_bench.json 2> $O/loglike_bench.err; grep -c ms $O/loglike_bench.json
echo "== where inside(active_u) goes"; timeout 200 python scripts/active_u_breakdown.py --save > $O/active_u_breakdown.log 2>&1; tail -9 $O/active_u_breakdown.log | tr -d '\n'; echo
echo "== FP64 / DPP issue probes"; timeout 100 scripts/probes/bin/fp64_issue_probe > $O/fp64_issue_probe.json 2>&1; timeout 60 scripts/probes/bin/minphase_probe > $O/minphase_probe.json 2>&1; wc -l $O/fp64_issue_probe.json $O/minphase_probe.json
echo "== small batches through the reference API"; timeout 300 python scripts/small_batch_latency.py --save > $O/small_batch.log 2>&1; tail -2 $O/small_batch.log | cut -c1-200
if [ "$2" != "quick" ]; then
echo "== step sampler bench (row f1)"; timeout 600 python scripts/walk_bench.py device > $O/walk_bench.log 2>&1; tail -3 $O/walk_bench.log
echo "== device sampling bench (row f2)"; timeout 300 python scripts/sample_bench.py > $O/sample_bench.json 2> $O/sample_bench.err; tail -3 $O/sample_bench.err
echo "== bench via torchrun (1 rank, RCCL init)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?"
echo "== plain --gpus 2 (self-spawn; gloo + both ranks on device 0 because this box has one GPU)"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "selfspawn rc=$?"
echo "== the same with --scaling strong"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu --scaling strong > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err; echo "strong rc=$?"
echo "== N = 2 control flow on this one GPU (2 processes, gloo collectives, both pinned to device 0)"; MLF_BENCH_DEVICE=0 MLF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank rc=$?"
echo "== size check (P = 8e6, N = 2e5)"; timeout 600 python scripts/big_batch_check.py > $O/big_batch.json 2> $O/big_batch.err; tail -4 $O/big_batch.json
echo "== end-to-end run (eggbox d=2, N=1000, device-resident batches)"; timeout 300 python scripts/e2e_run.py > $O/e2e_run.log 2>&1; tail -1 $O/e2e_run.log | cut -c1-300
echo "== config bench"; timeout 600 python scripts/config_bench.py > $O/config_bench.json 2> $O/config_bench.err; tail -2 $O/config_bench.err
fi
if [ "$1" != "noprof" ]; then
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --headline-only > $O/prof_stats.log 2>&1
tail -2 $O/prof_stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 pmc $C"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --headline-only > $O/pmc_$C.log 2>&1
  tail -1 $O/pmc_$C.log
done
echo "== rocprofv3 pmc SQ"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_SQ -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --headline-only > $O/pmc_SQ.log 2>&1
tail -1 $O/pmc_SQ.log
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_SQ2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --headline-only > $O/pmc_SQ2.log 2>&1
tail -1 $O/pmc_SQ2.log
echo "== rocprofv3 pmc over the REBUILD kernels (k_boot, k_scan flags, k_subtract_accum, k_boot_*): instruction mix and traffic"
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_rebuild_SQ -o rb -- python $R/scripts/rebuild_modes.py > $O/pmc_rebuild_SQ.log 2>&1
python $R/scripts/pmc_table.py $(find $O/pmc_rebuild_SQ -name "*counter_collection.csv" | head -1) 1000 > $O/pmc_rebuild_SQ.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/pmc_rebuild_ACT -o rb -- python $R/scripts/rebuild_modes.py > $O/pmc_rebuild_ACT.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_rebuild_$C -o rb -- python $R/scripts/rebuild_modes.py > $O/pmc_rebuild_$C.log 2>&1
  python $R/scripts/pmc_table.py $(find $O/pmc_rebuild_$C -name "*counter_collection.csv" | head -1) 1000 > $O/pmc_rebuild_$C.txt 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rebuild -o rb -- python $R/scripts/rebuild_modes.py > $O/prof_rebuild.log 2>&1
find $O -name "*.csv" | head -40
# keep the merge-back small: drop anything big
find $O -size +8M -delete
fi
