#!/bin/bash
# One GPU-box session: parity tests, smoke, headline bench, rocprofv3 kernel stats, PMC passes.
# Everything lands under gpurun_out/ (merged back by gpurun); scripts/collect_profiles.py then
# copies the summaries that are to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -3 $O/bench.err
echo "== step sampler bench (row f1)"; timeout 600 python scripts/walk_bench.py device > $O/walk_bench.log 2>&1; tail -3 $O/walk_bench.log
echo "== device sampling bench (row f2)"; timeout 300 python scripts/sample_bench.py > $O/sample_bench.json 2> $O/sample_bench.err; tail -3 $O/sample_bench.err
echo "== bench via torchrun (1 rank, RCCL init)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?"
echo "== plain --gpus 2 (self-spawn; gloo + both ranks on device 0 because this box has one GPU)"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "selfspawn rc=$?"
echo "== small batches through the reference API"; timeout 300 python scripts/small_batch_latency.py --save > $O/small_batch.log 2>&1; tail -2 $O/small_batch.log | cut -c1-200
echo "== N = 2 control flow on this one GPU (2 processes, gloo collectives, both pinned to device 0)"; MLF_BENCH_DEVICE=0 MLF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank rc=$?"
echo "== size check (P = 8e6, N = 2e5)"; timeout 600 python scripts/big_batch_check.py > $O/big_batch.json 2> $O/big_batch.err; tail -4 $O/big_batch.json
echo "== end-to-end run (eggbox d=2, N=1000, device-resident batches)"; timeout 300 python scripts/e2e_run.py > $O/e2e_run.log 2>&1; tail -1 $O/e2e_run.log | cut -c1-300
echo "== config bench"; timeout 600 python scripts/config_bench.py > $O/config_bench.json 2> $O/config_bench.err; tail -2 $O/config_bench.err
if [ "$1" != "noprof" ]; then
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --headline-only > $O/prof_stats.log 2>&1
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
find $O -name "*.csv" | head -30
# keep the merge-back small: drop anything big
find $O -size +8M -delete
fi
