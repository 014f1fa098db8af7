#!/bin/bash
# quick GPU session: given pytest args + optional microbench
mkdir -p gpurun_out
timeout 900 python -m pytest $1 -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_quick.log; tail -25 gpurun_out/pytest_quick.log
if [ "$2" == "bench" ]; then
  timeout 300 python scripts/microbench.py > gpurun_out/microbench.log 2>&1; tail -12 gpurun_out/microbench.log
  MLF_NOFILTER=1 timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1800 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
fi
