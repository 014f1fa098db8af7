#!/bin/bash
# first GPU contact: kernel parity tests + micro-benchmark; logs under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_first.log
tail -15 gpurun_out/pytest_first.log
timeout 300 python scripts/microbench.py > gpurun_out/microbench.log 2>&1
cat gpurun_out/microbench.log | tail -20
