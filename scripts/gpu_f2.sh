#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_philox.py tests/test_distributed.py tests/test_gpu_filter.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_f2.log; tail -25 gpurun_out/pytest_f2.log
timeout 300 python scripts/sample_bench.py > gpurun_out/sample_bench.json 2> gpurun_out/sample_bench.err; cat gpurun_out/sample_bench.json; tail -5 gpurun_out/sample_bench.err
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
