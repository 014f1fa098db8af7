#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scripts/walk_bench.py device > gpurun_out/walk_bench.log 2>&1; tail -14 gpurun_out/walk_bench.log
# distributed launch path of bench.py exactly as the driver does it (one rank, RCCL init)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/bench_torchrun.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?"; tail -c 400 gpurun_out/bench_torchrun.json; tail -3 gpurun_out/bench_torchrun.err
