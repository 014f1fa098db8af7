#!/bin/bash
# the launch paths of bench.py with more than one rank on the one GPU this pool offers (control flow of N = 2; RCCL with one rank)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== bench via torchrun (1 rank, RCCL init)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?"
echo "== plain --gpus 2 (self-spawn; gloo + both ranks on device 0 because this box has one GPU)"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "selfspawn rc=$?"
echo "== the same with --scaling strong"; timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu --scaling strong > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err; echo "strong rc=$?"
echo "== N = 2 through torchrun (gloo collectives, both pinned to device 0)"; MLF_BENCH_DEVICE=0 MLF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2-rank rc=$?"
python - <<'PY'
import json
for n in ("bench_torchrun", "bench_2rank_selfspawn", "bench_2rank_strong", "bench_2rank_gloo"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d.get("strong_scaling", {}).get("bootstrap30_sharded_ms"), d.get("strong_scaling", {}).get("bootstrap_result"))
    except Exception as e:
        print(n, "failed:", e)
PY
tail -2 $O/bench_2rank_selfspawn.err | cut -c1-200
