#!/bin/bash
# soak: the random-shape tests of the filter / small path / one-launch path under shifted seeds, and the full-size cases
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/soak.log
for off in ${SOAK_A:-301 302 303 304 305 306 307 308 309 310 311 312}; do
  MLF_FUZZ_OFFSET=$off timeout 300 python -m pytest tests/test_gpu_filter.py tests/test_small_path.py -m gpu -q -k "random or fuzz or shapes or replace" 2>&1 | tail -1 >> $O/soak.log
done
for off in ${SOAK_B:-41 42 43}; do
  MLF_FUZZ_OFFSET=$off timeout 600 python -m pytest tests/test_config_sizes.py -m gpu -q -k "full_size or c5" 2>&1 | tail -1 >> $O/soak.log
done
cat $O/soak.log
