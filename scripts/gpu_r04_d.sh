cd $GRAFT_REPO_ROOT
scripts/probes/bin/sweep_src_probe 2 > gpurun_out/r04_sweep_src_probe2.json 2>&1; cat gpurun_out/r04_sweep_src_probe2.json
scripts/probes/bin/mfma16_issue_probe 2>&1 | head -4
