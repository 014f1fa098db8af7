# ablation timings of k_filter (results are wrong on purpose): rebuilds mlf_filter.o on the box
cd $GRAFT_REPO_ROOT
for abl in ${ABLS:-1 2 3}; do
  touch ultranest_amd/csrc/mlf_filter.hip
  MLF_ABL=$abl python ultranest_amd/csrc/build.py > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl$abl -o st -- python $GRAFT_REPO_ROOT/scripts/stage_profile.py 10 > $GRAFT_REPO_ROOT/gpurun_out/abl$abl.log 2>&1)
  python - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/abl$abl/st_kernel_stats.csv')):
    if 'k_filter' in r['Name']: print('ABL $abl', r['Name'][:48], r['Calls'], r['AverageNs'])
PY
done
