#!/bin/bash
# laboratory: kernel duration (rocprofv3 kernel trace) of the sparse pillar kernel per grid sizing / ablation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/pillar
for cfg in ${CFGS:-"PAIRS=6" "PAIRS=5" "PAIRS=8" "PAIRS=10"}; do
  rm -rf /tmp/pp
  ( cd /tmp && env COALIGN_LAB=1 COALIGN_SPARSE_${cfg} timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $ROOT/tools/kernels_only.py 30 pillar_sparse > /dev/null 2>&1 )
  python - "$cfg" <<PY
import csv,glob,sys
for f in glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pillar_sparse" in r["Name"]: print(sys.argv[1], r["Calls"], "calls, avg us", round(float(r["AverageNs"])/1e3, 2), "min", round(float(r["MinNs"])/1e3, 2))
PY
done | tee $ROOT/gpurun_out/pillar/sweep.txt
