#!/bin/bash
# kernel duration (rocprofv3 kernel trace) of the PRODUCT library's sparse pillar kernel and of the fusion kernel (tools/kernels_only.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/pillar
for rep in 1 2; do
rm -rf /tmp/pp
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $ROOT/tools/kernels_only.py 30 pillar_sparse fuse_nhwc_3scales > /dev/null 2>&1 )
python - <<PY
import csv,glob
for f in glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pillar_sparse" in r["Name"] or "warp_fuse" in r["Name"]: print(r["Name"][:60], r["Calls"], "calls, avg us", round(float(r["AverageNs"])/1e3, 2), "min", round(float(r["MinNs"])/1e3, 2))
PY
done | tee -a $ROOT/gpurun_out/pillar/product.txt
