#!/bin/bash
# rocprofv3 kernel stats of tools/kernels_only.py (hand-written kernels only). Usage: bash tools/gpu_prof_kernels.sh <tag> [ops...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift; OUT=$ROOT/gpurun_out/kprof_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/tools/kernels_only.py 20 "$@" > $OUT/run.log 2>&1 )
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'anonymous' in r['Name'] or 'rocclr' in r['Name']:
        print(f"{r['Name'][:96]:96s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f}")
PY
grep "^{" $OUT/run.log
