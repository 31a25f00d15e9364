#!/bin/bash
# kernel-trace timeline of the frame pipeline for each value of COALIGN_S2_SP given as arguments
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp
for v in "$@"; do
  OUT=$ROOT/gpurun_out/timeline_$v; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && COALIGN_S2_SP=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/bench.py --steps 200 --no-cpu-baseline --no-numerics --no-side-modes --no-size-sweep --no-from-points --no-latency > $OUT/bench.log 2>&1 )
  f=$(find $OUT -name "*kernel_trace.csv" | head -1)
  echo "== COALIGN_S2_SP=$v: $(grep -o '"value": [0-9.]*' $OUT/bench.log | head -1)"
  python $ROOT/tools/timeline_busy.py $f 150
  rm -f $f
done
