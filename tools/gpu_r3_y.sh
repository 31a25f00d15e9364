#!/bin/bash
# round 3: rows kernel with one prologue per wavefront and double-buffered rounds: round size 3 (12 wavefronts per CU) against 5 (8)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3y; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
COALIGN_PILLAR_ROUND=3 timeout 1500 python -m pytest tests -m gpu -x -q -k "pillar or matrix_core or submit_points or encode_stream or scatter or canvas" 2>&1 | grep -v "^$" | tail -6 | tee $OUT/pytest.log
for v in "3 0 0" "5 0 0" "3 9 0" "3 0 7" "3 0 4" "3 0 3"; do
  set -- $v
  echo "== ROUND=$1 PAIRS=$2 DEBUG=$3"
  COALIGN_PILLAR_ROUND=$1 COALIGN_PILLAR_PAIRS=$2 COALIGN_PILLAR_DEBUG=$3 bash tools/gpu_prof_kernels.sh r3y_$1_$2_$3 pillar_nhwc_persistent 2>&1 | grep -E "rows_mx|prep|^\{" | tee -a $OUT/ablate.log
done
