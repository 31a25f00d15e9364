#!/bin/bash
# round 3, call G: rows kernel with per-round metadata; pillar / feeder tests + the cfg 4 sweep; ablations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3g; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "pillar or matrix_core or submit_points or encode_stream or cfg4_correction" 2>&1 | grep -v "^$" | tail -14 | tee $OUT/pytest.log
for v in "0 0" "7 0"; do
  set -- $v
  echo "== DEBUG=$1 BLOCKS=$2"
  COALIGN_PILLAR_DEBUG=$1 COALIGN_PILLAR_BLOCKS=$2 bash tools/gpu_prof_kernels.sh r3g_$1_$2 pillar_nhwc_persistent 2>&1 | grep -E "rows_mx|prep|^\{" | tee -a $OUT/ablate.log
done
