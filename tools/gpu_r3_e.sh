#!/bin/bash
# round 3, call E: the LDS-DMA rounds version of pillar_rows_mx_kernel: pillar + feeder tests, ablations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3e; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "pillar or matrix_core or submit_points or encode_stream" 2>&1 | tail -8 | tee $OUT/pytest.log
for v in "0 0" "1 0" "3 0" "4 0" "7 0"; do
  set -- $v
  echo "== DEBUG=$1 BLOCKS=$2"
  COALIGN_PILLAR_DEBUG=$1 COALIGN_PILLAR_BLOCKS=$2 bash tools/gpu_prof_kernels.sh r3e_$1_$2 pillar_nhwc_persistent 2>&1 | grep -E "rows_mx|prep|^\{" | tee -a $OUT/ablate.log
done
