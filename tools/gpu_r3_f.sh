#!/bin/bash
# round 3, call F: new tests, multirank check incl. gather mode, rows kernel timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3f; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests/test_round3_gpu.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 | tee $OUT/pytest_r3.log
bash tools/gpu_prof_kernels.sh r3f pillar_nhwc_persistent 2>&1 | grep -E "rows_mx|prep|^\{" | tee $OUT/kprof.log
bash tools/gpu_multirank_check.sh 2>&1 | tail -12 | tee $OUT/multirank.log
tail -3 gpurun_out/multirank/g2.err gpurun_out/multirank/g5.err 2>/dev/null | tail -12
