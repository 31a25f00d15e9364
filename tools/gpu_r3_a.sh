#!/bin/bash
# round 3, call A: full GPU suite, pillar-op kernel times with the matrix-core encoder on / off, a bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3a; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.log
for m in 1 0; do
  echo "== COALIGN_PILLAR_MFMA=$m"
  COALIGN_PILLAR_MFMA=$m bash tools/gpu_prof_kernels.sh r3a_mfma$m pillar_nhwc_persistent pillar_nhwc pillar_nchw 2>&1 | tee $OUT/kprof_mfma$m.log
done
timeout 400 python bench.py --no-cpu-baseline 2> $OUT/bench.err | tee $OUT/bench.json | cut -c1-1500
tail -5 $OUT/bench.err
