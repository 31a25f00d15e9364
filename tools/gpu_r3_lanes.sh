#!/bin/bash
# round 3: frames in flight with the final kernels (graph replays), same box, two passes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3lanes; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for rep in 1 2; do for l in 2 3 4; do
  timeout 300 python bench.py --lanes $l --no-cpu-baseline --no-side-modes --no-from-points --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lanes=$l', d['value'], 'frames/s')" | tee -a $OUT/ab.log
done; done
