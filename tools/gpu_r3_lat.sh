#!/bin/bash
# round 3: one frame in flight (latency mode): does the 6 x 32 stacked variant (better balance inside a launch) pay there?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3lat; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for rep in 1 2; do for st in 3 1; do
  COALIGN_EMU_STACK=$st timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STACK=$st', d['value'], 'frames/s; one frame in flight', d['latency_ms']['one_frame_in_flight'])" | tee -a $OUT/ab.log
done; done
