#!/bin/bash
# frame-rate A/B of COALIGN_EMU_STACK values (same box, alternating): bash tools/gpu_r3_ab.sh "13 5 13 5"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3ab; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for st in ${1:-13 5 13 5}; do
  COALIGN_EMU_STACK=$st timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points 2>$OUT/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STACK=$st', d['value'], 'frames/s; one frame in flight p50', d['latency_ms']['one_frame_in_flight']['p50'], 'digest0', d['frame_digests']['0'])" | tee -a $OUT/ab.log || tail -3 $OUT/err.log
done
