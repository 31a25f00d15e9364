#!/bin/bash
# round 3: 4 x 8-pixel blocks in four block columns on the 88-wide maps (COALIGN_EMU_STACK bit 2): bit-equality, layer time, clock / power, frame rate A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3nbx; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1200 python -m pytest tests/test_round3_gpu.py -m gpu -x -q -k "stacked" 2>&1 | tail -5 | tee $OUT/pytest.log
for st in 5 1; do COALIGN_EMU_STACK=$st python tools/probe_power_variants.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT/power.log; done
for st in 5 1 5 1; do
  COALIGN_EMU_STACK=$st timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STACK=$st', d['value'], 'frames/s; one frame in flight p50', d['latency_ms']['one_frame_in_flight']['p50'])" | tee -a $OUT/ab.log
done
