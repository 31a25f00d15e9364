#!/bin/bash
# round 3: XCD-aware logical workgroup order in the split-bf16 convolution (COALIGN_EMU_XCD=1): bit-equality, clock / power per layer, frame rate A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3xcd; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
COALIGN_EMU_XCD=1 timeout 1200 python -m pytest tests -m gpu -x -q -k "stacked or stream_k or conv3x3_emu or emu or benchmarked_frame or pipeline_equals" 2>&1 | tail -4 | tee $OUT/pytest.log
for shape in 5,256,25,88 5,128,50,176 5,64,100,352 1,256,100,352; do for x in 1 0; do SHAPE=$shape COALIGN_EMU_XCD=$x python tools/probe_power_variants.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/XCD=$x /" | tee -a $OUT/power.log; done; done
for x in 1 0 1 0; do
  COALIGN_EMU_XCD=$x timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('XCD=$x', d['value'], 'frames/s; one frame in flight p50', d['latency_ms']['one_frame_in_flight']['p50'], 'digest0', d['frame_digests']['0'])" | tee -a $OUT/ab.log
done
