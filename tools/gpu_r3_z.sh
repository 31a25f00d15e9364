#!/bin/bash
# round 3: pointwise layers on the split-bf16 matrix cores: tests, isolated kernel times next to the fp32 kernel's, frame rate A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3z; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "pointwise or nhwc_route or benchmarked_frame or golden or backbone" 2>&1 | grep -v "^$" | tail -8 | tee $OUT/pytest.log
bash tools/gpu_prof_kernels.sh r3z pointwise_up4 pointwise_up4_bf16x3 pointwise_up2 pointwise_up2_bf16x3 pointwise_up1 pointwise_up1_bf16x3 pointwise_skip1_canvas pointwise_skip1_canvas_bf16x3 pointwise_skip2 pointwise_skip2_bf16x3 pointwise_skip3 pointwise_skip3_bf16x3 2>&1 | tail -12 | tee $OUT/iso.log
for pw in 1 0 1 0; do
  COALIGN_PW_EMU=$pw timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PW_EMU=$pw', d['value'], 'frames/s')" | tee -a $OUT/ab.log
done
