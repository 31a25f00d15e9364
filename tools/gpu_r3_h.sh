#!/bin/bash
# round 3, call H: stacked convolution tiles: correctness + per-layer times (stack on / off) + frame rate A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1200 python -m pytest tests/test_round3_gpu.py -m gpu -x -q -k "stacked or cfg4" 2>&1 | tail -12 | tee $OUT/pytest.log
for st in 1 0; do
  echo "== COALIGN_EMU_STACK=$st"
  COALIGN_EMU_STACK=$st NO_BENCH=1 SETTINGS=tapk_v3_r0 timeout 300 python tools/bench_conv_tapk.py 2>&1 | tail -8 | tee $OUT/layers_stack$st.log
done
for st in 1 0 1 0; do
  COALIGN_EMU_STACK=$st timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STACK=$st', d['value'], 'frames/s')" | tee -a $OUT/ab.log
done
