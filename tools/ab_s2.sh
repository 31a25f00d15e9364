#!/bin/bash
# A/B of the strided-layer route (COALIGN_S2_SP = 0 | sparse | all) inside ONE gpurun call: kernel bench, then the frame pipeline, alternating
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 200 python tools/bench_conv_s2.py 2>/dev/null | tail -1
for rep in 1 2; do
  for v in 0 sparse all; do
    env COALIGN_S2_SP=$v timeout 300 python bench.py --no-cpu-baseline --no-numerics --no-side-modes --no-from-points "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('S2_SP=$v rep $rep fps', d['value'], 'lat1', (d.get('latency_ms') or {}).get('one_frame_in_flight', {}).get('p50'))"
  done
done
