#!/bin/bash
# A/B of one environment switch inside ONE gpurun call (boxes of the pool differ by ~10 %): usage: bash tools/ab_bench.sh VAR valA valB [bench args]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; VAR=$1; A=$2; B=$3; shift 3
cd $ROOT; mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-numerics --no-side-modes --no-from-points "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$v rep $rep fps', d['value'], 'lat1', (d.get('latency_ms') or {}).get('one_frame_in_flight', {}).get('p50'), 'iso', d['roofline']['isolated_ms'])"
  done
done
