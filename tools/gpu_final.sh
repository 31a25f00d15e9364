#!/bin/bash
# final check of the round: full GPU suite, then an A/B of the 26 x 16 tile rule inside the same call
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/final; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for v in 0 1; do
  COALIGN_EMU_TAPK_26=$v timeout 300 python bench.py --no-cpu-baseline --no-side-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('COALIGN_EMU_TAPK_26=$v fps', d['value'], 'reproducible', d.get('frame_digests_reproducible'))" | tee -a gpurun_out/final/ab26.txt
done
