#!/bin/bash
# pointwise layers: pixel-block mapping (COALIGN_PW_PB=1 = the round-1 mapping) -- tests, per-layer timing, whole-frame A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/pw
timeout 300 python -m pytest tests -q -m gpu -k "pointwise or model_mini or full_frame" 2>&1 | tail -3
for v in 1 4; do echo "COALIGN_PW_PB=$v"; COALIGN_PW_PB=$v timeout 200 python tools/bench_pointwise.py 2>&1 | grep -v amdgpu.ids | cut -c1-200; done | tee gpurun_out/pw/pointwise.txt
bash tools/ab_bench.sh COALIGN_PW_PB 1 4 2>&1 | tee gpurun_out/pw/ab.txt
