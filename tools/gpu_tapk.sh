#!/bin/bash
# one gpurun call: correctness of the tap-major / asm-DMA variants of the split-bf16 convolution, per-layer timing, whole-frame A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/tapk
timeout 400 python -m pytest tests/test_hip_parity.py -q -x -k "tap_major" 2>&1 | tail -3
timeout 1200 python tools/bench_conv_tapk.py 2>&1 | tee gpurun_out/tapk/conv_tapk.txt
