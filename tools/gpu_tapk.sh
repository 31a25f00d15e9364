#!/bin/bash
# one gpurun call: correctness of the variants of the split-bf16 convolution, per-layer timing, whole-frame A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/tapk gpurun_out/trace
timeout 1200 python tools/bench_conv_tapk.py 2>&1 | tee gpurun_out/tapk/conv_tapk.txt
