#!/bin/bash
# one gpurun call: correctness of the variants of the split-bf16 convolution, per-layer timing, whole-frame A/B, timelines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/tapk gpurun_out/trace
timeout 1200 python tools/bench_conv_tapk.py 2>&1 | tee gpurun_out/tapk/conv_tapk.txt
{
COALIGN_EMU_PRIO=1 TAPK=0 WAVES=8 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trace/trace.txt
