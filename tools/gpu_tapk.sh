#!/bin/bash
# one gpurun call: correctness of the tap-major / asm-DMA variants of the split-bf16 convolution, per-layer timing, whole-frame A/B, timelines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/tapk gpurun_out/trace
COALIGN_EMU_TAPK_VAR=${TVAR:-3} timeout 400 python -m pytest tests/test_hip_parity.py -q -x -k "tap_major" 2>&1 | tail -3
timeout 1200 python tools/bench_conv_tapk.py 2>&1 | tee gpurun_out/tapk/conv_tapk.txt
{
COALIGN_EMU_TAPK_VAR=${TVAR:-3} TAPK=1 WAVES=12 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
COALIGN_EMU_TAPK_VAR=${TVAR:-3} TAPK=1 WAVES=8 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 256 256 25 88
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trace/trace.txt
