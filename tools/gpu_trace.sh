#!/bin/bash
# interval timelines + ablations of the split-bf16 convolution (tools/trace_conv_emu.py; the -DEMU_TRACE library is prebuilt on the CPU side)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/trace
{
TAPK=1 WAVES=12 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
TAPK=1 WAVES=8 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 256 256 25 88
TAPK=0 WAVES=8 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
} 2>&1 | grep -v amdgpu.ids | grep -E "tap-major|tap pairs|ablation|duration" | tee gpurun_out/trace/ablate.txt
