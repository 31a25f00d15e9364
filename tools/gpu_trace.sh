#!/bin/bash
# ablations of the split-bf16 convolution kernels (tools/trace_conv_emu.py; the -DEMU_TRACE library is prebuilt on the CPU side)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/trace
{
COALIGN_EMU_PC=1 TAPK=0 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
COALIGN_EMU_PC=1 TAPK=0 RESIDUAL=0 timeout 200 python tools/trace_conv_emu.py 3 5 64 64 100 352
COALIGN_EMU_PC=1 TAPK=0 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py 3 5 256 256 25 88
} 2>&1 | grep -v amdgpu.ids | grep -E "producer|ablation|Error|error" | tee gpurun_out/trace/ablate.txt
