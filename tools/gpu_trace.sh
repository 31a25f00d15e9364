#!/bin/bash
# interval timelines + ablations of the split convolution kernel (tools/trace_conv_emu.py; the -DEMU_TRACE library is prebuilt on the CPU side)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/trace
{
for T in ${TERMS_LIST:-16 3}; do
TAPK=1 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py $T 5 64 64 100 352
TAPK=1 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py $T 5 128 128 50 176
TAPK=1 RESIDUAL=1 timeout 200 python tools/trace_conv_emu.py $T 5 256 256 25 88
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trace/ablate.txt
