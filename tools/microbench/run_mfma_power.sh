#!/bin/bash
set -e
cd "$(dirname "$0")"
[ -x mfma_power ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
timeout 120 ./mfma_power
