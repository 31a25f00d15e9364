#!/bin/bash
set -e
cd "$(dirname "$0")"
[ -x lds_read_rate ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 lds_read_rate.hip -o lds_read_rate
timeout 60 ./lds_read_rate
