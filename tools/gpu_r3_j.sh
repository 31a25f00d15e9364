#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
for r in 106 206 308 8 12; do echo "== rows $r"; ONLY25=1 COALIGN_EMU_TAPK_ROWS=$r timeout 300 python tools/diag_fuse_corun.py 2>&1 | grep -v amdgpu.ids | tail -2; done
echo "== rows 106, no setprio"; ONLY25=1 COALIGN_EMU_TAPK_ROWS=106 COALIGN_EMU_PRIO=9 timeout 300 python tools/diag_fuse_corun.py 2>&1 | grep -v amdgpu.ids | tail -2
