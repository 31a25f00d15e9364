#!/bin/bash
# round 3: after the 6 x 32 stacked variant was switched off -- pipeline bisect aid, the whole GPU suite, smoke, and one look at which scale the
# withdrawn variant disturbs
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== rows 106 per scale"; ONLY25=1 COALIGN_EMU_TAPK_ROWS=106 timeout 300 python tools/diag_fuse_corun.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== pipeline"; timeout 600 python tools/diag_pipeline.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_k.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
