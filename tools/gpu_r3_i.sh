#!/bin/bash
# round 3, call I: the whole GPU suite on the current tree
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3i; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
