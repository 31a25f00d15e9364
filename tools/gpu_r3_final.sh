#!/bin/bash
# round 3, final check of the tree: whole GPU suite, smoke, default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3final; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2> $OUT/bench.err | tee $OUT/bench_n1.json | cut -c1-200
