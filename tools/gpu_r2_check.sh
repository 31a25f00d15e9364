#!/bin/bash
# Round-2 check on the GPU box: full GPU test suite (all failures listed), then a default bench run.  usage: bash tools/gpu_r2_check.sh <tag> [pytest -k expr]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-a}; KEXPR=${2:-}; OUT=$ROOT/gpurun_out/r2_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -k "$KEXPR" --durations=8 > $OUT/pytest.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1
fi
echo "pytest rc=$?" >> $OUT/pytest.log
grep -v "amdgpu.ids" $OUT/pytest.log | tail -60
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
tail -5 $OUT/bench.err
cut -c1-1500 $OUT/bench.json
