#!/bin/bash
# Run on the GPU box through gpurun: bench line, rocprofv3 kernel stats of the same command, CPU-baseline leg.
# Everything is bounded by its own timeout; outputs go to gpurun_out/ (copy the summaries to profiles/ afterwards).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== bench (no cpu baseline)"; timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-numerics 2>$OUT/bench1.err | tee $OUT/bench1.json | cut -c1-1500
echo "== rocprofv3 kernel-trace stats"
export TMPDIR=/tmp
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-numerics > $OUT/prof_stats.log 2>&1 )
ls -R $OUT/prof_stats | head -20
f=$(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
echo "== bench with cpu baseline"; timeout 400 python bench.py --steps 10 --warmup 3 --cpu-frames 2 --cpu-budget-s 40 2>$OUT/bench2.err | tee $OUT/bench2.json | cut -c1-600
tail -3 $OUT/bench1.err $OUT/bench2.err
