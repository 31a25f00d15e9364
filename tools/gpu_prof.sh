#!/bin/bash
# rocprofv3 kernel stats of the bench command (run through gpurun). Usage: bash tools/gpu_prof.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1  # warm the MIOpen find-db first
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1 )
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms={int(r['TotalDurationNs'])/1e6:7.2f} pct={r['Percentage']}")
PY
tail -1 $OUT/run.log | cut -c1-300
