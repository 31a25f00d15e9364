#!/bin/bash
# round 3: all kernels built without packed fp32 instructions (-fno-slp-vectorize).  Interference check of every convolution variant against the
# fusion kernel, soak of the 3-lane graph pipeline with the 6 x 32 stacked variant on, frame rate A/B, isolated kernel times.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/interference; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
echo "== fusion beside every convolution variant (COALIGN_EMU_STACK=3)"; COALIGN_EMU_STACK=3 timeout 300 python tools/diag_fuse_corun.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/corun.log
echo "== soak"; for st in 3; do COALIGN_EMU_STACK=$st FRAMES=3000 timeout 600 python tools/soak_pipeline.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/soak.log; done
for st in 3 1 3 1; do
  COALIGN_EMU_STACK=$st timeout 300 python bench.py --no-cpu-baseline --no-side-modes --no-from-points --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STACK=$st', d['value'], 'frames/s')" | tee -a $OUT/ab.log
done
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso -- python $ROOT/tools/kernels_only.py 20 > $OUT/iso.log 2>&1 )
cp $(find $OUT/iso -name "*kernel_stats.csv" | head -1) $OUT/kernels_isolated_stats.csv
python - $OUT/kernels_isolated_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
