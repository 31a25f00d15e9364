#!/bin/bash
# round 3, call C: new tests, full GPU suite, NMS + pillar kernel times (new / legacy), bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3c; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1200 python -m pytest tests/test_round3_gpu.py -m gpu -x -q 2>&1 | tail -30 | tee $OUT/pytest_r3.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_round3_gpu.py 2>&1 | tail -8 | tee $OUT/pytest.log
bash tools/gpu_prof_kernels.sh r3c pillar_nhwc_persistent nms_gather_K600 2>&1 | tee $OUT/kprof.log
COALIGN_NMS_LEGACY=1 bash tools/gpu_prof_kernels.sh r3c_legacy nms_then_gather_K600 2>&1 | tee $OUT/kprof_legacy.log
timeout 600 python bench.py --no-cpu-baseline --no-side-modes 2> $OUT/bench.err | tee $OUT/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step')})
print('latency', d.get('latency_ms'))
print('from_points', {k:v for k,v in (d.get('from_points') or {}).items() if k!='input'})
n=d['north_star_hbm']; print('north', {k:n[k] for k in n if k in ('frac','achieved','bytes_moved','bytes_source','ms','frac_fresh_canvas','frac_kernel_trace')})
print('iso', d['roofline']['isolated_ms'])
for k in d['kernels']: print(k['name'], k['avg_ms'])
"
tail -5 $OUT/bench.err
