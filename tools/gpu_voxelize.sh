#!/bin/bash
# GPU-box script: new tests, voxeliser timing, kernel trace.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/vox
timeout 300 python -m pytest tests -m gpu -x -q -k "voxelize or synthetic_inference or evaluation" 2>&1 | tail -15
timeout 120 python tools/bench_voxelize.py > gpurun_out/vox/bench.json 2> gpurun_out/vox/bench.err; cat gpurun_out/vox/bench.json; tail -3 gpurun_out/vox/bench.err
timeout 120 python tools/bench_voxelize.py --shuffle >> gpurun_out/vox/bench.json; tail -1 gpurun_out/vox/bench.json
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vox/prof -o vox -- python tools/bench_voxelize.py > gpurun_out/vox/prof.log 2>&1
f=$(find gpurun_out/vox/prof -name "*kernel_stats.csv" | head -1); head -12 "$f"
