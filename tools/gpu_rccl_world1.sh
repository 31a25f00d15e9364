#!/bin/bash
# Round 6: first contact with RCCL on a 1-GPU box.  bench.py --force-dist = the multi-rank path with a world of one: the collectives run as self
# exchanges on RCCL between the two graph replays of a lane.  Produces: the bench line, a rocprofv3 kernel trace of the same command, and an excerpt of the trace
# around RCCL's kernels (what ran right before and after them on the GPU).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/rccl; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for mode in ring gather; do
  timeout 300 python bench.py --force-dist --mode $mode --steps 20 --warmup 5 --no-cpu-baseline --no-numerics > $OUT/bench_force_dist_$mode.json 2> $OUT/bench_force_dist_$mode.err
  cut -c1-200 $OUT/bench_force_dist_$mode.json
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-numerics > $OUT/trace.log 2>&1 )
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_force_dist.csv
python - $OUT <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0][:90]
idx = [i for i, r in enumerate(rows) if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
with open(out + "/rccl_kernels_in_the_trace.txt", "w") as w:
    w.write(f"{len(rows)} kernel dispatches in the trace, {len(idx)} of them RCCL kernels: {sorted({short(rows[i]['Kernel_Name']) for i in idx})}\n")
    w.write("columns: start (us since the first dispatch), duration (us), queue, kernel\n")
    t0 = int(rows[0]["Start_Timestamp"])
    # the LAST three-collective group of the timed loop (steady state): 8 dispatches before its first RCCL kernel, through 8 after its last
    if idx:
        last = idx[-1]
        first = last
        while first - 1 in idx or (first - 2 in idx) or (first - 3 in idx):
            first = max(j for j in idx if j < first)
            if last - first > 12: break
        for i in range(max(0, first - 10), min(len(rows), last + 10)):
            r = rows[i]
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            w.write(f"{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:9.1f}  q{r.get('Queue_Id', '?'):>3}  {'>>> ' if i in idx else '    '}{short(r['Kernel_Name'])}\n")
print(open(out + "/rccl_kernels_in_the_trace.txt").read()[:3000])
PY
rm -rf $OUT/trace
