#!/bin/bash
# Kernels of ONE steady-state frame: the bench under rocprofv3 --kernel-trace --stats at two step counts; the difference of the per-kernel call counts / total
# times divided by the extra steps is what a frame launches once every graph is captured (set-up, capture and microbenchmark launches cancel).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/steady; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
A=${STEPS_A:-40}; B=${STEPS_B:-140}
for S in $A $B; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$S -- python $ROOT/bench.py --steps $S ${BENCH_ARGS:-} --no-cpu-baseline --no-numerics --no-side-modes --no-size-sweep --no-from-points --no-latency > $OUT/s$S.log 2>&1 )
done
python - $OUT $A $B <<'PY'
import csv, glob, sys, json
out, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
def load(s):
    f = glob.glob(f"{out}/s{s}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(A), load(B)
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    if cb != ca:
        rows.append((k, (cb - ca) / (B - A), (tb - ta) / max(1, cb - ca) / 1e3, (tb - ta) / (B - A) / 1e3))
rows.sort(key=lambda r: -r[3])
with open(out + "/steady_state_kernels_per_frame.csv", "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches_per_frame", "avg_us", "us_per_frame"])
    for r in rows: w.writerow([r[0], round(r[1], 3), round(r[2], 2), round(r[3], 1)])
lib = [r for r in rows if r[0].startswith("void at::") or r[0].startswith("Cijk_") or "rocprim" in r[0] or "miopen" in r[0].lower()]
print(f"{len(rows)} kernels differ between {A} and {B} steps; {sum(r[1] for r in rows):.1f} launches and {sum(r[3] for r in rows):.0f} us of kernel time per frame (kernels of three frames in flight overlap)")
for r in rows[:40]: print(f"  {r[1]:7.2f} x {r[2]:8.1f} us = {r[3]:8.1f} us/frame  {r[0][:130]}")
print("library kernels (at::native / rocBLAS / rocPRIM / MIOpen) per steady-state frame:", [(r[0][:90], round(r[1], 2)) for r in lib] or "none")
json.dump({"steps": [A, B], "library_kernels_per_frame": [(r[0], r[1]) for r in lib], "launches_per_frame": sum(r[1] for r in rows)}, open(out + "/steady_state_summary.json", "w"), indent=1)
PY
rm -rf $OUT/s$A $OUT/s$B      # (the raw traces are bulky; the per-frame table is what is kept)
