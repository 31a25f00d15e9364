#!/bin/bash
# Produce the round's measured artefacts on the GPU box (run through gpurun), all bounded by timeouts:
#   bench line with cpu_baseline, rocprofv3 --kernel-trace --stats of the SAME bench command, isolated kernel stats,
#   PMC passes per op (separate passes; never combined with tracing domains), CPU-oracle thread sweep.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/round; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 600 python bench.py 2> $OUT/bench.err | tee $OUT/bench_n1.json | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline --no-numerics --no-side-modes --no-size-sweep --no-from-points --no-latency > $OUT/stats.log 2>&1 )
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats -name "*domain_stats.csv" | head -1) $OUT/domain_stats.csv
grep -h '"metric"' $OUT/stats.log > $OUT/bench_n1_under_rocprof.json
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iso -- python $ROOT/tools/kernels_only.py 20 > $OUT/iso.log 2>&1 )
cp $(find $OUT/iso -name "*kernel_stats.csv" | head -1) $OUT/kernels_isolated_stats.csv
grep -h '^{' $OUT/iso.log > $OUT/kernels_isolated_events.json
for op in ${PMC_OPS:-pillar_sparse fuse_nhwc_3scales conv_sp_64ch conv_sp_128ch conv_sp_256ch conv_sp_shrink2_256ch_100x352 conv_sp_shrink1_384ch_100x352 conv_sp_s2_64to128 conv_sp_s2_128to256 conv_sp_s2_sparse_canvas_with_row_pack}; do
  i=0
  for ctrs in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA"; do
    i=$((i+1))
    ( cd /tmp && timeout 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_${op}_$i -- python $ROOT/tools/kernels_only.py 3 $op > $OUT/pmc_${op}_$i.log 2>&1 )
  done
done
python - $OUT <<'PY'
import csv,glob,sys,collections,json,os
out=sys.argv[1]
CALLS=4                      # kernels_only.py 3 -> every op runs iters + 1 = 4 times
res={}
for d in sorted(glob.glob(out+"/pmc_*_1")):
    op=os.path.basename(d)[4:-2]
    per_kernel=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out+f"/pmc_{op}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            # kernels of the harness, not of the op: torch helpers, runtime blits / fills of the set-up, rocBLAS of the model construction, one-off folds
            if k.startswith("void at::") or "elementwise" in k or "CatArray" in k or k.startswith("__amd_rocclr") or k.startswith("Cijk_") or "normalize_affine" in k or "pillar_fold" in k or "sp_pack" in k or "fill_" in k: continue
            short=k.split("(anonymous namespace)::")[1].split("(")[0] if "anonymous" in k else k.split("(")[0]
            per_kernel[short][r["Counter_Name"]] += float(r["Counter_Value"])
    op_tot=collections.defaultdict(float)
    kern={}
    for k,c in per_kernel.items():
        kern[k]={n: v/CALLS for n,v in c.items()}
        for n,v in c.items(): op_tot[n]+=v/CALLS
    # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B
    # (MI355X_MICROARCH.md, "HBM"): reads are doubled, writes taken as reported
    fs, ws = op_tot.get("FETCH_SIZE",0.0), op_tot.get("WRITE_SIZE",0.0)
    if op_tot.get("SQ_VALU_MFMA_BUSY_CYCLES") and op_tot.get("GRBM_GUI_ACTIVE"):      # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: share = busy / (1024 * active / 8)
        op_tot["mfma_busy_share"] = op_tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * op_tot["GRBM_GUI_ACTIVE"])
    res[op]={"per_op_call": dict(op_tot), "kernels": kern, "hbm_bytes_raw": (fs+ws)*1024, "hbm_bytes_read_x2": (2*fs+ws)*1024,
             "hbm_read_bytes_x2": 2*fs*1024, "hbm_write_bytes": ws*1024}
json.dump(res, open(out+"/pmc_summary.json","w"), indent=1)
for op,d in res.items():
    print(op, {k: round(v) for k,v in d.items() if k.startswith("hbm")}, "L2 hit", round(d["per_op_call"].get("TCC_HIT_sum",0)/max(1,d["per_op_call"].get("TCC_HIT_sum",0)+d["per_op_call"].get("TCC_MISS_sum",0)),3))
PY
if [ -z "${SKIP_SWEEP:-}" ]; then
  timeout 900 python tools/cpu_baseline_sweep.py > $OUT/cpu_baseline_sweep.json 2> $OUT/cpu_sweep.err
  cat $OUT/cpu_baseline_sweep.json | cut -c1-600
fi
python - $OUT/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows); print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
# the raw traces and counter dumps are bulky (gpurun merges at most 64 MiB back): the summaries above are what is kept
rm -rf $OUT/stats $OUT/iso; for d in $OUT/pmc_*; do [ -d "$d" ] && rm -rf "$d"; done
