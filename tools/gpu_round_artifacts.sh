#!/bin/bash
# Produce the round's measured artefacts on the GPU box (run through gpurun), all bounded by timeouts:
#   bench line with cpu_baseline, rocprofv3 --kernel-trace --stats of the SAME bench command, PMC traffic passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/round; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 200 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > /dev/null 2>&1      # warm the MIOpen find-db
timeout 300 python bench.py 2> $OUT/bench.err | tee $OUT/bench_n1.json | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1 )
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats -name "*domain_stats.csv" | head -1) $OUT/domain_stats.csv
grep -h '"metric"' $OUT/stats.log > $OUT/bench_n1_under_rocprof.json
timeout 200 python tools/bench_conv3x3.py 2>/dev/null | grep '^{' > $OUT/conv3x3_bench.json
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/emu_stats -- python $ROOT/bench.py --conv-emu 2 --no-cpu-baseline --no-opt-in > $OUT/emu_stats.log 2>&1 )
cp $(find $OUT/emu_stats -name "*kernel_stats.csv" | head -1) $OUT/emu_kernel_stats.csv
grep -h '"metric"' $OUT/emu_stats.log > $OUT/emu_bench_n1.json
i=0
for ctrs in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc$i -- python $ROOT/tools/kernels_only.py 3 > $OUT/pmc$i.log 2>&1 )
done
python - $OUT <<'PY'
import csv,glob,sys,collections,json
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "anonymous" not in k: continue
        short=k.split("(anonymous namespace)::")[1].split("(")[0]
        if short.startswith("conv3x3_emu_kernel"): short = "conv3x3_emu_kernel_bf16x" + short.split("<")[1].split(",")[3].strip()
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in agg.items()}
for k,d in res.items():
    # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 under-reports wide (16 B/lane) coalesced reads by 2x
    # (MI355X_MICROARCH.md, HBM section) -> both the raw and the doubled-read figure are recorded
    fs, ws = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
    if fs is not None and ws is not None:
        d["hbm_bytes_raw"] = (fs + ws) * 1024
        d["hbm_bytes_read_x2"] = (2 * fs + ws) * 1024
json.dump(res, open(out+"/pmc_summary.json","w"), indent=1)
for k,d in res.items():
    print(k, {c: round(v) for c,v in d.items() if c in ("FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum","TCC_MISS_sum","hbm_bytes_raw","hbm_bytes_read_x2")})
PY
python - $OUT/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows); print("total kernel ms", tot/1e6)
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
