#!/bin/bash
# PMC counter passes over the hand-written kernels only (separate passes; never combined with tracing domains).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
python $ROOT/tools/kernels_only.py 20
i=0
for ctrs in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python $ROOT/tools/kernels_only.py 3 > $OUT/p$i.log 2>&1
done
python - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "anonymous" not in k: continue
        short=k.split("::")[1].split("(")[0][:40]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} mean={sum(vals)/len(vals):14.1f} n={len(vals)}")
PY
