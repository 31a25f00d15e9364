#!/bin/bash
# Winograd vs direct kernel: clock / power / energy per call, then PMC passes of the Winograd kernel on the 25 x 88 shape (separate passes).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/wino_probe; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for shape in 5,256,25,88 5,128,50,176 5,64,100,352; do
  for k in direct wino; do SHAPE=$shape KERNEL=$k timeout 60 python tools/wino_probe.py power | python -c "import json,sys; d=json.loads(sys.stdin.read()); d.pop('sclk_MHz'); d.pop('power_W'); print(d)" ; done
done | tee $OUT/power.txt
cd /tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  for k in wino direct; do
    SHAPE=5,256,25,88 KERNEL=$k timeout 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p${i}_$k -- python $ROOT/tools/wino_probe.py few 4 > $OUT/p${i}_$k.log 2>&1
  done
done
python - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv3x3" not in k: continue
        short=("wino" if "wino" in k else "direct")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} mean={sum(vals)/len(vals):16.1f} n={len(vals)}")
PY
