#!/bin/bash
# round 3, call D: what bounds pillar_rows_mx_kernel -- ablations (stores off, compute off, fewer workgroups) + PMC counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r3d; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
for v in "0 0" "1 0" "2 0" "3 0" "4 0" "7 0" "0 512" "0 256"; do
  set -- $v
  echo "== DEBUG=$1 BLOCKS=$2"
  COALIGN_PILLAR_DEBUG=$1 COALIGN_PILLAR_BLOCKS=$2 bash tools/gpu_prof_kernels.sh r3d_$1_$2 pillar_nhwc_persistent 2>&1 | grep -E "rows_mx|prep|^\{" | tee -a $OUT/ablate.log
done
i=0
for ctrs in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$i -- python $ROOT/tools/kernels_only.py 3 pillar_nhwc_persistent > $OUT/pmc_$i.log 2>&1 )
done
python - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "pillar" not in k: continue
        short="rows" if "rows" in k else "prep"
        tot[short][r["Counter_Name"]]+=float(r["Counter_Value"])/4
for k,c in tot.items():
    print(k, {n: round(v) for n,v in sorted(c.items())})
PY
