#!/bin/bash
# SQ counters of one dense GEMM launch (tools/gemm_one.py), one rocprofv3 pass per counter group.
# usage (on the GPU box): bash tools/pmc_gemm.sh [M N K]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
if [ -n "$PMC_GROUPS" ]; then IFS=";" read -ra GROUPS_ARR <<< "$PMC_GROUPS"; else GROUPS_ARR=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES"); fi
for grp in "${GROUPS_ARR[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/gemm_one.py "$@" > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_grouped" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(acc):
        line = f"{k:32s} {acc[k][0] / acc[k][1]:16.0f}  (avg of {acc[k][1]} dispatches)"
        print(line); fo.write(line + "\n")
PY
