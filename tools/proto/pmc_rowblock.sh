#!/bin/bash
# SQ counters of the row-block prototype variants: one rocprofv3 pass per variant.
# usage (GPU box): bash tools/proto/pmc_rowblock.sh M K variant...
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/pmc_rb
rm -rf $OUT && mkdir -p $OUT
M=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/$v -- python3 $R/tools/proto/bench_rowblock.py $M $K $v > $OUT/$v.log 2>&1 || { tail -5 $OUT/$v.log; exit 1; }
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    cc = glob.glob(f"{out}/{v}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{out}/{v}/**/*kernel_trace.csv", recursive=True)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(cc[0])):
        a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, c in acc.items():
        if "rowblock" not in k and "gemm" not in k: continue
        m = {n: x[0] / x[1] for n, x in c.items()}
        d = sum(dur[k]) / len(dur[k])
        busy = m["SQ_BUSY_CYCLES"] / 32
        print(f"{v:12s} {k[:40]:40s} dur_us={d:7.1f} mfma_util={m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/busy:.3f} clock_GHz={busy/d/1e3:.2f} "
              f"waves/simd={m['SQ_WAVE_CYCLES']*4/1024/busy:.2f} parked={m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} "
              f"issue_stall={m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} active={m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
