#!/bin/bash
# SQ counters of every kernel of one update (tools/profile_stages.py --reps 1): MFMA pipe utilisation, shader clock,
# resident waves, share of wave time parked / issue-stalled.  usage (GPU box): bash tools/pmc_stages.sh [profile_stages args]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmc_stages
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/sq -- python3 $R/tools/profile_stages.py --reps 1 "$@" > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
cc = glob.glob(f"{out}/sq/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{out}/sq/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
acc = collections.OrderedDict()
for r in csv.DictReader(open(cc[0])):
    key = (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", r.get("Grid_Size_X", "0")))
    a = acc.setdefault(key, {"n": set(), "c": collections.defaultdict(float), "d": 0.0})
    if r["Dispatch_Id"] not in a["n"]:
        a["n"].add(r["Dispatch_Id"]); a["d"] += dur.get(r["Dispatch_Id"], 0.0)
    a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, grid), a in acc.items():
    n = len(a["n"]); m = {c: v / n for c, v in a["c"].items()}; d = a["d"] / n
    if "fdql" not in k or m.get("SQ_BUSY_CYCLES", 0) == 0 or d < 15: continue
    busy = m["SQ_BUSY_CYCLES"] / 32
    print(f"{k[:44]:44s} grid={int(grid):8d} n={n:3d} dur_us={d:7.1f} mfma_util={m['SQ_VALU_MFMA_BUSY_CYCLES']/1024/busy:.3f} clock_GHz={busy/d/1e3:.2f} "
          f"waves/simd={m['SQ_WAVE_CYCLES']*4/1024/busy:.2f} parked={m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} "
          f"issue_stall={m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f} active={m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
