#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_lds
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/tools/conv_bench.py 2560 > $OUT/log.txt 2>&1 || { tail $OUT/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05_lds"
f = sorted(glob.glob(out + "/pmc/**/*_counter_collection.csv", recursive=True))[-1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_conv" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"][r["Kernel_Name"].index("k_conv"):][:60]
    d = acc.setdefault(k, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"]); d["n_" + r["Counter_Name"]] += 1
for k, d in acc.items():
    g = lambda c: d[c] / max(d["n_" + c], 1)
    print(f"{k:60s} LDS insts {g('SQ_INSTS_LDS'):12.0f}  bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT'):14.0f}  idx-active {g('SQ_LDS_IDX_ACTIVE'):14.0f}  conflict/active {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):5.2f}  busy {g('SQ_BUSY_CYCLES'):12.0f}")
PY
