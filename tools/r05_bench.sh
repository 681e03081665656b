#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_bench
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc $?"
tail -3 $OUT/bench_n1.err
python3 - <<'PY'
import json
for line in open("gpurun_out/r05_bench/bench_n1.json"):
    if line.startswith("{"):
        d = json.loads(line)
        print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"][:40])
        for k, v in (d.get("other_configs") or {}).items():
            print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "error", "workspace_GiB", "her_vmap_ingest_records_per_s", "her_vmap_ingest_records_per_s_list_of_records", "her_vmap_ingest_records_per_s_per_record_add", "N2_B128_ms", "N4_B64_ms", "N8_B32_ms", "N2_B512_ms", "N4_B256_ms", "N8_B128_ms", "N8_speedup_bound")})
        print("t2", d.get("also_temporal_len_2"))
        print("facade", d.get("facade_path"))
        print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
