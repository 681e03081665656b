#!/bin/bash
# N > 1 code path on the one-GPU box (gloo rehearsal, ranks share the card) + the new per-rank oracle cases
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_dp
mkdir -p $OUT
cd $R
FDQL_BENCH_BACKEND=gloo FDQL_BENCH_RING=200000 timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo_plain_launch.json 2> $OUT/bench_2rank.err; echo "bench rc $?"
tail -3 $OUT/bench_2rank.err
python3 - <<'PY'
import json
for line in open("gpurun_out/r05_dp/bench_2rank_gloo_plain_launch.json"):
    if line.startswith("{"):
        d = json.loads(line)
        print("value", d["value"], d["scaling"], d["ms_per_step"], "plan", d.get("dp_plan"))
        print("weak", d.get("config2_weak"))
        print("c4", d.get("config4_strong"))
PY
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "share" 2>&1 | tail -4
