#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "== T=2 $1"; env $1 timeout -k 10 120 python3 tools/t2_latency.py stages 2>&1 | grep -v amdgpu.ids | tail -28; }
run FDQL_X=0
run "FDQL_NO_DUAL=1 FDQL_NO_HEAD_FUSE=1"
run FDQL_NO_HEAD_FUSE=1
run FDQL_NO_DUAL=1
echo "== B=32"; timeout -k 10 120 python3 tools/profile_stages.py --B 32 --reps 20 2>&1 | grep -E "total|update-only|small"
