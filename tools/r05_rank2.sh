#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { echo "== B=$1 $2"; env $2 timeout -k 10 120 python3 tools/profile_stages.py --B $1 --reps 20 2>&1 | grep -E "chain|wgrad|total|update-only" | cut -c1-120; }
run 128 FDQL_X=0
run 64 FDQL_X=0
run 64 FDQL_WGRAD_STAT=0
run 64 FDQL_CHAIN=0
run 128 FDQL_WGRAD_STAT=0
run 32 FDQL_CHAIN=enc
