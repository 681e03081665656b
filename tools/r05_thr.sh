#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for B in 128 64 32; do echo "== B=$B"; timeout -k 10 120 python3 tools/profile_stages.py --B $B --reps 20 2>&1 | grep -E "dpi|total|update-only" | cut -c1-110; done
echo "== T=2"; timeout -k 10 120 python3 tools/t2_latency.py 2>&1 | tail -1
