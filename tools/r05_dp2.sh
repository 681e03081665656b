#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== two-bucket plan B=256"; timeout -k 10 120 python3 tools/profile_stages.py --world 2 2>&1 | grep -E "wgrad|total|update-only" | cut -c1-110
for B in 128 64 32; do echo "== single plan B=$B"; timeout -k 10 120 python3 tools/profile_stages.py --B $B --reps 20 2>&1 | grep -E "wgrad|total|update-only" | cut -c1-110; done
for B in 128 64; do echo "== two-bucket plan B=$B"; timeout -k 10 120 python3 tools/profile_stages.py --world 2 --B $B --reps 20 2>&1 | grep -E "wgrad|total|update-only" | cut -c1-110; done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -x -q -k "share or bucket or rank or WGRAD_STAT or distributed" 2>&1 | tail -3
