#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_misc
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 tools/her_vmap_profile2.py > $OUT/her_vmap_profile2.txt 2>&1; head -50 $OUT/her_vmap_profile2.txt
for B in 32 64; do echo "== config 2 per rank, B=$B"; timeout -k 10 100 python3 tools/profile_stages.py --B $B --reps 20 2>&1 | grep -E "dstate|total|update-only"; done
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "rank or three_way" --durations=5 2>&1 | tail -12
timeout -k 10 300 python3 -m pytest tests/test_gpu_facade.py -x -q -k "vmap" 2>&1 | tail -4
