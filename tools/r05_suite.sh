#!/bin/bash
# whole GPU parity suite + the headline bench line (no extras) on the current sources
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_suite
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout -k 10 120 python3 tools/her_vmap_profile.py > $OUT/her_vmap_profile.txt 2>&1; head -45 $OUT/her_vmap_profile.txt
for B in 32 64; do echo "== config 2 per rank, B=$B"; timeout -k 10 100 python3 tools/profile_stages.py --B $B --reps 20 2>&1 | tail -30; done > $OUT/stage_times_config2_per_rank.txt 2>&1
