#!/bin/bash
# stage times of config 2 at the per-rank batches of a strong-scaled step (global batch 256 windows over 2 / 4 / 8 GPUs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_rank
mkdir -p $OUT
cd $R
for B in 128 64 32; do
  echo "== config 2 per rank, B=$B" | tee -a $OUT/stage_times_config2_per_rank.txt
  timeout -k 10 120 python3 tools/profile_stages.py --B $B --reps 20 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee -a $OUT/stage_times_config2_per_rank.txt
done
