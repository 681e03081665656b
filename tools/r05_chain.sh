#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_chain
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "rowdgrad or ROWDGRAD or full or config4 or Humanoid" > $OUT/pytest_chain.log 2>&1 || { echo "tests FAILED"; tail -30 $OUT/pytest_chain.log; exit 1; }
tail -2 $OUT/pytest_chain.log
for w in 8 4; do
  echo "== config 2, FDQL_ROWCHAIN_WAVES=$w"; FDQL_ROWCHAIN_WAVES=$w timeout -k 10 120 python3 tools/profile_stages.py --reps 20 2>&1 | grep -E "rowdchain|update-only|total"
  echo "== config 4 B=1024, FDQL_ROWCHAIN_WAVES=$w"; FDQL_ROWCHAIN_WAVES=$w timeout -k 10 120 python3 tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 --reps 5 2>&1 | grep -E "rowdchain|update-only|total"
done
