#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_c5
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pixel or Atari or rowdgrad_chain or 25-quantile" > $OUT/pytest_sel.log 2>&1; echo "sel tests rc $?"; tail -12 $OUT/pytest_sel.log
timeout -k 10 400 python3 tools/config5_bench.py --ring 100000 --steps 10 > $OUT/stage_times_config5_implicit.txt 2>&1; echo "c5 rc $?"; cat $OUT/stage_times_config5_implicit.txt
