#!/bin/bash
# whole GPU parity suite on the current sources
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_suite
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x --durations=40 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -18 $OUT/pytest_gpu.log | cut -c1-200
