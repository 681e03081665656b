#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_conv
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_conv.py -x -q > $OUT/pytest_conv.log 2>&1; echo "conv tests rc $?"; tail -15 $OUT/pytest_conv.log
timeout -k 10 300 python3 tools/conv_bench.py > $OUT/conv_bench.txt 2>&1; echo "bench rc $?"; cat $OUT/conv_bench.txt
