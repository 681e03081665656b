#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_ab
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -x -q > $OUT/pytest_conv.log 2>&1 || { echo "conv tests FAILED"; tail -20 $OUT/pytest_conv.log; exit 1; }
tail -2 $OUT/pytest_conv.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "Atari or pixel" > $OUT/pytest_pix.log 2>&1 || { echo "pixel tests FAILED"; tail -20 $OUT/pytest_pix.log; exit 1; }
tail -2 $OUT/pytest_pix.log
timeout -k 10 200 python3 tools/conv_bench.py > $OUT/conv.txt 2>&1; cat $OUT/conv.txt
