#!/bin/bash
# Round 5, first GPU call: the parity suite on the sources as they stand, config 5's per-stage times and PMC traffic BEFORE the
# implicit-GEMM conv path, the sampler's PMC evidence (item 5).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_first
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/config5_bench.py --ring 50000 --steps 10 > $OUT/stage_times_config5_before.txt 2>&1 || { tail -5 $OUT/stage_times_config5_before.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5_write.log 2>&1 || exit 1
bash $R/tools/r05_sampler_pmc.sh || exit 1
echo done
