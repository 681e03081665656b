#!/bin/bash
# north_star: "evidenced by rocprof HBM GB/s on the sampler".  Three rocprofv3 passes over tools/sampler_bench.py (config 2 / 3 /
# 4 / 5 row sizes): FETCH_SIZE, WRITE_SIZE (their own passes: TCC counter slots) and a kernel trace for the durations.
#   gpurun -- 'bash tools/r05_sampler_pmc.sh'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05_sampler
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/sampler_bench.py > $OUT/sampler_hbm.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/sampler_bench.py --reps 5 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/sampler_bench.py --reps 5 > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/sampler_bench.py --reps 5 > $OUT/trace.log 2>&1 || exit 1
python3 $R/tools/sampler_pmc.py $OUT > $OUT/sampler_pmc.txt 2>&1 || exit 1
cat $OUT/sampler_hbm.txt >> $OUT/sampler_pmc.txt
