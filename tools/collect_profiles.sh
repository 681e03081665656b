#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
# Writes into gpurun_out/profiles_new/ (merged back by gpurun); copy what should be judged into profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_new
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. bench line (N=1) and its rocprofv3 kernel stats (same command)
python3 $R/bench.py --steps 50 --warmup 10 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
# 2. per-stage times (HIP events inside the library)
python3 $R/tools/profile_stages.py > $OUT/stage_times.txt 2>&1 || exit 1
# 3. HBM traffic of every kernel of one update: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_write.log 2>&1 || exit 1
# 4. matrix-pipe utilisation of every GEMM launch of one update (SQ counters, their own pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_sq.log 2>&1 || exit 1
# 5. instruction mix of every dense kernel (what the fp32 MFMA stream is shared with: VALU / LDS / vector-memory instructions
#    take MFMA issue time on gfx950): counts in their own pass
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/pmc_insts -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_insts.log 2>&1 || exit 1
python3 $R/tools/summarize_profiles.py $OUT
