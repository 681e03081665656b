#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh a'     (the headline: bench line, kernel stats, PMC passes)
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh b'     (the other workloads, config 5, the N > 1 rehearsal)
#   gpurun --timeout 900  -- 'bash tools/collect_profiles.sh s'     (the sampler: HBM GB/s + PMC traffic of k_gather_windows)
# Writes into gpurun_out/profiles_new/ (merged back by gpurun); copy what should be judged into profiles/ (round prefix).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
PART=${1:-ab}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT $OUT/c5
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *a* ]]; then
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_insts
# 1. bench line (N=1) and its rocprofv3 kernel stats (same command)
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
# 2. per-stage times (HIP events inside the library)
python3 $R/tools/profile_stages.py > $OUT/stage_times.txt 2>&1 || exit 1
# 3. HBM traffic of every kernel of one update: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_write.log 2>&1 || exit 1
# 4. matrix-pipe utilisation of every GEMM launch of one update (SQ counters, their own pass)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_sq.log 2>&1 || exit 1
# 5. instruction mix of every dense kernel (what the fp32 MFMA stream is shared with): counts in their own pass
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/pmc_insts -- python3 $R/tools/profile_stages.py --reps 1 > $OUT/pmc_insts.log 2>&1 || exit 1
python3 $R/tools/summarize_profiles.py $OUT || exit 1
echo "part a collected"
fi
if [[ $PART == *b* ]]; then
rm -rf $OUT/c5; mkdir -p $OUT/c5
# 6. the other workloads' per-stage times, the small-batch step, act() latency, the GRU joiner step
python3 $R/tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 --reps 3 > $OUT/stage_times_config4_B1024.txt 2>&1 || exit 1
python3 $R/tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 128 --reps 5 > $OUT/stage_times_config4_B128_per_rank.txt 2>&1 || exit 1
python3 $R/tools/profile_stages.py --world 2 > $OUT/stage_times_config2_two_bucket_plan.txt 2>&1 || exit 1
for B in 128 64 32; do echo "== config 2 per rank of a 256-window global batch: B=$B"; python3 $R/tools/profile_stages.py --B $B --reps 10 2>&1 | tail -32; done > $OUT/stage_times_config2_per_rank.txt 2>&1 || exit 1
python3 $R/tools/t2_latency.py stages > $OUT/temporal_len_2_stages.txt 2>&1 || exit 1
python3 $R/tools/act_bench.py 1 8 64 256 > $OUT/act_latency.txt 2>&1 || exit 1
python3 $R/tools/profile_stages.py --gru zero --reps 3 > $OUT/stage_times_gru.txt 2>&1 || exit 1
# 7. config 5 (the implicit-GEMM conv path): stage times, the conv kernels alone, PMC traffic / pipe utilisation / instruction mix
python3 $R/tools/config5_bench.py --ring 100000 --steps 10 > $OUT/stage_times_config5.txt 2>&1 || exit 1
python3 $R/tools/conv_bench.py > $OUT/conv_kernels.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5/pmc_fetch -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5/pmc_write -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/c5/pmc_sq -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5/pmc_sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/c5/pmc_insts -- python3 $R/tools/config5_bench.py --ring 20000 --steps 1 > $OUT/c5/pmc_insts.log 2>&1 || exit 1
python3 $R/tools/summarize_profiles.py $OUT/c5 "tools/config5_bench.py (BASELINE config 5, T=50, B=512, frames read from the uint8 ring in place)" || exit 1
# 8. the N > 1 code path from the plain command line (ranks share this box's one GPU: gloo rehearsal, not a scaling figure)
FDQL_BENCH_BACKEND=gloo FDQL_BENCH_RING=200000 python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2rank_gloo_plain_launch.json 2> $OUT/bench_2rank.err || exit 1
echo "part b collected"
fi
if [[ $PART == *s* ]]; then
# 9. north_star: "evidenced by rocprof HBM GB/s on the sampler".  Three rocprofv3 passes over tools/sampler_bench.py (config 2 / 3 /
#    4 / 5 row sizes): FETCH_SIZE, WRITE_SIZE (their own passes: TCC counter slots) and a kernel trace for the durations
S=$OUT/sampler
rm -rf $S; mkdir -p $S
python3 $R/tools/sampler_bench.py > $S/sampler_hbm.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $S/pmc_fetch -- python3 $R/tools/sampler_bench.py --reps 5 > $S/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $S/pmc_write -- python3 $R/tools/sampler_bench.py --reps 5 > $S/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $S/trace -- python3 $R/tools/sampler_bench.py --reps 5 > $S/trace.log 2>&1 || exit 1
python3 $R/tools/sampler_pmc.py $S > $S/sampler_pmc.txt 2>&1 || exit 1
cat $S/sampler_hbm.txt >> $S/sampler_pmc.txt
echo "part s collected"
fi
