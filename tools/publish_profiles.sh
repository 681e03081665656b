#!/bin/bash
# Copies the evidence collect_profiles.sh left under gpurun_out/profiles_new/ into profiles/ under a round prefix:
#   bash tools/publish_profiles.sh r06
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gpurun_out/profiles_new
PRE=${1:?round prefix, e.g. r06}
D=$R/profiles
clean() { grep -v "amdgpu.ids" "$1" > "$2"; }
for f in stage_times stage_times_config4_B1024 stage_times_config4_B128_per_rank stage_times_config2_two_bucket_plan stage_times_config2_per_rank \
         stage_times_gru stage_times_config5 temporal_len_2_stages act_latency conv_kernels; do
  [ -f $P/$f.txt ] && clean $P/$f.txt $D/${PRE}_$f.txt
done
for f in hbm_traffic_pmc mfma_utilisation_pmc valu_per_mfma; do
  [ -f $P/$f.txt ] && cp $P/$f.txt $D/${PRE}_$f.txt
  [ -f $P/c5/$f.txt ] && sed 's#tools/profile_stages.py --reps 1 (config 2, T=50, B=256)#tools/config5_bench.py --ring 20000 --steps 1 (config 5, T=50, B=512)#' $P/c5/$f.txt > $D/${PRE}_config5_$f.txt
done
[ -f $P/dominant_kernel_traffic.json ] && cp $P/dominant_kernel_traffic.json $D/${PRE}_dominant_kernel_traffic.json
[ -f $P/rocprofv3_kernel_stats_bench.csv ] && cp $P/rocprofv3_kernel_stats_bench.csv $D/${PRE}_rocprofv3_kernel_stats_bench.csv
[ -f $P/bench_n1.json ] && grep "^{" $P/bench_n1.json > $D/${PRE}_bench_n1.json
[ -f $P/bench_2rank_gloo_plain_launch.json ] && grep "^{" $P/bench_2rank_gloo_plain_launch.json > $D/${PRE}_bench_2rank_gloo_plain_launch.json
[ -f $P/sampler/sampler_pmc.txt ] && clean $P/sampler/sampler_pmc.txt $D/${PRE}_sampler_pmc.txt
[ -f $P/sampler/sampler_traffic.json ] && cp $P/sampler/sampler_traffic.json $D/${PRE}_sampler_traffic.json
[ -f $R/gpurun_out/parity_report.txt ] && cp $R/gpurun_out/parity_report.txt $D/${PRE}_parity_report.txt
[ -f $R/gpurun_out/parity_three_way.txt ] && cp $R/gpurun_out/parity_three_way.txt $D/${PRE}_parity_three_way.txt
ls $D | grep "^${PRE}_"
