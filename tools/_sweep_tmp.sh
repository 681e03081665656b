mkdir -p gpurun_out/r06b
python -m pytest tests -x -q -m gpu > gpurun_out/r06b/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06b/gputest.log; tail -3 gpurun_out/r06b/gputest.log
O=gpurun_out/r06b/sweep.txt
: > $O
for cfg in "--B 256" "--B 128" "--B 64" "--B 32" "--T 2 --B 256" "--obs 376 --act 17 --Q 25 --B 1024 --reps 3" "--obs 376 --act 17 --Q 25 --B 128 --reps 5"; do
  for bm in off 64 32 16; do
    echo "=== $cfg bm=$bm" >> $O
    if [ $bm = off ]; then FDQL_NO_ROWDGRAD_CHAIN=1 python tools/profile_stages.py $cfg 2>&1 | grep -E "dstate|joiner|denc|enc_obs.dpre|rowdchain|wgrad|total|wall" >> $O
    else FDQL_EXP_CHAIN_BM=$bm python tools/profile_stages.py $cfg 2>&1 | grep -E "dstate|joiner|denc|enc_obs.dpre|rowdchain|wgrad|total|wall|rror" >> $O; fi
  done
done
echo sweep done
