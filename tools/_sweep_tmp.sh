mkdir -p gpurun_out/r06k
python -m pytest tests -x -q -m gpu > gpurun_out/r06k/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06k/gputest.log; tail -3 gpurun_out/r06k/gputest.log
for cfg in "--B 256" "--B 128" "--B 64" "--B 32" "--T 2 --B 256" "--obs 376 --act 17 --Q 25 --B 32 --reps 5" "--obs 376 --act 17 --Q 25 --B 128 --reps 5"; do echo "== $cfg"; python tools/profile_stages.py $cfg 2>&1 | grep -E "fwd3|wall"; done > gpurun_out/r06k/fwd3.txt 2>&1
