mkdir -p gpurun_out/r06o
python -m pytest tests -x -q -m gpu > gpurun_out/r06o/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06o/gputest.log; tail -3 gpurun_out/r06o/gputest.log
for cfg in "--B 32" "--obs 376 --act 17 --Q 25 --B 32 --reps 5"; do echo "== $cfg"; python tools/profile_stages.py $cfg 2>&1 | grep -E "critics|wall"; done > gpurun_out/r06o/thr.txt 2>&1
