mkdir -p gpurun_out/r06d
python -m pytest tests -x -q -m gpu > gpurun_out/r06d/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06d/gputest.log; tail -3 gpurun_out/r06d/gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err; echo bench rc $?
