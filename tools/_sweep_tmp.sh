mkdir -p gpurun_out/r06g
python -m pytest tests -x -q -m gpu > gpurun_out/r06g/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06g/gputest.log; tail -3 gpurun_out/r06g/gputest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err; echo bench rc $?
