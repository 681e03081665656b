#!/bin/bash
# the whole GPU suite, then the headline bench and the other workloads' stage times
mkdir -p gpurun_out/full
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.txt 2>&1
tail -3 gpurun_out/full/pytest.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/full/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('cpu_baseline')); oc=d.get('other_configs',{}); print({k:(v.get('steps_per_s') if isinstance(v,dict) else v) for k,v in oc.items()})"
