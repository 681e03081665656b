#!/bin/bash
# final pass on frozen sources: the whole GPU suite (parity reports), then the bench line with the committed PMC traffic
mkdir -p gpurun_out/final
rm -f gpurun_out/parity_report.txt gpurun_out/parity_three_way.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.txt 2>&1
tail -3 gpurun_out/final/pytest.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/final/bench_n1.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r.get('traffic'), r.get('stale_profile'))"
