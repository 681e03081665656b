#!/bin/bash
mkdir -p gpurun_out/t10
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -k "three_way or (other_configs and (config4 or quantiles or MASKED or 25))" > gpurun_out/t10/pytest.txt 2>&1
tail -3 gpurun_out/t10/pytest.txt
python3 tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 --reps 3 > gpurun_out/t10/c4.txt 2>&1
FDQL_NO_HEAD_DGRAD_MASKED=1 python3 tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 --reps 3 > gpurun_out/t10/c4_old.txt 2>&1
grep -E "critics.dpre|update-only|total" gpurun_out/t10/c4.txt gpurun_out/t10/c4_old.txt
