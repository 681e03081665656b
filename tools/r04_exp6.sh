#!/bin/bash
mkdir -p gpurun_out/t6
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_golden or oracle_config2 or act_ or three_way or (other_configs and (discrete or GRU or gru or config4 or policy or ragged))" > gpurun_out/t6/pytest.txt 2>&1
tail -3 gpurun_out/t6/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t6/stages.txt 2>&1
python3 tools/profile_stages.py --T 2 > gpurun_out/t6/stages_t2.txt 2>&1
python3 tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 > gpurun_out/t6/stages_c4.txt 2>&1
grep -E "policy|update-only|total" gpurun_out/t6/stages.txt gpurun_out/t6/stages_t2.txt gpurun_out/t6/stages_c4.txt
