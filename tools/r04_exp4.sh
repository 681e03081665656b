#!/bin/bash
mkdir -p gpurun_out/t4
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "oracle_config2 or reference_golden or (other_configs and (COLSUM or action or policy or temporal_len or ragged or odd))" > gpurun_out/t4/pytest.txt 2>&1
tail -3 gpurun_out/t4/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t4/stages.txt 2>&1
FDQL_NO_COLSUM_STREAM=1 python3 tools/profile_stages.py > gpurun_out/t4/stages_old.txt 2>&1
python3 tools/profile_stages.py --world 2 > gpurun_out/t4/stages_w2.txt 2>&1
grep -E "wgrad|colsum|adam|update-only|total" gpurun_out/t4/stages.txt gpurun_out/t4/stages_old.txt gpurun_out/t4/stages_w2.txt
