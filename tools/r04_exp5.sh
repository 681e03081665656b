#!/bin/bash
mkdir -p gpurun_out/t5
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "oracle_config2 or reference_golden or (other_configs and (COLSUM or action or policy or config4 or ragged or odd or config))" > gpurun_out/t5/pytest.txt 2>&1
tail -3 gpurun_out/t5/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t5/stages.txt 2>&1
FDQL_NO_COLSUM_STREAM=1 python3 tools/profile_stages.py > gpurun_out/t5/stages_old.txt 2>&1
python3 tools/profile_stages.py --obs 376 --act 17 --Q 25 --B 1024 > gpurun_out/t5/stages_c4.txt 2>&1
grep -E "wgrad|colsum|adam|update-only|total" gpurun_out/t5/stages.txt gpurun_out/t5/stages_old.txt
grep -E "dstate|wgrad|colsum|update-only|total" gpurun_out/t5/stages_c4.txt
