#!/bin/bash
mkdir -p gpurun_out/t8
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -k "rowgemm_forms or oracle_config2 or reference_golden or three_way or split_phase or (other_configs and (stationary or MASKS or PRESUM or 3-layer or config))" > gpurun_out/t8/pytest.txt 2>&1
tail -3 gpurun_out/t8/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t8/stages.txt 2>&1
FDQL_NO_GATE_MASKS=1 python3 tools/profile_stages.py > gpurun_out/t8/stages_old.txt 2>&1
python3 tools/profile_stages.py > gpurun_out/t8/stages2.txt 2>&1
grep -E "critics|update-only|total" gpurun_out/t8/stages.txt gpurun_out/t8/stages_old.txt gpurun_out/t8/stages2.txt
