#!/bin/bash
mkdir -p gpurun_out/t9
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -k "oracle_config2 or reference_golden or three_way or split_phase or (other_configs and (ROWDGRAD or dgrad or FOLD or stationary or config))" > gpurun_out/t9/pytest.txt 2>&1
tail -3 gpurun_out/t9/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t9/stages.txt 2>&1
FDQL_NO_ROWDGRAD_CHAIN=1 python3 tools/profile_stages.py > gpurun_out/t9/stages_old.txt 2>&1
python3 tools/profile_stages.py --world 2 > gpurun_out/t9/stages_w2.txt 2>&1
grep -E "dstate|joiner|denc|enc_obs.dpre|update-only|total" gpurun_out/t9/stages.txt gpurun_out/t9/stages_old.txt gpurun_out/t9/stages_w2.txt
