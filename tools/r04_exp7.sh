#!/bin/bash
mkdir -p gpurun_out/t7
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_golden or oracle_config2 or (other_configs and (chain or Chain or config))" > gpurun_out/t7/pytest.txt 2>&1
tail -3 gpurun_out/t7/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t7/stages.txt 2>&1
FDQL_CHAIN_STAMPS=1 python3 tools/chain_stamps.py > gpurun_out/t7/chain_stamps.txt 2>&1
grep -E "chain|update-only|total" gpurun_out/t7/stages.txt; tail -2 gpurun_out/t7/chain_stamps.txt
