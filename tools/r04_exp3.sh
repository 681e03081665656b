#!/bin/bash
# parity of the fused policy-backward launch and the in-kernel head plane sum, then the stage times
mkdir -p gpurun_out/t3
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "rowgemm_forms or oracle_config2 or reference_golden or (other_configs and (policy or PRESUM or stationary or temporal_len or 8-quantiles or ragged))" > gpurun_out/t3/pytest.txt 2>&1
tail -3 gpurun_out/t3/pytest.txt
python3 tools/profile_stages.py > gpurun_out/t3/stages.txt 2>&1
python3 tools/profile_stages.py --T 2 > gpurun_out/t3/stages_t2.txt 2>&1
FDQL_NO_HEAD_PRESUM=1 FDQL_NO_POLICY_DPRE_FUSE=1 python3 tools/profile_stages.py > gpurun_out/t3/stages_old.txt 2>&1
grep -E "policy|actor.dpre|critics.fwd|critics.head|update-only|total" gpurun_out/t3/stages.txt gpurun_out/t3/stages_t2.txt gpurun_out/t3/stages_old.txt
