timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "config2" 2>&1 | tail -3
FDQL_ROWGEMM_FORMS=1 timeout -k 10 120 python tools/profile_stages.py --reps 3 2>&1 | grep "critics.fwd1\|rowgemm\|row-block"
timeout -k 10 120 python tools/profile_stages.py --reps 5 2>&1 | grep "critics\|rowgemm\|row-block\|wall"
timeout -k 5 120 python tools/proto/bench_rows.py 12544 15 1 2>&1 | grep "err"
