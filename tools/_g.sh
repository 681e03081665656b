FDQL_GEMM_VARIANT=5 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm_forms or config2 or golden" 2>&1 | tail -4
for v in 1 5; do echo "== variant $v"; FDQL_GEMM_VARIANT=$v timeout -k 10 120 python tools/profile_stages.py --reps 5 2>&1 | grep "gemm64x64\|wall"; done
