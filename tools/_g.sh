for T in 50 2; do for G in 0 1; do
echo "== T=$T FDQL_GRAPH=$G"
FDQL_GRAPH=$G timeout -k 10 120 python tools/profile_stages.py --T $T --reps 5 2>&1 | grep -E "adam|^total|wall"
done; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "graph_replay or golden or config2" 2>&1 | tail -3
