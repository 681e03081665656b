#!/bin/bash
# round-4 experiment batch 1: temporal_len 2 under chain variants; stand-in collective (hold mode) with / without CU reserve
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
for v in "" "FDQL_CHAIN_MIN_BLOCKS=1 FDQL_CHAIN_BM=32" "FDQL_CHAIN=all FDQL_CHAIN_BM=32" "FDQL_CHAIN=all FDQL_CHAIN_BM=32 FDQL_GRAPH=1" "FDQL_CHAIN_MIN_BLOCKS=1 FDQL_CHAIN_BM=64"; do
  echo "== $v" >> $O/t2_variants.txt
  env $v timeout -k 10 120 python3 tools/t2_latency.py stages >> $O/t2_variants.txt 2>&1 || exit 1
done
timeout -k 10 300 python3 tools/dp_overlap.py --steps 150 > $O/dp_overlap.txt 2>&1 || exit 1
FDQL_CU_RESERVE=16 timeout -k 10 300 python3 tools/dp_overlap.py --steps 150 > $O/dp_overlap_reserve16.txt 2>&1 || exit 1
FDQL_CU_RESERVE=32 timeout -k 10 300 python3 tools/dp_overlap.py --steps 150 > $O/dp_overlap_reserve32.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/dp_overlap.py --trace-run > $O/trace.log 2>&1 || exit 1
python3 $R/tools/dp_overlap.py --summarize $O/trace > $O/dp_overlap_trace.txt 2>&1
rm -rf $O/trace
