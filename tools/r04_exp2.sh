#!/bin/bash
# timing experiments on k_wstat_grad (libraries built with -DWS_EXP=n: results are wrong, only the stage times are read).
# Build the variants first (here, no GPU needed), e.g. for n in 1 2 4 8 16 32 63:
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DWS_EXP=$n -c csrc/wstat.hip -o /tmp/w$n.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC csrc/{gemm,smallgemm,gruscan,chain,rowgemm,rowdgrad,wgrad,kernels,agent,ring}.o /tmp/w$n.o -o fastdeepqlearning_amd/exp/libfdql_e$n.so
# (the .so files travel with the snapshot; FDQL_LIB_PATH picks one)
mkdir -p gpurun_out/exp2
python3 tools/profile_stages.py > gpurun_out/exp2/base.txt 2>&1
for e in 64 68; do
  FDQL_LIB_PATH=$PWD/fastdeepqlearning_amd/exp/libfdql_e$e.so timeout -k 10 120 python3 tools/profile_stages.py > gpurun_out/exp2/e$e.txt 2>&1
done
python3 tools/profile_stages.py > gpurun_out/exp2/base2.txt 2>&1
grep -H "dpre1\|update-only" gpurun_out/exp2/base.txt gpurun_out/exp2/e64.txt gpurun_out/exp2/e68.txt gpurun_out/exp2/base2.txt
