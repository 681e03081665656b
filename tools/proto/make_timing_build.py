"""Builds build_ab/libfdql_timing.so: the product library with s_memtime stamps at five points of the register-staged GEMM
main loop (loop top / next chunk located / operands requested / MFMA k-steps done / LDS stores issued / barrier passed),
per-wave sums written to a device table and read back by fdql_debug_phase_times().  Used by phase_times.py; the
instrumentation costs ~15-25 % and is applied as text patches so that the product source carries none of it."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fastdeepqlearning_amd", "csrc")
s = open(os.path.join(CSRC, "gemm.hip")).read()

def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new, 1)

rep("template <int SHAPE> struct TileCfg;",
    "constexpr int PH_SLOTS = 1 << 18;\n__device__ unsigned long long g_phase_w[PH_SLOTS][6];\ntemplate <int SHAPE> struct TileCfg;")
rep("  while (have) {\n    // locate the next chunk\n",
    "  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, nit = 0, ph0a = 0;\n  while (have) {\n"
    "    const unsigned long long t0 = __builtin_readcyclecounter();\n    // locate the next chunk\n")
rep("    const bool has_next = (ns < nseg_main) && (ksplit == 1 || ns == 0);\n    if (has_next) {\n      if (ns != s) fetch_seg(ns);",
    "    const bool has_next = (ns < nseg_main) && (ksplit == 1 || ns == 0);\n    const unsigned long long t0a = __builtin_readcyclecounter();\n"
    "    if (has_next) {\n      if (ns != s) fetch_seg(ns);")
rep("    run_chunk();\n\n    if (!has_next) break;",
    "    const unsigned long long t1 = __builtin_readcyclecounter();\n    run_chunk();\n"
    "    const unsigned long long t2 = __builtin_readcyclecounter();\n\n    if (!has_next) break;")
rep("    __syncthreads();\n    cur ^= 1;\n    s = ns; k = nk; ke = nke;\n  }\n",
    "    const unsigned long long t3 = __builtin_readcyclecounter();\n    __syncthreads();\n"
    "    const unsigned long long t4 = __builtin_readcyclecounter();\n"
    "    ph0a += t0a - t0; ph0 += t1 - t0; ph1 += t2 - t1; ph2 += t3 - t2; ph3 += t4 - t3; nit += 1;\n"
    "    cur ^= 1;\n    s = ns; k = nk; ke = nke;\n  }\n"
    "  if ((tid & 63) == 0) {\n    const int slot = (int)((blockIdx.x * 4u + (tid >> 6)) & (PH_SLOTS - 1));\n"
    "    g_phase_w[slot][0] = ph0; g_phase_w[slot][1] = ph1; g_phase_w[slot][2] = ph2; g_phase_w[slot][3] = ph3;\n"
    "    g_phase_w[slot][4] = nit; g_phase_w[slot][5] = ph0a;\n  }\n")
rep("}  // namespace fdql",
    "}  // namespace fdql\nextern \"C\" int fdql_debug_phase_times(unsigned long long *out, int nslots) {\n"
    "  static unsigned long long host[fdql::PH_SLOTS][6];\n"
    "  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fdql::g_phase_w), sizeof(host)) != hipSuccess) return -1;\n"
    "  for (int j = 0; j < 6; ++j) out[j] = 0;\n"
    "  for (int i = 0; i < nslots && i < fdql::PH_SLOTS; ++i)\n    for (int j = 0; j < 6; ++j) out[j] += host[i][j];\n  return 0;\n}")
tmp = os.path.join(CSRC, "gemm_timing_tmp.hip")
open(tmp, "w").write(s)
os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
try:
    subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-Wno-unused-result", "-c", tmp, "-o", "/tmp/gemm_timing.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "/tmp/gemm_timing.o"] +
                          [os.path.join(CSRC, o) for o in ("kernels.o", "agent.o", "ring.o")] +
                          ["-o", os.path.join(ROOT, "build_ab", "libfdql_timing.so")])
finally:
    os.remove(tmp)
print("built build_ab/libfdql_timing.so")
