// Prototype (not product code): rowblock.hip made PERSISTENT with TWO workgroups per CU in anti-phase.
// C[M][256] = lrelu(A[M][K] * W[256][K]^T + b), K = 256.  Each 256-thread workgroup owns a 64-row block at a time
// (activation block in LDS, weights global -> registers as MFMA B fragments) and walks blocks blockIdx.x, + gridDim.x, ...
// Two workgroups share a CU (2 x 66.5 KB LDS, <= 256 registers per lane); the one that finds itself second on its CU
// starts half a block period late, so that one workgroup's load / store phase runs under the other's MFMA phase.
// -DSTAGGER=0: no delay; -DMINB=1: one workgroup per CU (the persistent loop alone).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;
typedef const __attribute__((address_space(1))) f32x4v *gcf4;
typedef __attribute__((address_space(1))) float *gf;
#ifndef MINB
#define MINB 2
#endif
#ifndef STAGGER
#define STAGGER 1
#endif
#ifndef PACKED
#define PACKED 0
#endif
#ifndef SLEEPS
#define SLEEPS 3
#endif
constexpr int K = 256, BM = 64, P = K + 4;

__device__ __forceinline__ unsigned lds_off(const float *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p; }
template <int OFF>
__device__ __forceinline__ void rd128(f32x4v &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, MINB) void k_rowblock2(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias,
                                                          float *__restrict__ C, int M, unsigned *__restrict__ dbg) {
  __shared__ __attribute__((aligned(16))) float lds[BM * P];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const unsigned lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);    // HW_REG_LDS_ALLOC
  if (dbg && tid == 0) {
    dbg[3 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    dbg[3 * blockIdx.x + 1] = lds_alloc;
    dbg[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  }
#if STAGGER
  if ((lds_alloc & 0xfff) != 0) {
#pragma unroll 1
    for (int i = 0; i < SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const int n0 = wave * 64;
  gcf4 wp[2];
#if PACKED   // W pre-packed in fragment order [wave][tn][g][j][lane][c]: every wave-load is 1 KiB contiguous
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) wp[tn] = (gcf4)(W) + (size_t)((wave * 2 + tn) * (K / 32)) * 4 * 64 + lane;
#else
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) wp[tn] = (gcf4)(W + (size_t)(n0 + 32 * tn + li) * K + 16 * lh);
#endif
  float bv[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) bv[tn] = bias ? bias[n0 + 32 * tn + li] : 0.f;
  const unsigned abase = lds_off(lds) + (unsigned)(li * P + 16 * lh) * 4;
  constexpr int G = K / 32;
  const int nblk = M / BM;
  unsigned long long t_load = 0, t_k = 0, t_store = 0, nb = 0;
  const unsigned long long c_begin = __builtin_amdgcn_s_memtime(), w_begin = wall_clock64();
#pragma unroll 1
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int r0 = blk * BM;
    const unsigned long long s0 = __builtin_amdgcn_s_memtime();
    for (int r = wave; r < BM; r += 4)
      __builtin_amdgcn_global_load_lds((glb_vp)(A + (size_t)(r0 + r) * K + lane * 4), (lds_vp)(lds + r * P), 16, 0, 0);
    f32x4v b[2][2][4];   // [buffer][tn][j]
    auto load_b = [&](int buf, int g) __attribute__((always_inline)) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
#if PACKED
        for (int j = 0; j < 4; ++j) b[buf][tn][j] = wp[tn][(4 * g + j) * 64];
#else
        for (int j = 0; j < 4; ++j) b[buf][tn][j] = wp[tn][8 * g + j];
#endif
    };
    load_b(0, 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    const unsigned long long s1 = __builtin_amdgcn_s_memtime();
    auto group = [&](int buf, int g) __attribute__((always_inline)) {
      const unsigned ag = abase + (unsigned)g * 128, ag1 = ag + 32 * P * 4;
      f32x4v a[2][2];
#ifdef NOA
      a[0][0] = a[0][1] = a[1][0] = a[1][1] = f32x4v{1.f, 2.f, 3.f, 4.f};
      asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]));
#else
      rd128<0>(a[0][0], ag);
      rd128<0>(a[0][1], ag1);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#ifdef NOA
        if (false) {
#else
        if (j < 3) {
#endif
          if (j == 0) { rd128<16>(a[1][0], ag); rd128<16>(a[1][1], ag1); }
          if (j == 1) { rd128<32>(a[0][0], ag); rd128<32>(a[0][1], ag1); }
          if (j == 2) { rd128<48>(a[1][0], ag); rd128<48>(a[1][1], ag1); }
          lgkm_wait<2>();
        } else {
#ifndef NOA
          lgkm_wait<0>();
#endif
        }
        asm volatile("" : "+v"(a[j & 1][0]), "+v"(a[j & 1][1]));
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][tm][c], b[buf][tn][j][c], acc[tm][tn], 0, 0, 0);
      }
      asm volatile("" ::"v"(ag), "v"(ag1));
    };
#ifdef NOB
    load_b(1, 1);
#endif
#pragma unroll 1
    for (int g = 0; g < G; g += 2) {
#ifndef NOB
      load_b(1, g + 1);
#endif
      group(0, g);
#ifndef NOB
      load_b(0, g + 2 < G ? g + 2 : g);
#endif
      group(1, g + 1);
    }
    const unsigned long long s2 = __builtin_amdgcn_s_memtime();
    __syncthreads();   // every wave has read its last A fragment: the next block's rows may overwrite the image
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int col = n0 + 32 * tn + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float x = acc[tm][tn][r] + bv[tn];
          x = x > 0.f ? x : 0.01f * x;
          ((gf)C)[(size_t)row * 256 + col] = x;
        }
      }
#ifdef STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const unsigned long long s3 = __builtin_amdgcn_s_memtime();
    t_load += s1 - s0; t_k += s2 - s1; t_store += s3 - s2; ++nb;
  }
  if (dbg && tid == 0) {
    unsigned long long *d = (unsigned long long *)(dbg + 3 * 4096) + 6 * blockIdx.x;
    d[0] = t_load; d[1] = t_k; d[2] = t_store; d[3] = nb;
    d[4] = __builtin_amdgcn_s_memtime() - c_begin; d[5] = wall_clock64() - w_begin;
  }
}

extern "C" int proto_rowblock(const float *A, const float *W, const float *bias, float *C, int M, int Kk, void *stream) {
  if (M % BM || Kk != K) return -1;
  static unsigned *dbg = nullptr;
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return -2;
    ncu = p.multiProcessorCount;
    if (getenv("RB2_DEBUG")) (void)hipMalloc(&dbg, 3 * 4 * 4096 + 48 * 4096);
  }
  int grid = ncu * MINB;
  if (const char *e = getenv("RB2_GRID")) grid = atoi(e);
  if (grid > M / BM) grid = M / BM;
  hipLaunchKernelGGL(k_rowblock2, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, W, bias, C, M, dbg);
  if (dbg) {
    static int launches = 0;
    if (++launches == 40) {
      (void)hipDeviceSynchronize();
      unsigned *h = (unsigned *)malloc(3 * 4 * grid);
      (void)hipMemcpy(h, dbg, 3 * 4 * grid, hipMemcpyDeviceToHost);
      free(h);
      unsigned long long *t = (unsigned long long *)malloc(48 * grid);
      (void)hipMemcpy(t, (char *)dbg + 3 * 4 * 4096, 48 * grid, hipMemcpyDeviceToHost);
      double a[3] = {0, 0, 0}, n = 0, cyc = 0, wall = 0, cmax = 0;
      for (int i = 0; i < grid; ++i) {
        for (int k = 0; k < 3; ++k) a[k] += (double)t[6 * i + k];
        n += (double)t[6 * i + 3]; cyc += (double)t[6 * i + 4]; wall += (double)t[6 * i + 5];
        if ((double)t[6 * i + 4] > cmax) cmax = (double)t[6 * i + 4];
      }
      printf("stamps over %d workgroups, %.0f blocks: load %.0f  kloop %.0f  store %.0f ticks per block; workgroup life %.0f ticks mean, %.0f max; "
             "memtime ticks per 100 MHz tick %.3f\n", grid, n, a[0] / n, a[1] / n, a[2] / n, cyc / grid, cmax, cyc / wall);
      free(t);
    }
  }
  return (int)hipGetLastError();
}
