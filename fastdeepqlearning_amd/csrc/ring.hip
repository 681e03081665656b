// HBM-resident structure-of-arrays replay ring + windowed minibatch gather + write-time
// episode transforms (n-step return, hindsight relabel).  gfx950, wave64.
//
// Layout: one [maxlen, dim_k] f32 block per key (obs_1d, action, reward, ...), so a window
// of T consecutive slots of one key is one contiguous run of T*dim_k floats (modulo the
// wrap at `len`).  The gather writes the reference's [T, B, dim] batch layout
// (franQ/Replay/replay_memory.py:62-70).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"

namespace fdql {

constexpr int RING_MAX_KEYS = 16;
constexpr int GATHER_THREADS = 256;
constexpr int STAGE_WINDOWS = 8;         // windows per block in the LDS-staged path
constexpr int STAGE_LDS_FLOATS = 8192;   // 32 KiB of staging per block

struct GatherKey {
  const float *src;  // [maxlen, pitch] (+ column offset already applied)
  float *dst;        // [T, B, dim]
  int dim;           // floats gathered per row
  int pitch;         // floats between rows of the ring block
  int staged;        // 1: LDS-staged transposition (narrow rows), 0: direct row copy (wide rows)
  int u8;            // the ring block holds bytes (src is really `const uint8_t *`, pitch / offsets in elements)
  int tchunk;        // staged: time steps per block
  int block_start;
  int blocks_b, blocks_t;
};
struct GatherArgs {
  int nkeys, T, B;
  long long len;            // modulus of the window rows (the ring's len; maxlen for explicit index gathers)
  const long long *starts;  // [B] supplied by the caller, or null: drawn in the kernel (Philox, below)
  long long range;          // starts == null: start[b] uniform in [0, range)
  uint64_t seed, counter;
  long long *starts_out;    // optional [B]: the starts used (written by the blocks of key 0)
  GatherKey key[RING_MAX_KEYS];
};

// Philox-free tiny hash would do, but keep one generator family in the library.
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                        uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// start[b] uniform in [0, range)  (replay_memory.py:59 draws numpy randint(0, len - T, B))
__device__ __forceinline__ long long draw_start(int b, long long range, uint64_t seed, uint64_t counter) {
  uint32_t r[4];
  philox4((uint32_t)b, (uint32_t)counter, (uint32_t)(counter >> 32), 0x72696e67u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const uint64_t x = ((uint64_t)r[0] << 32) | r[1];
  return (long long)(((unsigned __int128)x * (unsigned __int128)(uint64_t)range) >> 64);
}
// Window start of batch element b: drawn here (no separate launch) or read from the caller's array, where any int64 is
// accepted and reduced into [0, len) the way numpy's `%` does (replay_memory.py:64)
__device__ __forceinline__ long long window_start(const GatherArgs &a, int b) {
  if (!a.starts) return draw_start(b, a.range, a.seed, a.counter);
  long long s = a.starts[b] % a.len;
  return s < 0 ? s + a.len : s;
}

// One launch gathers every key.  Narrow keys (dim*4 < 256 B): a block takes 16 windows x
// tchunk steps, reads each window's contiguous run into LDS with the window starts staged
// in LDS, and writes [t][16 windows][dim] runs; both sides are >= 16*dim*4-byte bursts.
// Wide keys: every (t, b) row is already a long contiguous run on both sides; plain
// coalesced row copy (float4 when dim % 4 == 0).
__global__ __launch_bounds__(GATHER_THREADS) void k_gather_windows(GatherArgs a) {
  __shared__ __attribute__((aligned(16))) float stage[STAGE_LDS_FLOATS];
  __shared__ long long sstart[STAGE_WINDOWS];
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int ki = 0;
  for (int i = 1; i < a.nkeys; ++i)
    if (bid >= a.key[i].block_start) ki = i;
  const GatherKey &K = a.key[ki];
  const int local = bid - K.block_start;
  const int dim = K.dim, T = a.T, B = a.B;
  const long long len = a.len;
  if (K.staged) {
    const int bb = local % K.blocks_b, tb = local / K.blocks_b;
    const int b0 = bb * STAGE_WINDOWS, t0 = tb * K.tchunk;
    const int nw = min(STAGE_WINDOWS, B - b0), nt = min(K.tchunk, T - t0);
    if (tid < nw) {   // index block staged in LDS
      const long long st = window_start(a, b0 + tid);
      sstart[tid] = (st + t0) % len;
      if (a.starts_out && ki == 0 && tb == 0) a.starts_out[b0 + tid] = st;
    }
    __syncthreads();
    const int run = nt * dim;  // floats per window in this chunk: ONE contiguous source run
    for (int w = wave; w < nw; w += GATHER_THREADS / 64) {
      const long long s0 = sstart[w];
      float *dstl = stage + w * run;
      if (s0 + nt <= len && K.pitch == dim) {
        const float *src = K.src + s0 * dim;
        for (int e = lane; e < run; e += 64) dstl[e] = src[e];
      } else if (s0 + nt <= len) {   // sub-row selection: runs of `dim` floats, `pitch` apart
        for (int e = lane; e < run; e += 64) {
          const int t = e / dim;
          dstl[e] = K.src[(s0 + t) * K.pitch + (e - t * dim)];
        }
      } else {  // the window crosses the wrap point (only with caller-supplied starts)
        for (int e = lane; e < run; e += 64) {
          const int t = e / dim;
          long long row = s0 + t;
          if (row >= len) row -= len;
          dstl[e] = K.src[row * K.pitch + (e - t * dim)];
        }
      }
    }
    __syncthreads();
    const int wrun = nw * dim;  // floats per time step written by this block (contiguous in the output)
    float *dst0 = K.dst + ((long long)t0 * B + b0) * dim;
    const long long tstride = (long long)B * dim;
    for (int r = tid; r < wrun; r += GATHER_THREADS) {
      const int w = r / dim, c = r - w * dim;
      const float *sp = stage + w * run + c;
      float *dp = dst0 + r;
      for (int t = 0; t < nt; ++t) dp[t * tstride] = sp[t * dim];
    }
  } else {
    // wide rows: one (t, b) row per wave iteration, lanes stride over the row.  (Round 5 tried blocks of (window, 32 time steps)
    // with four rows per wave in flight - contiguous 48 KB reads, rows scattered B x dim apart on the write side: 66 us against
    // 41 us for this form at config 4, B = 1024 (profiles/r05_sampler_pmc.txt); consecutive (t, b) rows on consecutive waves
    // keep the WRITE side contiguous, which is what matters.)
    const int rows = T * B;
    const int waves_total = K.blocks_b * (GATHER_THREADS / 64);
    for (int row = local * (GATHER_THREADS / 64) + wave; row < rows; row += waves_total) {
      const int t = row / B, b = row - t * B;
      const long long st = window_start(a, b);
      long long srow = st + t;
      if (srow >= len) srow %= len;
      if (a.starts_out && ki == 0 && t == 0 && lane == 0) a.starts_out[b] = st;
      float *dst = K.dst + (long long)row * dim;
      if (K.u8) {   // widen bytes to float32 (torch_dataloader.py:36), 4 elements per lane step when aligned
        const uint8_t *sb = reinterpret_cast<const uint8_t *>(K.src) + srow * K.pitch;
        if ((dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(sb) & 3) == 0)) {
          for (int e = lane; e < (dim >> 2); e += 64) {
            const uchar4 v = reinterpret_cast<const uchar4 *>(sb)[e];
            reinterpret_cast<float4 *>(dst)[e] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
          }
        } else {
          for (int e = lane; e < dim; e += 64) dst[e] = (float)sb[e];
        }
        continue;
      }
      const float *src = K.src + srow * K.pitch;
      if ((dim & 3) == 0 && (K.pitch & 3) == 0 && ((reinterpret_cast<uintptr_t>(K.src) & 15) == 0)) {
        for (int e = lane; e < (dim >> 2); e += 64)
          reinterpret_cast<float4 *>(dst)[e] = reinterpret_cast<const float4 *>(src)[e];
      } else {
        for (int e = lane; e < dim; e += 64) dst[e] = src[e];
      }
    }
  }
}

// packed AoS rows [n, rowfloats] -> SoA ring slots (top + i) % maxlen
struct ScatterArgs {
  int nkeys;
  long long n, top, maxlen;
  int rowfloats;
  const float *rows;
  float *dst[RING_MAX_KEYS];
  int dim[RING_MAX_KEYS];
  int off[RING_MAX_KEYS];
  int u8[RING_MAX_KEYS];   // the key's block holds bytes
};
__global__ void k_scatter_rows(ScatterArgs a) {
  const long long total = a.n * a.rowfloats;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / a.rowfloats;
    const int c = (int)(e - i * a.rowfloats);
    int k = 0;
    for (int j = 1; j < a.nkeys; ++j)
      if (c >= a.off[j]) k = j;
    const long long slot = (a.top + i) % a.maxlen;
    const long long at = slot * a.dim[k] + (c - a.off[k]);
    if (a.u8[k]) reinterpret_cast<uint8_t *>(a.dst[k])[at] = (uint8_t)fminf(fmaxf(a.rows[e], 0.f), 255.f);
    else a.dst[k][at] = a.rows[e];
  }
}

// inverse of k_scatter_rows for a run of slots [slot0, slot0 + n): SoA -> packed rows (ring checkpoint)
__global__ void k_pack_slots(ScatterArgs a) {
  const long long total = a.n * a.rowfloats;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / a.rowfloats;
    const int c = (int)(e - i * a.rowfloats);
    int k = 0;
    for (int j = 1; j < a.nkeys; ++j)
      if (c >= a.off[j]) k = j;
    const long long slot = a.top + i;
    const long long at = slot * a.dim[k] + (c - a.off[k]);
    const_cast<float *>(a.rows)[e] = a.u8[k] ? (float)reinterpret_cast<const uint8_t *>(a.dst[k])[at] : a.dst[k][at];
  }
}

// nstep_return.py:60-72: ret[i] = r[i] + gamma * ret[i+1] for OLDEST-FIRST arrays, float32
// multiply then add (no fma: the reference loop rounds the product before the add).
__global__ void k_mc_return(const float *__restrict__ r, float *__restrict__ ret, int n, float gamma, long long sr,
                            long long sret, float *__restrict__ first_only) {
#pragma clang fp contract(off)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float acc = 0.f;
  for (int i = n - 1; i >= 0; --i) {
    float prod = acc * gamma;      // rounded product, then rounded sum (no fma)
    acc = (i == n - 1) ? r[i * sr] : r[i * sr] + prod;
    if (!first_only) ret[i * sret] = acc;
  }
  if (first_only && n > 0) *first_only = acc;   // NStepReturn._pop: only record 0's return is used
}

__device__ __forceinline__ float reward_fn(const fdql_reward_fn_t &fn, const float *ag, const float *g, int gd) {
#pragma clang fp contract(off)
  float s = 0.f;
  for (int j = 0; j < gd; ++j) {
    const float d = ag[j] - g[j];
    const float sq = d * d;
    s = s + sq;
  }
  return sqrtf(s) > fn.threshold ? fn.miss_reward : 0.f;
}

// her.py:55-95.  Pass 1 (all threads): relabelled reward and done per step.  Pass 2: the
// synthetic sub-episode of step i starts after the last relabelled-done step BEFORE i, so
// episode_step'[i] = step[i] - step[p(i) + 1] with p = exclusive prefix-max of
// (done[j] ? j : -1): a wavefront scan carried across 64-step chunks.
struct HerStrides {
  long long reward, estep, ag, dg, r_out, d_out, s_out;  // floats between consecutive steps
  int dup_prev;  // also write step 0's outputs one stride earlier (the _pop record's slot)
};
__global__ __launch_bounds__(64) void k_her_relabel(const float *reward, const float *estep, const float *ag,
                                                    const float *dg, const float *goal, int n, int gd,
                                                    fdql_reward_fn_t fn, float *r_out, float *d_out, float *s_out,
                                                    HerStrides st) {
  const int lane = threadIdx.x;
  int carry = -1;  // last done index seen in earlier chunks
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    int flag = -1;
    if (i < n) {
      const float gr = reward_fn(fn, ag + i * st.ag, goal, gd);
      const float dr = reward_fn(fn, ag + i * st.ag, dg + i * st.dg, gd);
      const float rn = (reward[i * st.reward] - dr) + gr;
      const bool done = (gr == 0.f);
      r_out[i * st.r_out] = rn;
      d_out[i * st.d_out] = done ? 1.f : 0.f;
      if (st.dup_prev && i == 0) { r_out[-st.r_out] = rn; d_out[-st.d_out] = done ? 1.f : 0.f; }
      flag = done ? i : -1;
    }
    // inclusive prefix max over the wave
    int incl = flag;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl = max(incl, o);
    }
    int excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -1;
    excl = max(excl, carry);
    if (i < n) {
      const float sn = estep[i * st.estep] - estep[(excl + 1) * st.estep];
      s_out[i * st.s_out] = sn;
      if (st.dup_prev && i == 0) s_out[-st.s_out] = sn;
    }
    carry = max(carry, __shfl(incl, 63, 64));
  }
}

// Episode append: copy the n staged rows into `parts` blocks of (pop + n) rows (block 1 = the
// hindsight copy with desired_goal := the virtual goal); a block's first row, when pop, is the
// duplicate of record 0 that NStepReturn._pop emits (quirk q3).
struct ExpandArgs {
  const float *in;
  float *out;
  long long n;
  int F, pop, parts;
  int dg_off, ag_off, gd;
  long long goal_row;
};
__global__ void k_episode_expand(ExpandArgs a) {
  const long long per = a.n + a.pop;
  const long long total = per * a.parts * a.F;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long o = e / a.F;
    const int c = (int)(e - o * a.F);
    const int part = (int)(o / per);
    const long long j = o - part * per;
    const long long i = a.pop ? (j > 0 ? j - 1 : 0) : j;
    float v = a.in[i * a.F + c];
    if (part == 1 && c >= a.dg_off && c < a.dg_off + a.gd) v = a.in[a.goal_row * a.F + a.ag_off + (c - a.dg_off)];
    a.out[e] = v;
  }
}

// her_vmap.py:30-45: one thread per (step, virtual goal column); the last column is the real one
// (ld*: floats between consecutive records of each array: gd / K1 * gd / K1 for packed arrays, the ring's row width when the
// arrays are columns of packed rows - fdql_ring_append_episode_vmap relabels the staged rows in place)
__global__ void k_her_vmap(const float *reward, const float *task_done, const float *ag, const float *dg,
                           const int *goal_idx, int n, int gd, int K, fdql_reward_fn_t fn, float *vgoals,
                           float *vrew, float *vdone, long long ld_s, long long ld_g, long long ld_vg, long long ld_v) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int K1 = K + 1;
  if (e >= n * K1) return;
  const int i = e / K1, k = e - i * K1;
  const float *ag_i = ag + (long long)i * ld_g;
  const float *goal = k < K ? ag + (long long)goal_idx[k] * ld_g : dg + (long long)i * ld_g;
  float *vg = vgoals + (long long)i * ld_vg + (long long)k * gd;
  for (int j = 0; j < gd; ++j) vg[j] = goal[j];
  const long long o = (long long)i * ld_v + k;
  const float rew = reward[(long long)i * ld_s], td = task_done[(long long)i * ld_s];
  if (k == K) {
    vrew[o] = rew;
    vdone[o] = td != 0.f ? 1.f : 0.f;
    return;
  }
  const float dr = reward_fn(fn, ag_i, dg + (long long)i * ld_g, gd);
  const float vr = reward_fn(fn, ag_i, goal, gd);
  vrew[o] = (rew - dr) + vr;
  const bool agnostic_done = (td != 0.f) && !(dr == 0.f);
  vdone[o] = (agnostic_done || vr == 0.f) ? 1.f : 0.f;
}
__global__ void k_mc_return_vmap(const float *__restrict__ r, const float *__restrict__ d, float *__restrict__ ret,
                                 int n, int cols, float gamma, long long ld, float *__restrict__ first_out) {
#pragma clang fp contract(off)
  // ld: floats between consecutive records (cols for packed arrays); ret == null: only the scan's value at record 0 is kept,
  // in first_out[c] (the one-shot _pop record of nstep_return_vmap.py:33-34, 50-57)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (int i = n - 1; i >= 0; --i) {
    const float prod = (acc * gamma) * (d[(long long)i * ld + c] != 0.f ? 1.f : 0.f);
    acc = (i == n - 1) ? r[(long long)i * ld + c] : r[(long long)i * ld + c] + prod;
    if (ret) ret[(long long)i * ld + c] = acc;
  }
  if (first_out) first_out[c] = acc;
}

}  // namespace fdql

using namespace fdql;

// slot of row (t, b) of a windowed sample: (start[b] + t) % len (replay_memory.py:63-65), as int32 [T, B]
__global__ void k_window_slots(const long long *__restrict__ starts, int T, int B, long long len, int *__restrict__ slots) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)T * B) return;
  const long long t = i / B, b = i - t * B;
  long long st = starts[b] % len;
  if (st < 0) st += len;
  slots[i] = (int)((st + t) % len);
}

struct fdql_ring {
  // Every entry point that takes a ring locks `mu`: the Runner's pattern is one writer thread per shard calling add()
  // while the trainer thread samples the same shard (franQ/Replay/async_replay_memory.py:55-70, runner.py:177-191);
  // ctypes drops the GIL around each call.  The mutex covers host bookkeeping and launch order only - nothing waits
  // for the GPU under it except where an entry point's comment says it synchronises.
  std::mutex mu;
  int64_t maxlen = 0;
  int nkeys = 0;
  int dims[RING_MAX_KEYS] = {};
  int offs[RING_MAX_KEYS] = {};
  int rowfloats = 0;
  float *data[RING_MAX_KEYS] = {};   // uint8 keys: really `uint8_t *` (element offsets, not float offsets)
  int u8[RING_MAX_KEYS] = {};
  // bookkeeping (replay_memory.py:45-46)
  int64_t top = 0, len = 0;
  int64_t sample_len = 0;   // the length the last window gather reduced its starts by (fdql_ring_window_slots uses the same one)
  // staging of add(): two pinned / device buffer pairs, so that filling one never waits for the H2D copy of the other
  float *pinned[2] = {nullptr, nullptr};
  float *dev_stage[2] = {nullptr, nullptr};
  hipEvent_t stage_done[2] = {nullptr, nullptr};
  bool stage_inflight[2] = {false, false};
  int cur = 0;
  int64_t stage_cap = 0, staged = 0, stage_top = 0;
  // Device-side order between writers and readers on DIFFERENT streams (same stream: the stream is the order).
  // Used only once a second stream has been seen on this handle.
  hipStream_t first_stream = nullptr;
  bool have_stream = false, multi_stream = false;
  // One event for the last writer; one per distinct reader stream (RING_RD_SLOTS of them; when they run out a new
  // reader first waits for the slot it takes over, so its own event stands for both).  While the handle has seen one
  // stream the events are not recorded; the moment a second stream shows up every pending one is recorded on its own
  // stream (conservative: after everything issued there so far) before anybody waits on it.
  static constexpr int RING_RD_SLOTS = 4;
  hipEvent_t wr_done = nullptr;
  hipStream_t wr_stream = nullptr;
  bool has_wr = false;
  hipEvent_t rd_done[RING_RD_SLOTS] = {};
  hipStream_t rd_stream[RING_RD_SLOTS] = {};
  bool has_rd[RING_RD_SLOTS] = {};
  int rd_next = 0;
  // episode append staging (fdql_ring_append_episode)
  float *ep_pinned = nullptr, *ep_in = nullptr, *ep_out = nullptr;
  int64_t ep_in_cap = 0, ep_out_cap = 0;   // rows
  hipEvent_t ep_done = nullptr;
  bool ep_inflight = false;
};

namespace {

void advance(fdql_ring *r, int64_t n) {
  // n sequential adds of replay_memory.py:45-46
  for (int64_t done = 0; done < n;) {
    const int64_t step = std::min<int64_t>(n - done, r->maxlen - r->top);
    const int64_t newtop = (r->top + step) % r->maxlen;
    // len = max(top, len) after each add: the running max of top over this contiguous run
    const int64_t peak = (newtop == 0) ? (step > 1 ? r->maxlen - 1 : 0) : newtop;
    r->len = std::max(r->len, peak);
    r->top = newtop;
    done += step;
  }
}

int see_stream(fdql_ring *r, hipStream_t s) {
  if (!r->have_stream) { r->first_stream = s; r->have_stream = true; return 0; }
  if (s == r->first_stream || r->multi_stream) return 0;
  // second stream: whatever was issued while single-stream has no recorded event yet - record it now, on its stream
  r->multi_stream = true;
  if (r->has_wr) FDQL_HIP(hipEventRecord(r->wr_done, r->wr_stream));
  for (int i = 0; i < fdql_ring::RING_RD_SLOTS; ++i)
    if (r->has_rd[i]) FDQL_HIP(hipEventRecord(r->rd_done[i], r->rd_stream[i]));
  return 0;
}
// a launch on `s` that WRITES ring slots: after every read issued on another stream, and after writes elsewhere
int begin_write(fdql_ring *r, hipStream_t s) {
  int rc = see_stream(r, s);
  if (rc || !r->multi_stream) return rc;
  for (int i = 0; i < fdql_ring::RING_RD_SLOTS; ++i)
    if (r->has_rd[i] && r->rd_stream[i] != s) FDQL_HIP(hipStreamWaitEvent(s, r->rd_done[i], 0));
  if (r->has_wr && r->wr_stream != s) FDQL_HIP(hipStreamWaitEvent(s, r->wr_done, 0));
  return 0;
}
int end_write(fdql_ring *r, hipStream_t s) {
  r->wr_stream = s; r->has_wr = true;   // remembered even while single-stream: see_stream() records it later
  if (!r->multi_stream) return 0;
  FDQL_HIP(hipEventRecord(r->wr_done, s));
  return 0;
}
int begin_read(fdql_ring *r, hipStream_t s) {
  int rc = see_stream(r, s);
  if (rc || !r->multi_stream) return rc;
  if (r->has_wr && r->wr_stream != s) FDQL_HIP(hipStreamWaitEvent(s, r->wr_done, 0));
  return 0;
}
int end_read(fdql_ring *r, hipStream_t s) {
  int slot = -1;
  for (int i = 0; i < fdql_ring::RING_RD_SLOTS; ++i)
    if (r->has_rd[i] && r->rd_stream[i] == s) { slot = i; break; }
  if (slot < 0)
    for (int i = 0; i < fdql_ring::RING_RD_SLOTS; ++i)
      if (!r->has_rd[i]) { slot = i; break; }
  if (slot < 0) {
    // every slot tracks another stream: take one over; this stream waits for that reader first, so the event recorded
    // below also covers it (multi_stream is necessarily true here, so its event is recorded)
    slot = r->rd_next; r->rd_next = (r->rd_next + 1) % fdql_ring::RING_RD_SLOTS;
    FDQL_HIP(hipStreamWaitEvent(s, r->rd_done[slot], 0));
  }
  r->rd_stream[slot] = s; r->has_rd[slot] = true;
  if (!r->multi_stream) return 0;
  FDQL_HIP(hipEventRecord(r->rd_done[slot], s));
  return 0;
}

int scatter(fdql_ring *r, const float *dev_rows, int64_t n, int64_t top, hipStream_t s) {
  ScatterArgs a;
  memset(&a, 0, sizeof(a));
  a.nkeys = r->nkeys; a.n = n; a.top = top; a.maxlen = r->maxlen; a.rowfloats = r->rowfloats; a.rows = dev_rows;
  for (int k = 0; k < r->nkeys; ++k) { a.dst[k] = r->data[k]; a.dim[k] = r->dims[k]; a.off[k] = r->offs[k]; a.u8[k] = r->u8[k]; }
  const long long total = n * r->rowfloats;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  int rc = begin_write(r, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_scatter_rows, dim3(blocks), dim3(256), 0, s, a);
  FDQL_HIP(hipGetLastError());
  return end_write(r, s);
}

int wait_stage(fdql_ring *r, int i) {
  if (r->stage_inflight[i]) { FDQL_HIP(hipEventSynchronize(r->stage_done[i])); r->stage_inflight[i] = false; }
  return 0;
}

// staged rows -> HBM: one H2D copy + one scatter launch on `s`; the other buffer pair takes the next adds
int flush(fdql_ring *r, hipStream_t s) {
  if (r->staged == 0) return 0;
  const int i = r->cur;
  FDQL_HIP(hipMemcpyAsync(r->dev_stage[i], r->pinned[i], r->staged * r->rowfloats * sizeof(float), hipMemcpyHostToDevice, s));
  int rc = scatter(r, r->dev_stage[i], r->staged, r->stage_top, s);
  if (rc) return rc;
  FDQL_HIP(hipEventRecord(r->stage_done[i], s));
  r->stage_inflight[i] = true;
  r->staged = 0;
  r->cur = i ^ 1;
  return 0;
}

int gather(fdql_ring *r, int T, int B, long long modulus, const long long *starts, long long range, uint64_t seed,
           uint64_t counter, long long *starts_out, float *const *out, const int32_t *sel_off, const int32_t *sel_dim,
           hipStream_t s) {
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.B = B; a.len = modulus; a.starts = starts; a.range = range; a.seed = seed; a.counter = counter;
  a.starts_out = starts_out;
  int total = 0;
  for (int k = 0; k < r->nkeys; ++k) {
    if (!out[k]) {
      FDQL_REQUIRE(sel_dim != nullptr, "output pointer for key %d is null", k);   // skipping keys needs the _sel entry point
      continue;
    }
    GatherKey &g = a.key[a.nkeys++];
    const int off = (sel_off && sel_dim && sel_dim[k] > 0) ? sel_off[k] : 0;
    const int dim = (sel_dim && sel_dim[k] > 0) ? sel_dim[k] : r->dims[k];
    FDQL_REQUIRE(off >= 0 && off + dim <= r->dims[k], "selection [%d, %d) outside key %d of width %d", off, off + dim, k, r->dims[k]);
    g.u8 = r->u8[k];
    g.src = g.u8 ? reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(r->data[k]) + off) : r->data[k] + off;
    g.dst = out[k]; g.dim = dim; g.pitch = r->dims[k];
    g.block_start = total;
    if (g.dim * 4 < 256 && T > 1 && !g.u8) {
      g.staged = 1;
      int tc = STAGE_LDS_FLOATS / (STAGE_WINDOWS * g.dim);
      g.tchunk = std::max(1, std::min(T, tc));
      g.blocks_b = (B + STAGE_WINDOWS - 1) / STAGE_WINDOWS;
      g.blocks_t = (T + g.tchunk - 1) / g.tchunk;
      total += g.blocks_b * g.blocks_t;
    } else {
      g.staged = 0;
      const long long rows = (long long)T * B;   // 4 rows per block pass; cap the grid and stride the rest
      g.blocks_b = (int)std::max<long long>(1, std::min<long long>((rows + 3) / 4, 4096));
      g.blocks_t = 1;
      total += g.blocks_b;
    }
  }
  if (total == 0) return 0;
  int rc = begin_read(r, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k_gather_windows, dim3(total), dim3(GATHER_THREADS), 0, s, a);
  FDQL_HIP(hipGetLastError());
  return end_read(r, s);
}

typedef std::lock_guard<std::mutex> Lock;

}  // namespace

extern "C" {

int fdql_ring_create(fdql_ring_t **out, int64_t maxlen, int32_t n_keys, const int32_t *dims) {
  return fdql_ring_create_typed(out, maxlen, n_keys, dims, nullptr);
}

int fdql_ring_create_typed(fdql_ring_t **out, int64_t maxlen, int32_t n_keys, const int32_t *dims, const int32_t *dtypes) {
  FDQL_REQUIRE(out && dims && maxlen > 0 && n_keys > 0 && n_keys <= RING_MAX_KEYS, "bad ring arguments");
  for (int k = 0; dtypes && k < n_keys; ++k)
    FDQL_REQUIRE(dtypes[k] == FDQL_F32 || dtypes[k] == FDQL_U8, "key %d: unknown storage type %d", k, dtypes[k]);
  fdql_ring *r = new fdql_ring();
  for (int k = 0; dtypes && k < n_keys; ++k) r->u8[k] = dtypes[k] == FDQL_U8;
  r->maxlen = maxlen;
  r->nkeys = n_keys;
  for (int k = 0; k < n_keys; ++k) {
    if (dims[k] <= 0) { delete r; set_error("key %d has dim %d", k, dims[k]); return FDQL_EINVAL; }
    r->dims[k] = dims[k];
    r->offs[k] = r->rowfloats;
    r->rowfloats += dims[k];
  }
  for (int k = 0; k < n_keys; ++k) {
    const size_t bytes = (size_t)maxlen * dims[k] * (r->u8[k] ? 1 : sizeof(float));
    hipError_t e = hipMalloc(&r->data[k], bytes);
    if (e == hipSuccess) e = hipMemset(r->data[k], 0, bytes);  // replay_memory.py:35 np.zeros
    if (e != hipSuccess) { set_error("ring alloc of key %d (%zu bytes): %s", k, bytes, hipGetErrorString(e)); fdql_ring_destroy(r); return FDQL_ENOMEM; }
  }
  r->stage_cap = std::max<int64_t>(1, std::min<int64_t>(maxlen, (int64_t)(8 << 20) / (r->rowfloats * 4)));
  for (int i = 0; i < 2; ++i) {
    hipError_t e = hipHostMalloc(&r->pinned[i], r->stage_cap * r->rowfloats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&r->dev_stage[i], r->stage_cap * r->rowfloats * sizeof(float));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->stage_done[i], hipEventDisableTiming);
    if (e != hipSuccess) { set_error("ring staging buffers: %s", hipGetErrorString(e)); fdql_ring_destroy(r); return FDQL_ENOMEM; }
  }
  hipError_t e = hipEventCreateWithFlags(&r->wr_done, hipEventDisableTiming);
  for (int i = 0; i < fdql_ring::RING_RD_SLOTS && e == hipSuccess; ++i)
    e = hipEventCreateWithFlags(&r->rd_done[i], hipEventDisableTiming);
  if (e != hipSuccess) { set_error("ring events: %s", hipGetErrorString(e)); fdql_ring_destroy(r); return FDQL_EHIP; }
  *out = r;
  return 0;
}

int fdql_ring_destroy(fdql_ring_t *r) {
  if (!r) return 0;
  for (int k = 0; k < r->nkeys; ++k)
    if (r->data[k]) (void)hipFree(r->data[k]);
  for (int i = 0; i < 2; ++i) {
    if (r->stage_inflight[i]) (void)hipEventSynchronize(r->stage_done[i]);
    if (r->pinned[i]) (void)hipHostFree(r->pinned[i]);
    if (r->dev_stage[i]) (void)hipFree(r->dev_stage[i]);
    if (r->stage_done[i]) (void)hipEventDestroy(r->stage_done[i]);
  }
  if (r->wr_done) (void)hipEventDestroy(r->wr_done);
  for (int i = 0; i < fdql_ring::RING_RD_SLOTS; ++i)
    if (r->rd_done[i]) (void)hipEventDestroy(r->rd_done[i]);
  if (r->ep_pinned) (void)hipHostFree(r->ep_pinned);
  if (r->ep_in) (void)hipFree(r->ep_in);
  if (r->ep_out) (void)hipFree(r->ep_out);
  if (r->ep_done) (void)hipEventDestroy(r->ep_done);
  delete r;
  return 0;
}

int fdql_ring_add(fdql_ring_t *r, const float *host_rows, int64_t n, void *stream) {
  FDQL_REQUIRE(r && host_rows && n >= 0, "bad arguments");
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  while (n > 0) {
    if (r->staged == 0) {
      // this buffer pair was flushed two batches ago: its copy has normally long retired
      int rc = wait_stage(r, r->cur);
      if (rc) return rc;
      r->stage_top = r->top;
    }
    const int64_t take = std::min(n, r->stage_cap - r->staged);
    memcpy(r->pinned[r->cur] + r->staged * r->rowfloats, host_rows, take * r->rowfloats * sizeof(float));
    r->staged += take;
    advance(r, take);
    host_rows += take * r->rowfloats;
    n -= take;
    if (r->staged == r->stage_cap) { int rc = flush(r, s); if (rc) return rc; }
  }
  return 0;
}

int fdql_ring_add_device(fdql_ring_t *r, const float *dev_rows, int64_t n, void *stream) {
  FDQL_REQUIRE(r && dev_rows && n >= 0, "bad arguments");
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  if (n == 0) return 0;
  rc = scatter(r, dev_rows, n, r->top, s);
  if (rc) return rc;
  advance(r, n);
  return 0;
}

int fdql_ring_flush(fdql_ring_t *r, void *stream) {
  FDQL_REQUIRE(r, "null ring");
  Lock lk(r->mu);
  return flush(r, (hipStream_t)stream);
}

int fdql_ring_append_episode(fdql_ring_t *r, const float *host_rows, int64_t n, const fdql_episode_spec_t *sp,
                             int64_t *appended, void *stream) {
  FDQL_REQUIRE(r && host_rows && sp && n >= 0, "bad arguments");
  if (appended) *appended = 0;
  if (n == 0) return 0;
  Lock lk(r->mu);
  auto scalar_key = [&](int k) { return k >= 0 && k < r->nkeys && r->dims[k] == 1; };
  FDQL_REQUIRE(sp->return_key < 0 || (scalar_key(sp->return_key) && scalar_key(sp->reward_key)),
               "append_episode: reward / mc_return must be keys of width 1");
  FDQL_REQUIRE(sp->return_key < 0 || !sp->emit_pop || sp->n_step >= 1, "append_episode: n_step must be >= 1");
  int gd = 0;
  if (sp->her) {
    FDQL_REQUIRE(sp->reward_fn.kind == 0, "unknown reward function kind %d", sp->reward_fn.kind);
    FDQL_REQUIRE(scalar_key(sp->reward_key) && scalar_key(sp->task_done_key) && scalar_key(sp->step_key),
                 "append_episode: reward / task_done / episode_step must be keys of width 1");
    FDQL_REQUIRE(sp->achieved_key >= 0 && sp->achieved_key < r->nkeys && sp->desired_key >= 0 && sp->desired_key < r->nkeys &&
                     r->dims[sp->achieved_key] == r->dims[sp->desired_key],
                 "append_episode: achieved_goal / desired_goal keys must exist with equal width");
    FDQL_REQUIRE(sp->goal_row >= 0 && sp->goal_row < n, "append_episode: goal_row %d outside the episode", sp->goal_row);
    gd = r->dims[sp->achieved_key];
  }
  const int pop = (sp->return_key >= 0 && sp->emit_pop && n > sp->n_step) ? 1 : 0;
  const int parts = sp->her ? 2 : 1;
  const int64_t per = n + pop, n_out = per * parts;
  FDQL_REQUIRE(n_out <= r->maxlen, "append_episode: %lld rows do not fit a ring of %lld slots", (long long)n_out, (long long)r->maxlen);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  const int F = r->rowfloats;
  if (r->ep_inflight) { FDQL_HIP(hipEventSynchronize(r->ep_done)); r->ep_inflight = false; }
  if (!r->ep_done) FDQL_HIP(hipEventCreateWithFlags(&r->ep_done, hipEventDisableTiming));
  if (n > r->ep_in_cap) {
    const int64_t cap = std::max<int64_t>(n, 2 * r->ep_in_cap);
    if (r->ep_pinned) FDQL_HIP(hipHostFree(r->ep_pinned));
    if (r->ep_in) FDQL_HIP(hipFree(r->ep_in));
    r->ep_pinned = nullptr; r->ep_in = nullptr; r->ep_in_cap = 0;
    FDQL_HIP(hipHostMalloc(&r->ep_pinned, cap * F * sizeof(float)));
    FDQL_HIP(hipMalloc(&r->ep_in, cap * F * sizeof(float)));
    r->ep_in_cap = cap;
  }
  if (n_out > r->ep_out_cap) {
    const int64_t cap = std::max<int64_t>(n_out, 2 * r->ep_out_cap);
    if (r->ep_out) FDQL_HIP(hipFree(r->ep_out));
    r->ep_out = nullptr; r->ep_out_cap = 0;
    FDQL_HIP(hipMalloc(&r->ep_out, cap * F * sizeof(float)));
    r->ep_out_cap = cap;
  }
  memcpy(r->ep_pinned, host_rows, n * F * sizeof(float));
  FDQL_HIP(hipMemcpyAsync(r->ep_in, r->ep_pinned, n * F * sizeof(float), hipMemcpyHostToDevice, s));
  ExpandArgs ea;
  memset(&ea, 0, sizeof(ea));
  ea.in = r->ep_in; ea.out = r->ep_out; ea.n = n; ea.F = F; ea.pop = pop; ea.parts = parts;
  if (sp->her) { ea.dg_off = r->offs[sp->desired_key]; ea.ag_off = r->offs[sp->achieved_key]; ea.gd = gd; ea.goal_row = sp->goal_row; }
  {
    const long long total = n_out * F;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_episode_expand, dim3(blocks), dim3(256), 0, s, ea);
    FDQL_HIP(hipGetLastError());
  }
  if (sp->her) {   // her.py:55-95 into block 1 (and its _pop duplicate)
    float *hb = r->ep_out + (per + pop) * F;   // hindsight record 0
    const HerStrides st = {F, F, F, F, F, F, F, pop};
    hipLaunchKernelGGL(k_her_relabel, dim3(1), dim3(64), 0, s, r->ep_in + r->offs[sp->reward_key],
                       r->ep_in + r->offs[sp->step_key], r->ep_in + r->offs[sp->achieved_key],
                       r->ep_in + r->offs[sp->desired_key], r->ep_in + sp->goal_row * (int64_t)F + r->offs[sp->achieved_key],
                       (int)n, gd, sp->reward_fn, hb + r->offs[sp->reward_key], hb + r->offs[sp->task_done_key],
                       hb + r->offs[sp->step_key], st);
    FDQL_HIP(hipGetLastError());
  }
  if (sp->return_key >= 0) {   // nstep_return.py:36-57 per block: the flush scan, and _pop's scan of the first n_step rewards
    for (int p = 0; p < parts; ++p) {
      float *blk = r->ep_out + p * per * F;
      float *rec0 = blk + pop * F;
      hipLaunchKernelGGL(k_mc_return, dim3(1), dim3(64), 0, s, rec0 + r->offs[sp->reward_key], rec0 + r->offs[sp->return_key],
                         (int)n, sp->gamma, (long long)F, (long long)F, (float *)nullptr);
      if (pop)
        hipLaunchKernelGGL(k_mc_return, dim3(1), dim3(64), 0, s, rec0 + r->offs[sp->reward_key], (float *)nullptr, sp->n_step,
                           sp->gamma, (long long)F, (long long)F, blk + r->offs[sp->return_key]);
      FDQL_HIP(hipGetLastError());
    }
  }
  rc = scatter(r, r->ep_out, n_out, r->top, s);
  if (rc) return rc;
  advance(r, n_out);
  FDQL_HIP(hipEventRecord(r->ep_done, s));
  r->ep_inflight = true;
  if (appended) *appended = n_out;
  return 0;
}

int fdql_ring_snapshot(fdql_ring_t *r, float *host_rows_out, int64_t n_slots, void *stream) {
  FDQL_REQUIRE(r && host_rows_out && n_slots >= 0 && n_slots <= r->maxlen, "bad arguments");
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  for (int i = 0; i < 2; ++i) { rc = wait_stage(r, i); if (rc) return rc; }
  rc = begin_read(r, s);
  if (rc) return rc;
  const int F = r->rowfloats;
  for (int64_t done = 0; done < n_slots;) {   // through staging pair 0, stage_cap slots at a time
    const int64_t n = std::min<int64_t>(r->stage_cap, n_slots - done);
    ScatterArgs a;
    memset(&a, 0, sizeof(a));
    a.nkeys = r->nkeys; a.n = n; a.top = done; a.maxlen = r->maxlen; a.rowfloats = F; a.rows = r->dev_stage[0];
    for (int k = 0; k < r->nkeys; ++k) { a.dst[k] = r->data[k]; a.dim[k] = r->dims[k]; a.off[k] = r->offs[k]; a.u8[k] = r->u8[k]; }
    const long long total = n * F;
    hipLaunchKernelGGL(k_pack_slots, dim3((int)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, s, a);
    FDQL_HIP(hipGetLastError());
    FDQL_HIP(hipMemcpyAsync(r->pinned[0], r->dev_stage[0], total * sizeof(float), hipMemcpyDeviceToHost, s));
    FDQL_HIP(hipStreamSynchronize(s));
    memcpy(host_rows_out + done * F, r->pinned[0], total * sizeof(float));
    done += n;
  }
  return 0;
}

int fdql_ring_restore(fdql_ring_t *r, const float *host_rows, int64_t n_slots, int64_t top, int64_t len, void *stream) {
  FDQL_REQUIRE(r && (host_rows || n_slots == 0) && n_slots >= 0 && n_slots <= r->maxlen, "bad arguments");
  FDQL_REQUIRE(top >= 0 && top < r->maxlen && len >= 0 && len < r->maxlen && len <= n_slots,
               "restore: (top=%lld, len=%lld) inconsistent with %lld slots of a ring of %lld", (long long)top,
               (long long)len, (long long)n_slots, (long long)r->maxlen);
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  const int F = r->rowfloats;
  int i = r->cur;
  for (int64_t done = 0; done < n_slots; i ^= 1) {
    rc = wait_stage(r, i);
    if (rc) return rc;
    const int64_t n = std::min<int64_t>(r->stage_cap, n_slots - done);
    memcpy(r->pinned[i], host_rows + done * F, n * F * sizeof(float));
    FDQL_HIP(hipMemcpyAsync(r->dev_stage[i], r->pinned[i], n * F * sizeof(float), hipMemcpyHostToDevice, s));
    rc = scatter(r, r->dev_stage[i], n, done, s);
    if (rc) return rc;
    FDQL_HIP(hipEventRecord(r->stage_done[i], s));
    r->stage_inflight[i] = true;
    done += n;
  }
  r->top = top;
  r->len = len;
  r->sample_len = 0;
  return 0;
}

int64_t fdql_ring_len(const fdql_ring_t *r) {
  if (!r) return -1;
  Lock lk(const_cast<fdql_ring_t *>(r)->mu);
  return r->len;
}
int64_t fdql_ring_top(const fdql_ring_t *r) {
  if (!r) return -1;
  Lock lk(const_cast<fdql_ring_t *>(r)->mu);
  return r->top;
}
int64_t fdql_ring_row_floats(const fdql_ring_t *r) { return r ? r->rowfloats : -1; }

int fdql_ring_key_ptr(fdql_ring_t *r, int32_t key, float **dev_ptr) {
  FDQL_REQUIRE(r && dev_ptr && key >= 0 && key < r->nkeys, "bad key");
  FDQL_REQUIRE(!r->u8[key], "key %d is stored as uint8", key);
  *dev_ptr = r->data[key];
  return 0;
}

int fdql_ring_key_ptr_u8(fdql_ring_t *r, int32_t key, uint8_t **dev_ptr) {
  FDQL_REQUIRE(r && dev_ptr && key >= 0 && key < r->nkeys, "bad key");
  FDQL_REQUIRE(r->u8[key], "key %d is stored as float32", key);
  *dev_ptr = reinterpret_cast<uint8_t *>(r->data[key]);
  return 0;
}

int fdql_ring_window_slots(fdql_ring_t *r, int32_t T, int32_t B, const int64_t *starts_dev, int32_t *slots_out_dev, void *stream) {
  FDQL_REQUIRE(r && starts_dev && slots_out_dev && T >= 1 && B >= 1, "bad arguments");
  Lock lk(r->mu);
  FDQL_REQUIRE(r->len >= 1 && r->maxlen < (1LL << 31), "empty ring, or more slots than an int32 index holds");
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)T * B;
  // the window gather that drew `starts` wrapped its rows at the length it saw (replay_memory.py:63-65): a writer that
  // flushed between the two calls must not make the slots wrap somewhere else
  const long long len = r->sample_len > 0 ? r->sample_len : r->len;
  hipLaunchKernelGGL(k_window_slots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const long long *>(starts_dev), T, B,
                     len, slots_out_dev);
  FDQL_HIP(hipGetLastError());
  return 0;
}

int fdql_ring_external_read(fdql_ring_t *r, int32_t begin, void *stream) {
  FDQL_REQUIRE(r, "null ring");
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  if (begin) {
    int rc = flush(r, s);   // staged add()s become visible to the reader, like a sample would make them
    if (rc) return rc;
    return begin_read(r, s);
  }
  return end_read(r, s);
}

int fdql_ring_sample_windows(fdql_ring_t *r, int32_t T, int32_t B, const int64_t *starts_dev, uint64_t seed,
                             uint64_t counter, float *const *out, int64_t *starts_out_dev, void *stream) {
  return fdql_ring_sample_windows_sel(r, T, B, starts_dev, seed, counter, out, nullptr, nullptr, starts_out_dev, stream);
}

int fdql_ring_sample_windows_sel(fdql_ring_t *r, int32_t T, int32_t B, const int64_t *starts_dev, uint64_t seed,
                                 uint64_t counter, float *const *out, const int32_t *sel_off, const int32_t *sel_dim,
                                 int64_t *starts_out_dev, void *stream) {
  FDQL_REQUIRE(r && out && T >= 1 && B >= 1, "bad arguments");
  Lock lk(r->mu);
  if (r->len < 2 * (int64_t)T || r->len < B) {  // replay_memory.py:57-58
    set_error("Trying to sample more memories than available! (len=%lld, T=%d, B=%d)", (long long)r->len, T, B);
    return FDQL_EOVERSAMPLE;
  }
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  // starts drawn inside the gather (one launch); caller-supplied starts are reduced mod len there
  r->sample_len = r->len;
  return gather(r, T, B, r->len, reinterpret_cast<const long long *>(starts_dev), (long long)(r->len - T), seed, counter,
                reinterpret_cast<long long *>(starts_out_dev), out, sel_off, sel_dim, s);
}

int fdql_ring_sample_rows(fdql_ring_t *r, int32_t B, const int64_t *idx_dev, uint64_t seed, uint64_t counter,
                          float *const *out, int64_t *idx_out_dev, void *stream) {
  FDQL_REQUIRE(r && out && B >= 1, "bad arguments");
  Lock lk(r->mu);
  if (r->len < B && !idx_dev) {  // replay_memory.py:50 guards the random draw only; explicit indices are the caller's
    set_error("Trying to sample more memories than available! (len=%lld, B=%d)", (long long)r->len, B);
    return FDQL_EOVERSAMPLE;
  }
  FDQL_REQUIRE(r->len > 0, "the ring is empty");
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  return gather(r, 1, B, r->len, reinterpret_cast<const long long *>(idx_dev), (long long)r->len, seed, counter,
                reinterpret_cast<long long *>(idx_out_dev), out, nullptr, nullptr, s);
}

int fdql_ring_gather_rows(fdql_ring_t *r, int64_t n, const int64_t *idx_dev, float *const *out, void *stream) {
  FDQL_REQUIRE(r && out && idx_dev && n >= 0 && n < (1LL << 31), "bad arguments");
  if (n == 0) return 0;
  Lock lk(r->mu);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  // slots are addressed up to maxlen whatever `len` is, like numpy indexing of the [maxlen, ...] arrays
  return gather(r, 1, (int)n, r->maxlen, reinterpret_cast<const long long *>(idx_dev), 0, 0, 0, nullptr, out, nullptr, nullptr, s);
}

int fdql_episode_mc_return(const float *reward_dev, float *ret_dev, int32_t n, float gamma, void *stream) {
  FDQL_REQUIRE(reward_dev && ret_dev && n >= 0, "bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_mc_return, dim3(1), dim3(64), 0, (hipStream_t)stream, reward_dev, ret_dev, n, gamma, 1LL, 1LL,
                     (float *)nullptr);
  FDQL_HIP(hipGetLastError());
  return 0;
}

int fdql_episode_her_vmap(const float *reward, const float *task_done, const float *achieved_goal,
                          const float *desired_goal, const int32_t *goal_idx, int32_t n, int32_t goal_dim, int32_t K,
                          const fdql_reward_fn_t *fn, float *virtual_goals, float *virtual_rewards,
                          float *virtual_dones, void *stream) {
  FDQL_REQUIRE(reward && task_done && achieved_goal && desired_goal && goal_idx && fn && virtual_goals &&
                   virtual_rewards && virtual_dones && n >= 0 && goal_dim > 0 && K >= 0, "bad arguments");
  FDQL_REQUIRE(fn->kind == 0, "unknown reward function kind %d", fn->kind);
  if (n == 0) return 0;
  const int total = n * (K + 1);
  hipLaunchKernelGGL(k_her_vmap, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, reward, task_done,
                     achieved_goal, desired_goal, goal_idx, n, goal_dim, K, *fn, virtual_goals, virtual_rewards,
                     virtual_dones, 1LL, (long long)goal_dim, (long long)(K + 1) * goal_dim, (long long)(K + 1));
  FDQL_HIP(hipGetLastError());
  return 0;
}

int fdql_episode_mc_return_vmap(const float *rewards, const float *dones, float *ret, int32_t n, int32_t cols,
                                float gamma, void *stream) {
  FDQL_REQUIRE(rewards && dones && ret && n >= 0 && cols > 0, "bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_mc_return_vmap, dim3((cols + 63) / 64), dim3(64), 0, (hipStream_t)stream, rewards, dones, ret, n,
                     cols, gamma, (long long)cols, (float *)nullptr);
  FDQL_HIP(hipGetLastError());
  return 0;
}

int fdql_ring_append_episode_vmap(fdql_ring_t *r, const float *host_rows, int64_t n, const int32_t *goal_idx_host,
                                  const fdql_episode_vmap_spec_t *sp, int64_t *appended, void *stream) {
  FDQL_REQUIRE(r && host_rows && goal_idx_host && sp && n >= 0, "bad arguments");
  if (appended) *appended = 0;
  if (n == 0) return 0;
  Lock lk(r->mu);
  const int K = sp->K, K1 = K + 1;
  auto f32key = [&](int k) { return k >= 0 && k < r->nkeys && !r->u8[k]; };
  FDQL_REQUIRE(K >= 1 && sp->reward_fn.kind == 0, "append_episode_vmap: K >= 1 and a known reward function");
  FDQL_REQUIRE(f32key(sp->reward_key) && f32key(sp->task_done_key) && r->dims[sp->reward_key] == 1 && r->dims[sp->task_done_key] == 1,
               "append_episode_vmap: reward / task_done must be float32 keys of width 1");
  FDQL_REQUIRE(f32key(sp->achieved_key) && f32key(sp->desired_key) && r->dims[sp->achieved_key] == r->dims[sp->desired_key],
               "append_episode_vmap: achieved_goal / desired_goal keys must exist with equal width");
  const int gd = r->dims[sp->achieved_key];
  FDQL_REQUIRE(f32key(sp->vgoals_key) && r->dims[sp->vgoals_key] == K1 * gd && f32key(sp->vrewards_key) && r->dims[sp->vrewards_key] == K1 &&
                   f32key(sp->vdones_key) && r->dims[sp->vdones_key] == K1,
               "append_episode_vmap: virtual_goals / virtual_rewards / virtual_dones keys of widths (K+1)*g, K+1, K+1");
  FDQL_REQUIRE(sp->vreturn_key < 0 || (f32key(sp->vreturn_key) && r->dims[sp->vreturn_key] == K1 && sp->n_step >= 1),
               "append_episode_vmap: the return key must have width K+1 and n_step >= 1");
  for (int k = 0; k < K; ++k) FDQL_REQUIRE(goal_idx_host[k] >= 0 && goal_idx_host[k] < n, "append_episode_vmap: goal index %d outside the episode", (int)goal_idx_host[k]);
  const int pop = (sp->vreturn_key >= 0 && n > sp->n_step) ? 1 : 0;   // NStepReturnVmap._pop fires once (quirk q3)
  const int64_t n_out = n + pop;
  FDQL_REQUIRE(n_out <= r->maxlen, "append_episode_vmap: %lld rows do not fit a ring of %lld slots", (long long)n_out, (long long)r->maxlen);
  hipStream_t s = (hipStream_t)stream;
  int rc = flush(r, s);
  if (rc) return rc;
  const int F = r->rowfloats;
  if (r->ep_inflight) { FDQL_HIP(hipEventSynchronize(r->ep_done)); r->ep_inflight = false; }
  if (!r->ep_done) FDQL_HIP(hipEventCreateWithFlags(&r->ep_done, hipEventDisableTiming));
  // staging: the rows and, behind them, the K goal indices (one pinned buffer, one H2D copy)
  const int64_t need_rows = n + (K + F - 1) / F;
  if (need_rows > r->ep_in_cap) {
    const int64_t cap = std::max<int64_t>(need_rows, 2 * r->ep_in_cap);
    if (r->ep_pinned) FDQL_HIP(hipHostFree(r->ep_pinned));
    if (r->ep_in) FDQL_HIP(hipFree(r->ep_in));
    r->ep_pinned = nullptr; r->ep_in = nullptr; r->ep_in_cap = 0;
    FDQL_HIP(hipHostMalloc(&r->ep_pinned, cap * F * sizeof(float)));
    FDQL_HIP(hipMalloc(&r->ep_in, cap * F * sizeof(float)));
    r->ep_in_cap = cap;
  }
  if (n_out > r->ep_out_cap) {
    const int64_t cap = std::max<int64_t>(n_out, 2 * r->ep_out_cap);
    if (r->ep_out) FDQL_HIP(hipFree(r->ep_out));
    r->ep_out = nullptr; r->ep_out_cap = 0;
    FDQL_HIP(hipMalloc(&r->ep_out, cap * F * sizeof(float)));
    r->ep_out_cap = cap;
  }
  memcpy(r->ep_pinned, host_rows, n * F * sizeof(float));
  memcpy(r->ep_pinned + n * F, goal_idx_host, K * sizeof(int32_t));
  FDQL_HIP(hipMemcpyAsync(r->ep_in, r->ep_pinned, (n * F + K) * sizeof(float), hipMemcpyHostToDevice, s));
  // the records go to ep_out behind the _pop record's slot; the virtual columns are formed in place
  float *rec0 = r->ep_out + (int64_t)pop * F;
  FDQL_HIP(hipMemcpyAsync(rec0, r->ep_in, n * F * sizeof(float), hipMemcpyDeviceToDevice, s));
  const int *gidx = reinterpret_cast<const int *>(r->ep_in + n * F);
  {
    const int total = (int)n * K1;
    hipLaunchKernelGGL(k_her_vmap, dim3((total + 255) / 256), dim3(256), 0, s, r->ep_in + r->offs[sp->reward_key],
                       r->ep_in + r->offs[sp->task_done_key], r->ep_in + r->offs[sp->achieved_key], r->ep_in + r->offs[sp->desired_key],
                       gidx, (int)n, gd, K, sp->reward_fn, rec0 + r->offs[sp->vgoals_key], rec0 + r->offs[sp->vrewards_key],
                       rec0 + r->offs[sp->vdones_key], (long long)F, (long long)F, (long long)F, (long long)F);
    FDQL_HIP(hipGetLastError());
  }
  if (sp->vreturn_key >= 0) {   // nstep_return_vmap.py:61-74 per column (q10), and _pop's scan of the first n_step records
    hipLaunchKernelGGL(k_mc_return_vmap, dim3((K1 + 63) / 64), dim3(64), 0, s, rec0 + r->offs[sp->vrewards_key],
                       rec0 + r->offs[sp->vdones_key], rec0 + r->offs[sp->vreturn_key], (int)n, K1, sp->gamma, (long long)F, (float *)nullptr);
    if (pop) {
      FDQL_HIP(hipMemcpyAsync(r->ep_out, rec0, F * sizeof(float), hipMemcpyDeviceToDevice, s));   // record 0 again ...
      hipLaunchKernelGGL(k_mc_return_vmap, dim3((K1 + 63) / 64), dim3(64), 0, s, rec0 + r->offs[sp->vrewards_key],
                         rec0 + r->offs[sp->vdones_key], (float *)nullptr, sp->n_step, K1, sp->gamma, (long long)F,
                         r->ep_out + r->offs[sp->vreturn_key]);                                 // ... with the n_step-record return
    }
    FDQL_HIP(hipGetLastError());
  }
  rc = scatter(r, r->ep_out, n_out, r->top, s);
  if (rc) return rc;
  advance(r, n_out);
  FDQL_HIP(hipEventRecord(r->ep_done, s));
  r->ep_inflight = true;
  if (appended) *appended = n_out;
  return 0;
}

int fdql_episode_her_relabel(const float *reward, const float *episode_step, const float *achieved_goal,
                             const float *desired_goal, const float *goal, int32_t n, int32_t goal_dim,
                             const fdql_reward_fn_t *fn, float *reward_out, float *task_done_out,
                             float *episode_step_out, void *stream) {
  FDQL_REQUIRE(reward && episode_step && achieved_goal && desired_goal && goal && fn && reward_out && task_done_out &&
                   episode_step_out && n >= 0 && goal_dim > 0, "bad arguments");
  FDQL_REQUIRE(fn->kind == 0, "unknown reward function kind %d", fn->kind);
  if (n == 0) return 0;
  const HerStrides st = {1, 1, goal_dim, goal_dim, 1, 1, 1, 0};
  hipLaunchKernelGGL(k_her_relabel, dim3(1), dim3(64), 0, (hipStream_t)stream, reward, episode_step, achieved_goal,
                     desired_goal, goal, n, goal_dim, *fn, reward_out, task_done_out, episode_step_out, st);
  FDQL_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
