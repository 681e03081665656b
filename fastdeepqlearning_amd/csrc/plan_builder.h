// The builders of launch stages that the plan (agent_plan.hip) and the test hooks (agent_debug.hip) share: MLP layout helper,
// GEMM-problem / stage builder, chain-program builder, instance helper.
#pragma once
#include "agent_internal.h"

namespace fdql {

inline void add_mlp(fdql_agent *a, MlpDesc &m, const std::string &prefix, int din, const int32_t *hid, int nh, int dout,
             int64_t &top) {
  m.din = din;
  m.dout = dout;
  m.hid.assign(hid, hid + nh);
  for (int i = 0; i < nh; ++i) {
    const int in = m.in_of(i);
    m.w_off.push_back(top);
    a->tensors.push_back({prefix + ".feature_extractor." + std::to_string(i) + ".0.weight", 0, top, m.hid[i], in});
    top += pad4((int64_t)m.hid[i] * in);
    m.b_off.push_back(top);
    a->tensors.push_back({prefix + ".feature_extractor." + std::to_string(i) + ".0.bias", 0, top, m.hid[i], 0});
    top += pad4(m.hid[i]);
  }
  m.hw_off = top;
  a->tensors.push_back({prefix + ".head.weight", 0, top, dout, m.head_ld()});
  top += pad4((int64_t)dout * m.head_ld());
  m.hb_off = top;
  a->tensors.push_back({prefix + ".head.bias", 0, top, dout, 0});
  top += pad4(dout);
}

// --------------------------------------------------------------------------- plan builder

struct Builder {
  fdql_agent *a;
  std::vector<Stage> &st;
  Builder(fdql_agent *ag) : a(ag), st(ag->stages) {}
  // dense 256 x 256 weight-gradient blocks: candidates for the output-stationary launch (wgrad.h), each with the index of the
  // stage it rides in otherwise (-1: the tail stage)
  std::vector<std::pair<GemmProblem, int>> wg_cand;
  // The candidates gathered so far become ONE output-stationary launch (stage `name`, appended) when there are enough row
  // tiles for every workgroup to amortise its 256 KiB partial result; else they ride in their host stages / `fallback`.
  void flush_wgrad_stat(const std::string &name, Stage &fallback) {
    if (wg_cand.empty()) return;
    std::vector<GemmProblem> probs;
    for (auto &pc : wg_cand) probs.push_back(pc.first);
    Stage wst;
    wst.kind = ST_WGRAD_STAT; wst.name = name;
    if (a->wgrad_stat_pays((long long)probs.size()) && wgrad_stat_from_problems(probs.data(), (int)probs.size(), a->nsplit, a->n_train, wst.wga)) {
      // the few-column / few-row gradients that share an operand with one of the blocks (a critic's action columns, its
      // skip head's rows over the state and over h0) ride with it instead of re-reading the operand in the tail launches
      for (size_t i = 0; i < fallback.gemm.size();) {
        bool taken = false;
        for (int k = 0; k < wst.wga.ninst && !taken; ++k) taken = wgrad_stat_add_rider(wst.wga, k, fallback.gemm[i]);
        if (taken) fallback.gemm.erase(fallback.gemm.begin() + i);
        else ++i;
      }
      wgrad_stat_balance(wst.wga);
      wst.flops = wgrad_stat_flops(wst.wga);
      wst.bytes = 8.0 * wst.wga.M * WG_N * wst.wga.ninst;
      st.push_back(wst);
    } else {
      for (auto &pc : wg_cand) (pc.second >= 0 ? st[pc.second] : fallback).gemm.push_back(pc.first);
    }
    wg_cand.clear();
  }

  Stage &gemm_stage(const std::string &name) {
    st.emplace_back();
    st.back().kind = ST_GEMM;
    st.back().name = name;
    return st.back();
  }
  Stage &func_stage(const std::string &name, std::function<hipError_t(hipStream_t)> fn, int phase = FDQL_PHASE_GRAD) {
    st.emplace_back();
    st.back().kind = ST_FUNC;
    st.back().name = name;
    st.back().fn = std::move(fn);
    st.back().phase = phase;
    return st.back();
  }

  static GemmProblem new_gemm(int M, int N, float *C, int ldc) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.ksplit = 1; p.emit_seg = -1;
    return p;
  }
  static void add_seg(GemmProblem &p, const float *A, int lda, int a_kc, const float *B, int ldb, int b_kc, int K) {
    if (K <= 0) return;
    GemmSeg &s = p.seg[p.nseg++];
    s.A = A; s.lda = lda; s.a_kc = a_kc; s.B = B; s.ldb = ldb; s.b_kc = b_kc; s.K = K;
  }

  // forward hidden layer i of inst -> problem
  GemmProblem fwd_layer(const MlpInst &m, int i) {
    const MlpDesc &d = *m.d;
    GemmProblem p = new_gemm(m.rows, d.hid[i], m.h[i], d.hid[i]);
    if (i == 0) {
      int col = 0;
      for (const SegIn &s : m.in) {
        add_seg(p, s.ptr, s.ld, 1, m.W(0) + col, d.din, 1, s.width);
        col += s.width;
      }
    } else {
      add_seg(p, m.h[i - 1], d.hid[i - 1], 1, m.W(i), d.hid[i - 1], 1, d.hid[i - 1]);
    }
    p.bias = m.Bv(i);
    p.epi = EPI_LRELU;
    if ((size_t)i < m.gm.size()) p.gm_out = m.gm[i];
    return p;
  }
  // head: out = W_head cat(in, h_0..h_{n-1}) + b   (mlp.py:93-94); narrow heads get the 128x32 tile
  GemmProblem fwd_head(const MlpInst &m) {
    const MlpDesc &d = *m.d;
    const int ld = d.head_ld();
    GemmProblem p = new_gemm(m.rows, d.dout, m.out, m.ldout);
    int col = 0;
    for (const SegIn &s : m.in) { add_seg(p, s.ptr, s.ld, 1, m.HW() + col, ld, 1, s.width); col += s.width; }
    for (size_t i = 0; i < d.hid.size(); ++i) { add_seg(p, m.h[i], d.hid[i], 1, m.HW() + col, ld, 1, d.hid[i]); col += d.hid[i]; }
    p.bias = m.HB();
    return p;
  }
  // the head restricted to the MLP's inputs (the hidden activations' part comes from the fused partial sums)
  GemmProblem fwd_head_inputs_only(const MlpInst &m) {
    const MlpDesc &d = *m.d;
    const int ld = d.head_ld();
    GemmProblem p = new_gemm(m.rows, d.dout, m.out, m.ldout);
    int col = 0;
    for (const SegIn &s : m.in) { add_seg(p, s.ptr, s.ld, 1, m.HW() + col, ld, 1, s.width); col += s.width; }
    p.bias = m.HB();
    return p;
  }
  int head_col_of_hidden(const MlpDesc &d, int i) const {
    int col = d.din;
    for (int j = 0; j < i; ++j) col += d.hid[j];
    return col;
  }
  // dpre_i = (dY Wh[:, cols(h_i)] + dpre_{i+1} W_{i+1}) * lrelu'(h_i)
  GemmProblem bwd_dpre(const MlpInst &m, int i, const float *dY, int lddy) {
    const MlpDesc &d = *m.d;
    GemmProblem p = new_gemm(m.rows, d.hid[i], m.dpre[i], d.hid[i]);
    add_seg(p, dY, lddy, 1, m.HW() + head_col_of_hidden(d, i), d.head_ld(), 0, d.dout);
    if (i + 1 < (int)d.hid.size()) add_seg(p, m.dpre[i + 1], d.hid[i + 1], 1, m.W(i + 1), d.hid[i], 0, d.hid[i + 1]);
    p.epi = EPI_LRELU_GRAD;
    p.ref = m.h[i];
    p.ldref = d.hid[i];
    p.colsum = m.dpre_cs[i];
    if ((size_t)i < m.gm.size()) p.gm_ref = m.gm[i];
    return p;
  }
  // last hidden layer under a narrow head: the rank-dout outer product as a streaming kernel
  static bool narrow_head_last(const MlpDesc &d, int i) {
    return i + 1 == (int)d.hid.size() && d.dout <= HEAD_DGRAD_MAXQ;
  }
  HeadDgradProblem bwd_dpre_head(const MlpInst &m, int i, const float *dY, int lddy) {
    const MlpDesc &d = *m.d;
    HeadDgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = m.rows; p.N = d.hid[i]; p.Q = d.dout;
    p.dY = dY; p.lddy = lddy;
    p.Wh = m.HW() + head_col_of_hidden(d, i); p.ldw = d.head_ld();
    p.h = m.h[i]; p.dpre = m.dpre[i]; p.colsum = m.dpre_cs[i];
    return p;
  }
  // K-segments of d(input columns [col, col+width)) = dY Wh[:, cols] + dpre_0 W_0[:, cols]
  void input_grad_segs(const MlpInst &m, const float *dY, int lddy, int col, GemmProblem &p) {
    const MlpDesc &d = *m.d;
    add_seg(p, dY, lddy, 1, m.HW() + col, d.head_ld(), 0, d.dout);
    if (!d.hid.empty()) add_seg(p, m.dpre[0], d.hid[0], 1, m.W(0) + col, d.din, 0, d.hid[0]);
  }
  // dW[nout, width] (slabs) = dOut[R, nout]^T X[R, width], K-split over the R rows
  void wgrad_gemm(int R, const float *dOut, int ldo, int nout, const float *X, int ldx, int width, float *dst, int ldw,
                  Stage &gs, Stage &narrow) {
    if (width <= 0) return;
    GemmProblem p = new_gemm(nout, width, dst, ldw);
    add_seg(p, dOut, ldo, 0, X, ldx, 0, R);
    p.ksplit = a->nsplit;
    p.split_stride = a->n_train;
    if (wgrad_stat_takes(p)) {   // decided once every weight gradient of the plan is known (build_plan)
      int host = -1;
      for (size_t i = 0; i < st.size(); ++i)
        if (&st[i] == &gs) host = (int)i;
      wg_cand.push_back({p, host});
      return;
    }
    // narrow problems (other tile shapes = other launches) are pooled in one stage at the end
    // (33..36 outputs over a 256-wide input - the 2 x 17 logits of config 4's actor head - pick a square tile shape, 20 TF for an
    // HBM-bound product: they go with the narrow ones, where the streaming launch takes them)
    const bool streams = nout > 32 && nout <= STREAM_WGRAD_MAX_OUT && width == 256 && ldx == 256;
    (gemm_shape_is_dense(gemm_pick_shape(p, gemm_dense_shape())) && !streams ? gs : narrow).gemm.push_back(p);
  }
  // bias gradient = column sums of dOut over R rows, from per-64-row partials `cs` when the dgrad GEMM left them
  void wgrad_bias(int R, const float *dOut, int ldo, int nout, const float *cs, float *dst, Stage &ws, int cs_rows = 0) {
    SkinnyWgradProblem p;
    memset(&p, 0, sizeof(p));
    p.Nout = 1; p.K = nout; p.dY = nullptr;
    if (cs) { p.M = cs_rows > 0 ? cs_rows : (R + 63) / 64; p.X = cs; p.ldx = nout; }
    else { p.M = R; p.X = dOut; p.ldx = ldo; }
    p.dW = dst; p.sq = 0; p.sk = 1; p.split_stride = a->n_train; p.nsplit = a->nsplit;
    ws.swg.push_back(p);
  }
  // The narrow weight gradients left in `from` after the riders were dealt (flush_wgrad_stat) that have the streaming
  // form - a few outputs over a 256-wide input, or a few input columns under a 256-wide output gradient (roles swapped) -
  // move to the one-launch streaming stage `to` (kernels.hip, k_stream_wgrad); FDQL_STREAM_WGRAD=0: none.
  // The ones that are narrow BOTH ways (a skip head's rows over the action columns: 2 x 6) join the column sums' launch
  // (`skinny`, k_skinny_wgrad: any K) - left on the tile kernel they were a 40 us launch of their own for 2 MB.
  void take_stream_wgrads(Stage &from, Stage &to, Stage &skinny) {
    const int mode = plan_switches().stream_wgrad;
    if (mode == 0) return;
    // few rows (temporal_len 2): the narrow gradients stay GEMM problems of the tail stage, which the small-batch kernel takes
    // in its one launch - a streaming launch of their own is 15 us of a 0.28 ms step (0.277 -> 0.264 ms)
    if (mode != 2 && a->small_max_tiles > 0 && a->M <= 1024 && gemm_dense_shape() == GEMM_64x64)
      return;
    // Short K-split slabs (a few hundred rows, config 2: the tile kernels' narrow launches are one memory round trip per K
    // iteration there, 0.7-2.4 TB/s): everything that has the form.  Long slabs (config 4 at B = 1024: 1568 rows): the
    // 32x128 tile streams the head rows at 5 TB/s - better than the 4.4 TB/s here - but the 128x32 launch of the few-input-
    // column gradients does 1.8 TB/s: only those move, and the tiny ones stay where they are (the column sums' launch
    // would take them one column per thread).  FDQL_STREAM_WGRAD=2: everything, whatever the slab length (tests).
    const bool all = mode == 2 || a->M / a->nsplit <= STREAM_WGRAD_MAX_SLAB_ROWS;
    for (size_t i = 0; i < from.gemm.size();) {
      const GemmProblem &p = from.gemm[i];
      SkinnyWgradProblem q;
      memset(&q, 0, sizeof(q));
      bool ok = p.nseg == 1 && p.ksplit == a->nsplit && p.split_stride == a->n_train && !p.bias && p.epi == EPI_NONE && !p.colsum && !p.C2 &&
                !p.seg[0].a_kc && !p.seg[0].b_kc;
      if (ok && p.M <= SKINNY_MAX_OUT && p.N <= 64) {   // narrow both ways
        const GemmSeg &sg = p.seg[0];
        q.M = sg.K; q.Nout = p.M; q.K = p.N; q.dY = sg.A; q.lddy = sg.lda; q.X = sg.B; q.ldx = sg.ldb;
        q.dW = p.C; q.sq = p.ldc; q.sk = 1; q.split_stride = p.split_stride; q.nsplit = p.ksplit;
        const bool tiny = q.Nout <= 2 && q.K <= 8;   // (k_skinny_wgrad's threads-over-rows path: 0.001 ms beside the column sums)
        if (tiny && !all) { ++i; continue; }
        if (!tiny && !stream_wgrad_takes(q)) { ++i; continue; }
        (tiny ? skinny : to).swg.push_back(q);
        from.gemm.erase(from.gemm.begin() + i);
        continue;
      }
      if (ok) {
        const GemmSeg &sg = p.seg[0];   // dW[nout = p.M][width = p.N] = dOut[R, nout]^T X[R, width]
        q.M = sg.K; q.K = 256; q.ldx = 256; q.dW = p.C; q.split_stride = p.split_stride; q.nsplit = p.ksplit;
        // (33..36 outputs - config 4's 2 x 17 logits - are two row tiles on the 32-row tile shape and a 20 TF launch on the square
        // one: they stream whatever the slab length)
        if ((all || p.M > 32) && p.N == 256 && sg.ldb == 256 && p.M <= STREAM_WGRAD_MAX_OUT) {   // few outputs over a 256-wide input
          q.Nout = p.M; q.dY = sg.A; q.lddy = sg.lda; q.X = sg.B; q.sq = p.ldc; q.sk = 1;
        } else if (p.M == 256 && sg.lda == 256 && p.N <= 32) {   // few input columns: dW^T[a][n] = X2[R, a]^T dOut[R, n]
          q.Nout = p.N; q.dY = sg.B; q.lddy = sg.ldb; q.X = sg.A; q.sq = 1; q.sk = p.ldc;
        } else {
          ok = false;
        }
        ok = ok && stream_wgrad_takes(q);
      }
      if (ok) {
        to.swg.push_back(q);
        from.gemm.erase(from.gemm.begin() + i);
      } else {
        ++i;
      }
    }
  }
  // The column sums (bias gradients) and the tiny both-ways-narrow gradients of `skinny` that the streaming kernel takes join
  // its launch when there is one: a launch less at the end of the step (FDQL_NO_COLSUM_STREAM: k_skinny_wgrad keeps them).
  void colsums_into_stream(Stage &skinny, Stage &to) {
    if (to.swg.empty() || !plan_switches().colsum_stream) return;
    std::vector<SkinnyWgradProblem> first;   // (short waves: dispatched ahead of the streaming ones, they end under them)
    for (size_t i = 0; i < skinny.swg.size();) {
      if (stream_wgrad_takes(skinny.swg[i])) {
        first.push_back(skinny.swg[i]);
        skinny.swg.erase(skinny.swg.begin() + i);
      } else {
        ++i;
      }
    }
    to.swg.insert(to.swg.begin(), first.begin(), first.end());
  }
  // all weight / bias gradients of one MLP instance into the K-split slabs.  Every weight
  // gradient is a K-split GEMM (the narrow ones on the 128x32 / 32x128 tiles); bias gradients
  // come from the per-tile column sums the dgrad GEMMs leave behind (dy_cs / dpre_cs), or
  // directly from dY when dY is narrow (dz, d logits).
  void wgrads(const MlpInst &m, const float *dY, int lddy, const float *dy_cs, Stage &gs, Stage &narrow, Stage &ws, int dy_cs_rows = 0) {
    const MlpDesc &d = *m.d;
    float *slab = a->buf("slabs");
    const long long P = a->n_train;
    const int S = a->nsplit, R = m.rows;
    auto gemm_w = [&](const float *dOut, int ldo, int nout, const float *X, int ldx, int width, float *dst, int ldw) {
      wgrad_gemm(R, dOut, ldo, nout, X, ldx, width, dst, ldw, gs, narrow);
    };
    auto bias_w = [&](const float *dOut, int ldo, int nout, const float *cs, float *dst, int cs_rows = 0) {
      wgrad_bias(R, dOut, ldo, nout, cs, dst, ws, cs_rows);
    };
    for (size_t i = 0; i < d.hid.size(); ++i) {
      float *dst = slab + d.w_off[i];
      if (i == 0) {
        int col = 0;
        for (const SegIn &s : m.in) { gemm_w(m.dpre[0], d.hid[0], d.hid[0], s.ptr, s.ld, s.width, dst + col, d.din); col += s.width; }
      } else {
        gemm_w(m.dpre[i], d.hid[i], d.hid[i], m.h[i - 1], d.hid[i - 1], d.hid[i - 1], dst, d.hid[i - 1]);
      }
      bias_w(m.dpre[i], d.hid[i], d.hid[i], m.dpre_cs[i], slab + d.b_off[i], i < m.dpre_cs_rows.size() ? m.dpre_cs_rows[i] : 0);
    }
    float *dst = slab + d.hw_off;
    const int ld = d.head_ld();
    int col = 0;
    for (const SegIn &s : m.in) { gemm_w(dY, lddy, d.dout, s.ptr, s.ld, s.width, dst + col, ld); col += s.width; }
    for (size_t i = 0; i < d.hid.size(); ++i) { gemm_w(dY, lddy, d.dout, m.h[i], d.hid[i], d.hid[i], dst + col, ld); col += d.hid[i]; }
    bias_w(dY, lddy, d.dout, dy_cs, slab + d.hb_off, dy_cs_rows);
  }
};

// ------------------------------------------------------------------------ chain programs (chain.h)
// Emits the operations of MLP forward passes for the row-block chain kernel.  `ok` turns false as soon as something
// does not fit the kernel (a layer wider than 256 columns, more K-segments than an operation holds, LDS exhausted,
// too many operations): the caller then drops the stage and keeps the per-layer GEMM launches.
struct ChainImg { int slot = -1, pitch = 0, K = 0, size = 0; };
struct RowWin { int lo, hi, shift; };

struct ChainBuilder {
  Stage &st;
  bool ok = true;
  int op_start = 0, rows = 0, peak = 0;
  int stage_top = CH_LDS_FLOATS;      // weight staging areas of the CH_NARROW operations: carved downwards from the top of
                                      // LDS, alive for the whole program (they are filled before its first operation)
  std::vector<std::pair<int, int>> used;   // live LDS ranges (offset, size) of the program being built
  int bm;                             // rows per workgroup: images are [bm][pitch]
  explicit ChainBuilder(Stage &s) : st(s), bm(s.chain_bm) {}

  void begin(int nrows) { rows = nrows; op_start = (int)st.cops.size(); used.clear(); peak = 0; stage_top = CH_LDS_FLOATS; }
  void end() {
    ChainOp e;
    memset(&e, 0, sizeof(e));
    e.kind = CH_END;
    st.cops.push_back(e);
    if ((int)st.cops.size() - op_start > CH_MAX_OPS) ok = false;
    ChainProblem p;
    memset(&p, 0, sizeof(p));
    const int need = stage_top < CH_LDS_FLOATS ? CH_LDS_FLOATS : peak;   // staging areas sit at the top of the budget
    p.rows = rows; p.op_start = op_start; p.nops = (int)st.cops.size() - op_start; p.lds_floats = need;
    st.cprobs.push_back(p);
    st.lds_floats = std::max(st.lds_floats, need);
  }
  int alloc(int size) {   // first fit; sizes are multiples of 4 floats (16-byte aligned images)
    size = (size + 3) & ~3;
    std::sort(used.begin(), used.end());
    int at = 0;
    for (auto &u : used) {
      if (u.first - at >= size) break;
      at = u.first + u.second;
    }
    if (at + size > stage_top) { ok = false; return 0; }
    used.push_back({at, size});
    peak = std::max(peak, at + size);
    return at;
  }
  void release(const ChainImg &im) {
    for (size_t i = 0; i < used.size(); ++i)
      if (used[i].first == im.slot) { used.erase(used.begin() + i); return; }
  }
  ChainImg image(int K, int min_size = 0) {
    ChainImg im;
    im.K = K; im.pitch = chain_pitch(K); im.size = std::max(bm * im.pitch, min_size);
    im.slot = alloc(im.size);
    return im;
  }
  static ChainOp new_op(int kind) {
    ChainOp o;
    memset(&o, 0, sizeof(o));
    o.kind = kind; o.out_slot = -1;
    return o;
  }
  ChainImg load(const std::vector<SegIn> &segs, int min_size = 0) {
    int K = 0;
    for (auto &s : segs) K += s.width;
    ChainImg im = image(K, min_size);
    ChainOp o = new_op(CH_LOAD);
    if ((int)segs.size() > CH_MAX_SEG) { ok = false; return im; }
    o.slot = im.slot; o.pitch = im.pitch; o.kpad = chain_kpad(K); o.nseg = (int)segs.size();
    int col = 0;
    for (size_t i = 0; i < segs.size(); ++i) {
      o.ld[i].src = segs[i].ptr; o.ld[i].ld = segs[i].ld; o.ld[i].width = segs[i].width; o.ld[i].col = col;
      col += segs[i].width;
    }
    st.cops.push_back(o);
    return im;
  }
  // one Linear layer over cat(ins): W rows of pitch ldw, the k-th input image reads columns starting at its offset
  // in the concatenation.  dst: the LDS image that receives the result (may alias a dying input; slot -1: none)
  void gemm(const std::vector<ChainImg> &ins, const float *W, int ldw, int N, const float *bias, int act, const ChainImg &dst,
            float *out, int ldo, RowWin win, const float *rider_w = nullptr, int rider_ld = 0, int rider_n = 0, bool rider_begin = false) {
    if (N > 256 || (int)ins.size() > CH_MAX_SEG) { ok = false; return; }
    ChainOp o = new_op(CH_GEMM);
    o.N = N; o.flags = CHF_ZERO | CHF_EMIT | (rider_w && rider_begin ? CHF_HBEGIN : 0); o.act = act; o.bias = bias; o.nseg = (int)ins.size();
    o.hw = rider_w; o.hldw = rider_ld; o.hN = rider_n;
    if (rider_w) st.flops += 2.0 * (std::min(rows, win.hi) - win.lo) * (double)rider_n * [&] { int k = 0; for (auto &im : ins) k += im.K; return k; }();
    int col = 0;
    for (size_t i = 0; i < ins.size(); ++i) {
      o.seg[i].W = W + col; o.seg[i].ldw = ldw; o.seg[i].slot = ins[i].slot; o.seg[i].pitch = ins[i].pitch; o.seg[i].K = ins[i].K;
      col += ins[i].K;
    }
    o.out_slot = dst.slot; o.out_pitch = dst.pitch;
    o.out = out; o.ldo = ldo; o.row_lo = win.lo; o.row_hi = win.hi; o.row_shift = win.shift;
    st.cops.push_back(o);
    st.flops += chain_op_flops(o, std::min(rows, win.hi) - win.lo);
  }
  // part of a narrow head (N <= 32) over cat(ins) starting at column `col0` of the head weight
  void narrow(const std::vector<ChainImg> &ins, const float *W, int ldw, int col0, int N, bool begin, bool finish,
              const float *bias, float *out, int ldo, RowWin win, int scratch_slot) {
    if (N > 32 || (int)ins.size() > CH_MAX_SEG) { ok = false; return; }
    ChainOp o = new_op(CH_NARROW);
    o.N = N; o.flags = (begin ? CHF_BEGIN : 0) | (finish ? CHF_FINISH : 0); o.nseg = (int)ins.size();
    int col = col0;
    for (size_t i = 0; i < ins.size(); ++i) {
      o.seg[i].W = W + col; o.seg[i].ldw = ldw; o.seg[i].slot = ins[i].slot; o.seg[i].pitch = ins[i].pitch; o.seg[i].K = ins[i].K;
      col += ins[i].K;
    }
    int need = 0;
    for (auto &im : ins) need += N * (((im.K + 15) & ~15) + 4);
    need = (need + 3) & ~3;
    stage_top -= need;
    if (stage_top < peak) ok = false;   // (images allocated later are checked against stage_top in alloc())
    for (auto &u : used) if (u.first + u.second > stage_top) ok = false;
    (void)scratch_slot;
    o.slot = stage_top; o.bias = bias; o.out = out; o.ldo = ldo;
    o.row_lo = win.lo; o.row_hi = win.hi; o.row_shift = win.shift;
    st.cops.push_back(o);
    st.flops += chain_op_flops(o, std::min(rows, win.hi) - win.lo);
  }

  // SkipHeadMLP forward (mlp.py:88-94) on images already in LDS.
  //   in_dies: the input images are not needed after this MLP (their LDS may be reused).
  //   hidden activations go to m.h[i] (global, rows of `win`) when store_h; the head output to m.out (global) and,
  //   for a wide head, to the returned LDS image.
  // Narrow head (dout <= 32): the head's dot product is accumulated piecewise (CH_NARROW, per-wave 16-row tiles) as soon as
  // each block of its input exists, so a layer's input image can be overwritten in place by its output: one image per MLP.
  ChainImg mlp(const MlpInst &m, std::vector<ChainImg> ins, bool in_dies, bool store_h, RowWin win, bool want_out_image) {
    const MlpDesc &d = *m.d;
    const int nh = (int)d.hid.size(), ld_head = d.head_ld();
    ChainImg none;
    if (d.dout <= 32 && !want_out_image) {   // (an output that feeds the next MLP from LDS takes the GEMM path)
      // The head's block over a layer's INPUT rides in that layer's K loop when the layer is wide enough for every wave to
      // own a column tile (chain.hip, RIDER); otherwise it is a CH_NARROW pass of its own.  The block over the last
      // hidden activation always is one (nothing follows it to ride in).
      auto rides = [&](int i) { return i < nh && d.hid[i] > 192; };
      bool begun = false;
      if (!rides(0)) { narrow(ins, m.HW(), ld_head, 0, d.dout, true, nh == 0, m.HB(), m.out, m.ldout, win, 0); begun = true; }
      int col = 0;
      std::vector<ChainImg> cur = ins;
      bool cur_dies = in_dies;
      for (int i = 0; i < nh; ++i) {
        // the layer's output image: in place of its (single, dying) input when possible, else a new one
        const int need = bm * chain_pitch(d.hid[i]);
        ChainImg dst;
        if (cur_dies && cur.size() == 1 && cur[0].size >= need) {
          dst = cur[0];
          dst.K = d.hid[i]; dst.pitch = chain_pitch(d.hid[i]);
        } else {
          if (cur_dies) for (auto &c : cur) release(c);
          dst = image(d.hid[i], need);
        }
        const bool ride = rides(i);
        gemm(cur, m.W(i), d.in_of(i), d.hid[i], m.Bv(i), CHA_LRELU, dst, store_h ? m.h[i] : nullptr, d.hid[i], win,
             ride ? m.HW() + col : nullptr, ld_head, d.dout, ride && !begun);
        if (ride) begun = true;
        col += d.in_of(i);
        const bool last = i + 1 == nh;
        if (last || !rides(i + 1)) narrow({dst}, m.HW(), ld_head, col, d.dout, !begun, last, m.HB(), m.out, m.ldout, win, 0);
        begun = true;
        cur = {dst};
        cur_dies = true;
      }
      if (nh > 0) release(cur[0]);
      else if (in_dies) for (auto &c : ins) release(c);
      return none;
    }
    // wide head: every feature block stays in LDS until the head GEMM has read it
    std::vector<ChainImg> feats = ins, cur = ins;
    for (int i = 0; i < nh; ++i) {
      ChainImg dst = image(d.hid[i]);
      gemm(cur, m.W(i), d.in_of(i), d.hid[i], m.Bv(i), CHA_LRELU, dst, store_h ? m.h[i] : nullptr, d.hid[i], win);
      feats.push_back(dst);
      cur = {dst};
    }
    // the head's output image may reuse what dies here: the epilogue writes only after every wave has finished reading
    for (size_t i = in_dies ? 0 : ins.size(); i < feats.size(); ++i) release(feats[i]);
    ChainImg dst;
    if (want_out_image) dst = image(d.dout);
    gemm(feats, m.HW(), ld_head, d.dout, m.HB(), CHA_NONE, dst, m.out, m.ldout, win);
    return dst;
  }
};

inline MlpInst make_inst(fdql_agent *a, const MlpDesc &d, const std::string &p, const float *wbase, int64_t worigin, int rows,
                  bool bwd) {
  MlpInst m;
  m.d = &d;
  m.wbase = wbase;
  m.worigin = worigin;
  m.rows = rows;
  for (size_t i = 0; i < d.hid.size(); ++i) {
    m.h.push_back(a->buf(p + ".h" + std::to_string(i)));
    if (bwd) {
      m.dpre.push_back(a->buf(p + ".dpre" + std::to_string(i)));
      m.dpre_cs.push_back(a->buf(p + ".cs" + std::to_string(i)));
      m.dpre_cs_rows.push_back(0);
    }
    if (a->has_buf(p + ".gm" + std::to_string(i))) m.gm.push_back(reinterpret_cast<unsigned *>(a->buf(p + ".gm" + std::to_string(i))));
  }
  return m;
}

// Does the weight-stationary row-block kernel take a group of like problems (one launch)?  Else the tile kernels do.
// Deterministic in (problems, environment): the plan builder asks the same question where the answer changes what other
// stages read (the dgrad form's column sums).  (Round 5: the streamed-weights row-block kernel of round 2, k_rowgemm, is gone -
// it had been the fallback behind FDQL_WSTAT=0 since round 3 and lost every measurement since.)
inline bool rows_launch_of(const fdql_agent *a, const std::vector<GemmProblem> &grp, RowsLaunch &rl) {
  const long long tiles = (long long)grp.size() * (grp[0].M / ROWS_BM);
  if (tiles < a->rows_min_tiles) return false;
  rl.ws = wstat_from_problems(grp.data(), (int)grp.size(), rl.wa);
  return rl.ws;
}

}  // namespace fdql
