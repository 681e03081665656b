// Test and diagnostic hooks of the C ABI (include/fdql.h: fdql_test_*, fdql_debug_*): single launches of the kernel families
// through the same builders the plan uses (plan_builder.h).
#include "plan_builder.h"

extern "C" {

int fdql_debug_set_gemm_dense_shape(int32_t shape) {
  FDQL_REQUIRE(gemm_shape_is_dense(shape) || shape == GEMM_SMALL,
               "dense shape must be 0 (128x128), 3 (64x128), 5 (64x64) or 7 (the small-batch kernel wherever it has the form)");
  gemm_set_dense_shape(shape);
  return 0;
}

int fdql_debug_chain_stamps(uint64_t *out, int32_t cap) {
  if (!out) { chain_enable_stamps(cap != 0); return 0; }   // (out == NULL: switch the recording on / off)
  return chain_read_stamps(reinterpret_cast<unsigned long long *>(out), cap);
}

int fdql_test_chain_mlp(const float *x, int32_t rows, int32_t din, const int32_t *hid, int32_t nh, int32_t dout,
                        const float *weights, float *const *h_out, float *out, void *stream) {
  plan_switches_refresh();
  FDQL_REQUIRE(x && hid && weights && out && rows > 0 && din > 0 && dout > 0 && nh >= 0 && nh <= FDQL_MAX_HIDDEN, "bad arguments");
  fdql_agent tmp;
  MlpDesc d;
  int64_t top = 0;
  add_mlp(&tmp, d, "test", din, hid, nh, dout, top);
  MlpInst m;
  m.d = &d; m.wbase = weights; m.worigin = 0; m.rows = rows;
  m.in.push_back({x, din, din});
  for (int i = 0; i < nh; ++i) m.h.push_back(h_out ? h_out[i] : nullptr);
  m.out = out; m.ldout = dout;
  Stage cs;
  cs.kind = ST_CHAIN; cs.name = "test";
  ChainBuilder cb(cs);
  cb.begin(rows);
  ChainImg xi = cb.load(m.in);
  cb.mlp(m, {xi}, true, h_out != nullptr, {0, rows, 0}, false);
  cb.end();
  FDQL_REQUIRE(cb.ok, "this MLP does not fit the chain kernel");
  const int blocks = chain_finalize(cs.cprobs.data(), (int)cs.cprobs.size(), cs.chain_bm);
  void *dev = nullptr;
  const size_t pb = cs.cprobs.size() * sizeof(ChainProblem), ob = cs.cops.size() * sizeof(ChainOp);
  FDQL_HIP(hipMalloc(&dev, pb + ob + 256));
  FDQL_HIP(hipMemcpy(dev, cs.cprobs.data(), pb, hipMemcpyHostToDevice));
  void *odev = (char *)dev + (pb + 255) / 256 * 256;
  FDQL_HIP(hipMemcpy(odev, cs.cops.data(), ob, hipMemcpyHostToDevice));
  hipError_t e = chain_launch((const ChainProblem *)dev, (int)cs.cprobs.size(), (const ChainOp *)odev, blocks, cs.lds_floats, cs.chain_bm, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("chain launch: %s", hipGetErrorString(e)); (void)hipFree(dev); return FDQL_EHIP; }
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipFree(dev));
  return 0;
}

int fdql_test_gemm(const float *A, int32_t lda, int32_t a_kc, const float *B, int32_t ldb, int32_t b_kc,
                   const float *bias, float *C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                   const float *ref, int32_t ldref, int32_t ksplit, void *stream) {
  plan_switches_refresh();
  GemmProblem p;
  memset(&p, 0, sizeof(p));
  p.emit_seg = -1;
  p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.ksplit = ksplit < 1 ? 1 : ksplit; p.split_stride = (long long)M * ldc;
  p.bias = bias; p.epi = epilogue; p.ref = ref; p.ldref = ldref;
  p.nseg = 1;
  p.seg[0].A = A; p.seg[0].lda = lda; p.seg[0].a_kc = a_kc; p.seg[0].B = B; p.seg[0].ldb = ldb; p.seg[0].b_kc = b_kc; p.seg[0].K = K;
  const int shape = gemm_pick_shape(p, gemm_dense_shape());
  const int blocks = gemm_finalize(&p, 1, shape);
  GemmProblem *dev = nullptr;
  FDQL_HIP(hipMalloc(&dev, sizeof(p)));
  FDQL_HIP(hipMemcpy(dev, &p, sizeof(p), hipMemcpyHostToDevice));
  hipError_t e = gemm_launch(dev, 1, blocks, shape, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("gemm launch: %s", hipGetErrorString(e)); (void)hipFree(dev); return FDQL_EHIP; }
  FDQL_HIP(hipStreamSynchronize((hipStream_t)stream));
  FDQL_HIP(hipFree(dev));
  return 0;
}

/* Test hook for the output-stationary weight-gradient kernel (wgrad.hip): nprob blocks dW[i] [256, ldw] (slab 0 at dW + i *
 * 256 * ldw, slabs slab_stride floats apart, nslab of them) = G[i]^T X[i] over M rows each (G, X: [nprob * M, 256]).  Every
 * slab of every block is written (partials or zeros): their sum is the gradient.  FDQL_EINVAL: the kernel does not take the form. */
int fdql_test_conv(int32_t mode, const void *in, int32_t u8, const int32_t *slots,
                   const float *W, const float *bias, const float *dpre, const float *act_prev, float *out, float *scratch,
                   int64_t scratch_floats, int64_t nimg, int32_t C, int32_t H, int32_t Wd, int32_t k, int32_t s, int32_t cout,
                   void *stream) {
  plan_switches_refresh();
  ConvGeom g;
  g.C = C; g.H = H; g.W = Wd; g.k = k; g.s = s;
  FDQL_REQUIRE(k > 0 && s > 0 && H >= k && Wd >= k && nimg > 0, "fdql_test_conv: bad geometry");
  g.OH = (H - k) / s + 1; g.OW = (Wd - k) / s + 1;
  ConvSrc src;
  src.base = in; src.u8 = u8; src.slots = slots;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    FDQL_REQUIRE(conv_fwd_takes(g, cout, u8 != 0), "fdql_test_conv: no forward kernel for this layer");
    ConvFwdArgs a;
    a.in = src; a.W = W; a.bias = bias; a.out = out; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_fwd_launch(a, st));
  } else if (mode == 1) {
    FDQL_REQUIRE(!u8 && conv_dgrad_takes(g, cout), "fdql_test_conv: no data-gradient kernel for this layer");
    ConvDgradArgs a;
    a.dpre = dpre; a.W = W; a.act_prev = act_prev; a.dprev = out; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_dgrad_launch(a, st));
  } else if (mode == 2) {
    const int nslab = conv_wgrad_slabs(g, cout, u8 != 0, nimg);
    const long long n = (long long)cout * C * k * k + cout;
    FDQL_REQUIRE(nslab > 0, "fdql_test_conv: no weight-gradient kernel for this layer");
    FDQL_REQUIRE(scratch_floats >= nslab * n, "fdql_test_conv: scratch of %lld floats, %lld needed", (long long)scratch_floats, nslab * n);
    ConvWgradArgs a;
    a.in = src; a.dpre = dpre; a.wpart = scratch; a.nimg = nimg; a.g = g; a.cout = cout;
    FDQL_HIP(conv_wgrad_launch(a, st));
    FDQL_HIP(reduce_partials_launch(scratch, nslab, n, out, st));
  } else {
    FDQL_REQUIRE(false, "fdql_test_conv: unknown mode %d", (int)mode);
  }
  return 0;
}

int fdql_test_wgrad_stat(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                         int64_t slab_stride, void *stream) {
  return fdql_test_wgrad_stat_riders(G, X, dW, M, nprob, ldw, nslab, slab_stride, nullptr, 0, 0, nullptr, 0, 1, nullptr, 0, 0, nullptr, 0, stream);
}

int fdql_test_wgrad_stat_riders(const float *G, const float *X, float *dW, int32_t M, int32_t nprob, int32_t ldw, int32_t nslab,
                                int64_t slab_stride, const float *X2, int32_t nx2, int32_t ldx2, float *dW2, int32_t ldw2, int32_t x2_every,
                                const float *G2, int32_t ng2, int32_t ldg2, float *dW3, int32_t ldw3, void *stream) {
  plan_switches_refresh();
  std::vector<GemmProblem> probs;
  auto wgrad_problem = [&](int nout, int width, float *dst, int ldd, const float *dOut, int ldo, const float *Xp, int ldx) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = nout; p.N = width; p.C = dst; p.ldc = ldd; p.ksplit = nslab; p.split_stride = slab_stride; p.emit_seg = -1;
    GemmSeg &sg = p.seg[p.nseg++];
    sg.A = dOut; sg.lda = ldo; sg.a_kc = 0; sg.B = Xp; sg.ldb = ldx; sg.b_kc = 0; sg.K = M;
    return p;
  };
  for (int i = 0; i < nprob; ++i)
    probs.push_back(wgrad_problem(WG_N, WG_N, dW + (long long)i * WG_N * ldw, ldw, G + (long long)i * M * WG_N, WG_N, X + (long long)i * M * WG_N, WG_N));
  WgArgs wa;
  FDQL_REQUIRE(wgrad_stat_from_problems(probs.data(), nprob, nslab, slab_stride, wa), "the output-stationary kernel does not take this form");
  for (int i = 0; i < nprob; ++i) {
    if (X2 && x2_every > 0 && i % x2_every == 0)
      FDQL_REQUIRE(wgrad_stat_add_rider(wa, i, wgrad_problem(WG_N, nx2, dW2 + (long long)i * WG_N * ldw2, ldw2, G + (long long)i * M * WG_N, WG_N,
                                                            X2 + (long long)i * M * ldx2, ldx2)), "narrow-input rider refused");
    if (G2)
      FDQL_REQUIRE(wgrad_stat_add_rider(wa, i, wgrad_problem(ng2, WG_N, dW3 + (long long)i * ng2 * ldw3, ldw3, G2 + (long long)i * M * ldg2, ldg2,
                                                            X + (long long)i * M * WG_N, WG_N)), "narrow-output rider refused");
  }
  FDQL_REQUIRE(wgrad_stat_balance(wa), "no workgroups to deal");
  hipError_t e = wgrad_stat_launch(wa, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("wgrad_stat launch: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

/* Test hook for the weight-stationary row-block kernel (wstat.hip): `ninst` instances of one layer, instance i using rows
 * [i*M, (i+1)*M) of every activation / output array and its own weights W0[i] [256 x 256] (ks: [k][n], else [n][k] with
 * row stride ldw0), W1[i] / W2[i] ([256 x k1] rows of stride k1, or K-strided [k1 x 256]).  fz_h: fused head dgrad
 * (GemmProblem::fz_*): A0's rows are OUTPUT, formed from fz_h, A1 (= dY, k1 = 2) and fz_w[i] [2 x fz_ldw].  Returns
 * FDQL_EINVAL when the kernel does not take the form (the caller's fallback is the tile kernels). */
int fdql_test_rowgemm(const float *A0, const float *A1, int32_t k1, const float *A2, int32_t k2, const float *W0, int32_t ldw0,
                      const float *W1, const float *W2, const float *bias, float *C, float *C2, const float *ref, float *colsum,
                      const float *hf_w, int32_t hf_ldw, int32_t hf_q, float *hf_out, float *hf_out2, int32_t M, int32_t ninst,
                      int32_t ks, int32_t grad, int32_t dual, int32_t planes, const float *fz_h, const float *fz_w, int32_t fz_ldw,
                      float *fz_colsum, void *stream) {
  plan_switches_refresh();
  const bool presum = planes < 0;   // (negative plane count: the kernel sums a tile's head planes itself, WsArgs::hf_presum)
  if (planes < 0) planes = -planes;
  std::vector<GemmProblem> probs;
  for (int i = 0; i < ninst; ++i) {
    GemmProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.N = WS_N; p.ksplit = 1; p.emit_seg = -1;
    const long long r0 = (long long)i * M;
    p.C = C + r0 * WS_N; p.ldc = WS_N;
    auto seg = [&](const float *Aptr, int lda, const float *W, int ldw, int K) {
      GemmSeg &sg = p.seg[p.nseg++];
      sg.A = Aptr; sg.lda = lda; sg.a_kc = 1; sg.B = W; sg.ldb = ldw; sg.b_kc = ks ? 0 : 1; sg.K = K;
    };
    seg(A0 + r0 * WS_KMAIN, WS_KMAIN, W0 + (long long)i * WS_KMAIN * ldw0 * (ks ? 1 : 1), ks ? WS_N : ldw0, WS_KMAIN);
    if (A1) seg(A1 + r0 * k1, k1, W1 + (long long)i * WS_N * k1, ks ? WS_N : k1, k1);
    if (A2) seg(A2 + r0 * k2, k2, W2 + (long long)i * WS_N * k2, ks ? WS_N : k2, k2);
    if (grad) {
      p.epi = EPI_LRELU_GRAD; p.ref = ref + r0 * WS_N; p.ldref = WS_N;
      p.colsum = colsum + (long long)i * (M / 64) * WS_N;
    } else {
      p.epi = EPI_LRELU; p.bias = bias + (long long)i * WS_N;
    }
    if (dual) { p.emit_seg = p.nseg - 2; p.C2 = C2 + r0 * WS_N; p.ldc2 = WS_N; }
    if (fz_h) {   // the main segment's A block is formed from (fz_h, A1 = dY, fz_w) and lands in A0's rows
      p.fz_h = fz_h + r0 * WS_N; p.fz_w = fz_w + (long long)i * 2 * fz_ldw; p.fz_ldw = fz_ldw;
      p.fz_out = const_cast<float *>(A0) + r0 * WS_KMAIN; p.fz_colsum = fz_colsum + (long long)i * (M / 64) * WS_N;
    }
    if (hf_w) {
      p.hf_w = hf_w + (long long)i * hf_q * hf_ldw; p.hf_ldw = hf_ldw; p.hf_q = hf_q;
      p.hf_out = hf_out + (long long)i * planes * M * hf_q;
      if (dual) p.hf_out2 = hf_out2 + (long long)i * planes * M * hf_q;
    }
    probs.push_back(p);
  }
  RowsLaunch rl;
  rl.ws = wstat_from_problems(probs.data(), (int)probs.size(), rl.wa);
  FDQL_REQUIRE(rl.ws, "the weight-stationary kernel does not take this form");
  if (rl.ws && rl.wa.grad) {
    // the weight-stationary dgrad form writes wstat_colsum_rows() partial rows per instance, not one per 64 rows: clear the
    // caller's [ninst, M / 64, 256] buffers so that their sum over the row axis is the total either way
    const size_t bytes = (size_t)ninst * (M / 64) * WS_N * sizeof(float);
    FDQL_HIP(hipMemsetAsync(colsum, 0, bytes, (hipStream_t)stream));
    if (fz_colsum) FDQL_HIP(hipMemsetAsync(fz_colsum, 0, bytes, (hipStream_t)stream));
  }
  if (rl.ws && rl.wa.hf_q && presum) {
    // the planes summed inside the kernel (WsArgs::hf_presum): plane 0 of each instance holds the total, the caller's other
    // planes are cleared so that its sum over the plane axis is the total either way
    const size_t bytes = (size_t)ninst * planes * M * hf_q * sizeof(float);
    FDQL_HIP(hipMemsetAsync(hf_out, 0, bytes, (hipStream_t)stream));
    if (dual) FDQL_HIP(hipMemsetAsync(hf_out2, 0, bytes, (hipStream_t)stream));
    rl.wa.hf_presum = 1;
  }
  hipError_t e = rl.launch((hipStream_t)stream);
  if (e != hipSuccess) { set_error("row-block launch: %s", hipGetErrorString(e)); return FDQL_EHIP; }
  return 0;
}

}  // extern "C"
