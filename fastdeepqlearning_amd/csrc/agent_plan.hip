// Plan side of an agent (agent_internal.h): arena layout under the reference's state_dict names, the workspace carve, and
// build_plan() - the fixed list of launch stages of one SAC/TQC gradient step (grouped GEMM problem tables, row-block / stationary
// launches, fused loss / policy kernels) for one set of batch pointers - with upload_tables(), which decides the kernels of
// every stage and uploads their tables once; every later update replays the launches.
#include "plan_builder.h"

namespace fdql {

int layout(fdql_agent *a) {
  const fdql_agent_config_t &c = a->cfg;
  int64_t top = 0;
  a->conv.clear();
  a->conv_feat = 0;
  if (c.img_c > 0) {
    int ci = c.img_c, h = c.img_h, w = c.img_w;
    for (int i = 0; i < c.n_conv; ++i) {
      fdql_agent::ConvLayer L;
      L.g.C = ci; L.g.H = h; L.g.W = w; L.g.k = c.conv_k[i]; L.g.s = c.conv_s[i];
      L.g.OH = (h - L.g.k) / L.g.s + 1; L.g.OW = (w - L.g.k) / L.g.s + 1;
      L.cout = c.conv_out[i];
      const int K = ci * L.g.k * L.g.k;
      const std::string pre = "encoder.visible_layer_encoders.obs_2d.conv." + std::to_string(i);
      a->tensors.push_back({pre + ".weight", 0, top, L.cout, K});
      L.w_off = top; top += pad4((int64_t)L.cout * K);
      a->tensors.push_back({pre + ".bias", 0, top, L.cout, 0});
      L.b_off = top; top += pad4(L.cout);
      const bool u8_in = i == 0 && c.obs_2d_u8;
      if (i > 0 || u8_in) {   // (a float32 NCHW first layer has no implicit-GEMM kernel)
        L.fast_fwd = conv_fwd_takes(L.g, L.cout, u8_in);
        L.fast_wgrad = conv_wgrad_takes(L.g, L.cout, u8_in) && L.b_off == L.w_off + (int64_t)L.cout * K;
        L.fast_dgrad = i > 0 && conv_dgrad_takes(L.g, L.cout);
      }
      a->conv.push_back(L);
      ci = L.cout; h = L.g.OH; w = L.g.OW;
    }
    a->conv_feat = ci * h * w;
  }
  add_mlp(a, a->enc_obs, "encoder.visible_layer_encoders.obs_1d", c.obs_dim + 2 * c.goal_dim + a->conv_feat, c.enc_hidden,
          c.n_enc_hidden, c.enc_features, top);
  if (c.joiner_gru) {   // nn.GRU(hidden_features, latent, 1) + learnable start state (encoder.py:41-42)
    const int L3 = 3 * c.latent;
    a->joiner = MlpDesc();
    a->joiner.din = c.enc_features; a->joiner.dout = c.latent;
    a->tensors.push_back({"encoder.hidden_state", 0, top, c.latent, 0});
    a->gru_h0 = top; top += pad4(c.latent);
    a->tensors.push_back({"encoder.joiner.weight_ih_l0", 0, top, L3, c.enc_features});
    a->gru_wih = top; top += pad4((int64_t)L3 * c.enc_features);
    a->tensors.push_back({"encoder.joiner.weight_hh_l0", 0, top, L3, c.latent});
    a->gru_whh = top; top += pad4((int64_t)L3 * c.latent);
    a->tensors.push_back({"encoder.joiner.bias_ih_l0", 0, top, L3, 0});
    a->gru_bih = top; top += pad4(L3);
    a->tensors.push_back({"encoder.joiner.bias_hh_l0", 0, top, L3, 0});
    a->gru_bhh = top; top += pad4(L3);
  } else {
    add_mlp(a, a->joiner, "encoder.joiner", c.enc_features, c.joint_hidden, c.n_joint_hidden, c.latent, top);
  }
  a->tgt_begin = top;
  const int pi_out = c.discrete ? c.act_dim : 2 * c.act_dim;
  add_mlp(a, a->actor, "actor_critic.actor", c.latent, c.pi_hidden, c.n_pi_hidden, pi_out, top);
  a->crit_begin = top;
  a->critic.resize(c.n_critics);
  for (int k = 0; k < c.n_critics; ++k)
    add_mlp(a, a->critic[k], "actor_critic.critic.nets." + std::to_string(k), c.latent + c.act_dim, c.critic_hidden,
            c.n_critic_hidden, c.n_quantiles, top);
  a->crit_end = top;
  a->tgt_end = top;
  a->log_alpha_off = top;
  a->tensors.push_back({"actor_critic.log_alpha", 0, top, 0, 0});
  top += 4;
  a->n_train = top;
  // mirrored arenas
  const size_t n0 = a->tensors.size();
  for (size_t i = 0; i < n0; ++i) {
    const TensorInfo &t = a->tensors[i];
    if (t.off >= a->tgt_begin && t.off < a->tgt_end) {
      TensorInfo u = t;
      u.arena = 1;
      u.off = t.off - a->tgt_begin;
      size_t p;
      if ((p = u.name.find(".actor.")) != std::string::npos) u.name.replace(p, 7, ".actor_target.");
      else if ((p = u.name.find(".critic.")) != std::string::npos) u.name.replace(p, 8, ".critic_target.");
      a->tensors.push_back(u);
    }
  }
  for (size_t i = 0; i < n0; ++i) {
    const TensorInfo &t = a->tensors[i];
    if (t.off >= a->crit_begin && t.off < a->crit_end) {
      TensorInfo u = t;
      u.arena = 2;
      u.off = t.off - a->crit_begin;
      const size_t p = u.name.find(".critic.");
      u.name.replace(p, 8, ".critic_frozen.");
      a->tensors.push_back(u);
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------ workspace
void carve(fdql_agent *a) {
  a->carve_top = 0;
  a->named.clear();
  const fdql_agent_config_t &c = a->cfg;
  const int64_t N = a->N, M = a->M, Nq = a->Nq;
  a->alloc("dev_state", 64);
  a->alloc("scalars", 16);
  a->alloc("w", M);
  a->alloc("is_contiguous", M);
  auto mlp_bufs = [&](const std::string &p, const MlpDesc &d, int64_t rows, bool bwd, bool out) {
    for (size_t i = 0; i < d.hid.size(); ++i) {
      a->alloc(p + ".h" + std::to_string(i), rows * d.hid[i]);
      if (bwd) {
        a->alloc(p + ".dpre" + std::to_string(i), M * d.hid[i]);
        a->alloc(p + ".cs" + std::to_string(i), ((M + 15) / 16) * d.hid[i]);   // per 64 rows (tile kernels) ... per 16 rows (k_rowdgrad_chain<1>)
      }
    }
    if (out) a->alloc(p + ".out", rows * d.dout);
  };
  for (size_t i = 0; i < a->conv.size(); ++i) {   // im2col matrix, NHWC output, and their gradients over the M images
    const fdql_agent::ConvLayer &L = a->conv[i];
    const int64_t pos = (int64_t)L.g.OH * L.g.OW, K = (int64_t)L.g.C * L.g.k * L.g.k;
    const std::string p = "conv" + std::to_string(i);
    if (!L.fast_fwd || !L.fast_wgrad) a->alloc(p + ".col", N * pos * K);   // (the implicit-GEMM kernels have no column matrix)
    a->alloc(p + ".out", N * pos * L.cout);
    a->alloc(p + ".dpre", M * pos * L.cout);
    if (i > 0 && !L.fast_dgrad) a->alloc(p + ".dcol", M * pos * K);
    if (L.fast_wgrad) {   // one (dW, db) partial per slab of the output-stationary launch, then one reduction into slab 0
      a->alloc(p + ".wpart", (int64_t)conv_wgrad_slabs(L.g, L.cout, i == 0, M) * ((int64_t)L.cout * K + L.cout));
    } else {
      // weight gradient: K-split of its own over the M*pos rows (far more rows than the slab count serves), then a
      // reduction of the partials into slab 0; bias gradient: two-level column sum
      a->alloc(p + ".wpart", (int64_t)conv_wsplit(M * pos) * L.cout * K);
      a->alloc(p + ".bpart", (int64_t)(colsum_tall_blocks(M * pos) + colsum_tall_blocks(colsum_tall_blocks(M * pos))) * L.cout);
    }
  }
  mlp_bufs("enc_obs", a->enc_obs, N, true, true);
  if (c.joiner_gru) {
    const int64_t L3 = 3 * c.latent, Bw = a->B;
    a->alloc("gru.gi", N * L3);          // W_ih e + b_ih for every row
    a->alloc("gru.gh", N * L3);          // W_hh h_{t-1} + b_hh, step by step
    a->alloc("gru.hprev", N * c.latent); // h_{t-1} per row (start state for t = 0)
    a->alloc("gru.h0", Bw * c.latent);
    a->alloc("gru.dgi", M * L3);
    a->alloc("gru.dgh", M * L3);
    a->alloc("gru.dhz0", Bw * c.latent); // direct part of d h_{t-1} (dh * z), double-buffered over t
    a->alloc("gru.dhz1", Bw * c.latent);
    a->alloc("gru.dhw", GRU_KSPLIT_BWD * Bw * c.latent);  // part of d h_{t-1} through W_hh, K-split partials
    a->alloc("gru.ghp", GRU_KSPLIT_FWD * Bw * L3);        // K-split partials of W_hh h_{t-1} for the current step
    a->alloc("gru.wpack_f", (int64_t)c.latent * L3);      // persistent scans (gruscan.hip): W_hh packed in the forward scan's stream order
    a->alloc("gru.wpack_b", (int64_t)c.latent * L3);      // ... and W_hh^T in the backward scan's
    a->alloc("gru.dh_init", Bw * c.latent);               // ... and d h_{-1} per row (learned start state)
  } else {
    mlp_bufs("joiner", a->joiner, N, true, false);
  }
  a->alloc("state", N * c.latent);
  mlp_bufs("actor_t", a->actor, M, false, true);
  mlp_bufs("actor", a->actor, M, true, true);
  a->alloc("next_action", M * c.act_dim);
  a->alloc("next_log_pi", M);
  a->alloc("pi", M * c.act_dim);
  a->alloc("log_pi", M);
  a->alloc("noise_actor", M * c.act_dim);
  a->alloc("pi_diff", M * c.act_dim);
  if (c.discrete) a->alloc("action_onehot", N * c.act_dim);
  for (int k = 0; k < c.n_critics; ++k) {
    const std::string s = std::to_string(k);
    mlp_bufs("crit_t" + s, a->critic[k], M, false, false);
    mlp_bufs("crit" + s, a->critic[k], M, true, false);
    mlp_bufs("crit_f" + s, a->critic[k], M, true, false);
    for (size_t i = 0; i < a->critic[k].hid.size(); ++i)   // gate masks of the online and the frozen pass (wstat.hip; 32 bytes per row)
      for (const char *pre : {"crit", "crit_f"}) a->alloc(pre + s + ".gm" + std::to_string(i), (int64_t)((M + 31) / 32) * 256);
  }
  {   // head fusion: per critic instance (3C of them) the partial head sums of every hidden layer, then their total
    int planes = 0;
    for (int h : a->critic[0].hid) planes += ((h + 63) / 64) * 2;
    a->hf_planes = planes;
    const int q = c.n_quantiles;
    const bool can_fuse = planes > 0 && (q == 1 || q == 2 || q == 4 || q == 8);   // same rule as the plan below
    a->alloc("hf.parts", can_fuse ? (int64_t)3 * c.n_critics * planes * M * q : 1);
    a->alloc("hf.sum", can_fuse ? (int64_t)3 * c.n_critics * M * q : 1);
  }
  a->alloc("next_z", M * Nq);
  a->alloc("q_pred", M * Nq);
  a->alloc("q_frozen", M * Nq);
  a->alloc("td_target", M * (a->Nt > 0 ? a->Nt : 1));
  a->alloc("dz", M * Nq);
  a->alloc("dzf", M * Nq);
  a->alloc("q_loss", M);
  a->alloc("pi_loss", M);
  a->alloc("alpha_loss", M);
  a->alloc("dpi", M * c.act_dim);
  a->alloc("dlogits", M * a->actor.dout);
  a->alloc("dstate", M * c.latent);
  if (a->dstate_split) a->alloc("dstate.parts", (int64_t)(c.n_critics + 1) * M * c.latent);
  a->alloc("denc", M * c.enc_features);
  a->alloc("cs.dstate", ((M + 15) / 16) * c.latent);   // per 64 rows (one-problem d state), per 32 rows (sum of shares), per block of the chain launch
  a->alloc("cs.denc", ((M + 15) / 16) * c.enc_features);
  a->alloc("dpi_part", (int64_t)c.n_critics * M * c.act_dim);
  a->alloc("loss_partials", (int64_t)loss_blocks((int)M, 256) * LOSS_NPART + (int64_t)M * LOSS_NPART + LOSS_NPART);
  a->alloc("slabs", (int64_t)a->nsplit * a->n_train);
  a->alloc("loss_fin_args", 32);   // LossFinishArgs (update_kernels.h): k_loss's fused finish
  {   // fdql_agent_summaries: 4 scalars + one norm per trainable tensor, then the int64 (offset, count) table (8-byte aligned)
    int64_t nt = 0;
    for (const TensorInfo &t : a->tensors) nt += t.arena == 0;
    a->alloc("summaries", ((4 + nt + 1) & ~(int64_t)1) + 4 * nt + 4);
  }
}

int upload_tables(fdql_agent *a) {
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  size_t total = 0;
  for (Stage &s : a->stages) {
    if (s.kind == ST_GEMM) {
      for (auto &sub : s.sub) sub.probs.clear();
      s.rows.clear();
      std::vector<char> taken(s.gemm.size(), 0);
      if (s.try_rows) {   // like problems (same segment list shape and epilogue) -> one row-block launch per group
        {   // ... or the whole stage as ONE weight-stationary launch (critic layer 0: two-output and plain instances mixed)
          RowsLaunch rl;
          if (s.gemm.size() > 1 && rows_launch_of(a, s.gemm, rl) && rl.ws) {
            s.rows.push_back(rl);
            std::fill(taken.begin(), taken.end(), 1);
          }
        }
        for (size_t i = 0; i < s.gemm.size(); ++i) {
          if (taken[i]) continue;
          std::vector<GemmProblem> grp;
          std::vector<size_t> idx;
          for (size_t j = i; j < s.gemm.size(); ++j) {
            const GemmProblem &p = s.gemm[j], &q = s.gemm[i];
            bool like = !taken[j] && p.nseg == q.nseg && p.emit_seg == q.emit_seg && p.epi == q.epi && (p.C2 != nullptr) == (q.C2 != nullptr);
            for (int sg = 0; like && sg < p.nseg; ++sg)
              like = p.seg[sg].K == q.seg[sg].K && p.seg[sg].lda == q.seg[sg].lda && p.seg[sg].ldb == q.seg[sg].ldb &&
                     p.seg[sg].a_kc == q.seg[sg].a_kc && p.seg[sg].b_kc == q.seg[sg].b_kc;
            if (like) {
              grp.push_back(p);
              idx.push_back(j);
            }
          }
          RowsLaunch rl;
          if (rows_launch_of(a, grp, rl)) {
            s.rows.push_back(rl);
            for (size_t j : idx) taken[j] = 1;
          } else {
            for (size_t j : idx) taken[j] = 2;   // looked at, stays on the tile kernels
          }
        }
      }
      // a whole stage of narrow-output dgrads (d pi: one problem per frozen critic) as one streaming launch
      if (!s.gemm.empty() && std::find(taken.begin(), taken.end(), (char)1) == taken.end() && s.gemm[0].M >= a->rowdot_min_rows) {
        RowsLaunch rl;
        if (rowdot_from_problems(s.gemm.data(), (int)s.gemm.size(), rl.rdot)) {
          rl.dot = true;
          s.rows.push_back(rl);
          std::fill(taken.begin(), taken.end(), 1);
        }
      }
      // single-network dgrads (256-wide K-strided segments, gate / column sums) with enough 64-row blocks to fill most of the chip
      for (size_t i = 0; i < s.gemm.size(); ++i) {
        RowsLaunch rl;
        const bool small_chain = s.rd_chain_bm > 0 && s.rd_chain_bm < RD_BM;   // a member of a planned chain launch on 32- / 16-row blocks
        if (taken[i] != 1 && (small_chain || (s.gemm[i].M / RD_BM >= a->rowdgrad_min_blocks &&
                                              s.gemm[i].M / RD_BM <= (s.rd_max_blocks > 0 ? s.rd_max_blocks : a->rowdgrad_max_blocks))) &&
            rowdgrad_from_problem(s.gemm[i], rl.rda, small_chain ? s.rd_chain_bm : RD_BM)) {
          if (s.fold_sum && !rowdgrad_fold_sum(rl.rda, s.fold_parts, s.fold_n, s.fold_stride, s.fold_out, s.fold_cs)) {
            set_error("stage %s: the row-block dgrad kernel does not take the folded sum it was planned with", s.name.c_str());
            return FDQL_EINVAL;
          }
          rl.rd = true;
          s.rows.push_back(rl);
          taken[i] = 1;
        }
      }
      if (s.fold_sum && (s.rows.size() != 1 || !s.rows[0].rd)) {
        set_error("stage %s was planned with a folded sum but does not run on the row-block dgrad kernel", s.name.c_str());
        return FDQL_EINVAL;
      }
      // Small batches (temporal_len 2, a handful of env rows): a stage whose problems are too few tiles to fill the chip is as
      // long as one workgroup's serial K loop on the tile kernels; the small-batch kernel (smallgemm.hip) splits K over the 16
      // waves of a workgroup instead.  Per stage: every problem left for the tile kernels must have the form, and together they
      // are at most small_max_tiles 64x64 tiles (FDQL_SMALL_GEMM=0: never; FDQL_SMALL_GEMM_MAX_TILES).
      // A non-default tile shape / main-loop build (tests, experiments) keeps its kernels; dense shape GEMM_SMALL (test hook)
      // forces the small-batch kernel on every problem that has the form, whatever the size.
      const int dshape = gemm_dense_shape();
      const bool small_forced = dshape == GEMM_SMALL;
      bool small = a->small_max_tiles > 0 && dshape == GEMM_64x64;
      long long tiles64 = 0;
      for (size_t i = 0; i < s.gemm.size() && small; ++i) {
        if (taken[i] == 1) continue;
        small = gemm_small_takes(s.gemm[i]);
        tiles64 += (long long)((s.gemm[i].M + 63) / 64) * ((s.gemm[i].N + 63) / 64);
      }
      // (short-K stages - the rank-Q outer product of a last hidden layer's gradient under a Q-wide head - are one load round per
      // workgroup whatever their tile count: up to 4x the tiles)
      int kmax = 0;
      for (size_t i = 0; i < s.gemm.size(); ++i)
        if (taken[i] != 1) { int k = 0; for (int sg = 0; sg < s.gemm[i].nseg; ++sg) k += s.gemm[i].seg[sg].K; kmax = std::max(kmax, k); }
      small = small && tiles64 > 0 && (tiles64 <= a->small_max_tiles || (kmax <= 32 && tiles64 <= 4LL * a->small_max_tiles));
      for (size_t i = 0; i < s.gemm.size(); ++i) {
        if (taken[i] == 1) continue;
        if (s.gemm[i].fz_h) { set_error("stage %s: a fused head-dgrad problem was not taken by the row-block kernel", s.name.c_str()); return FDQL_ESTATE; }
        const bool sm = small || (small_forced && gemm_small_takes(s.gemm[i]));
        s.sub[sm ? (int)GEMM_SMALL : gemm_pick_shape(s.gemm[i], small_forced ? (int)GEMM_64x64 : dshape)].probs.push_back(s.gemm[i]);
      }
      for (auto &sub : s.sub) total += pad(sub.probs.size() * sizeof(GemmProblem));
    }
    if (s.kind == ST_SKINNY_WGRAD) total += pad(s.swg.size() * sizeof(SkinnyWgradProblem));
    if (s.kind == ST_HEAD_DGRAD) total += pad(s.hdg.size() * sizeof(HeadDgradProblem));
    if (s.kind == ST_CHAIN) total += pad(s.cprobs.size() * sizeof(ChainProblem)) + pad(s.cops.size() * sizeof(ChainOp));
  }
  if (a->tables_dev) { FDQL_HIP(hipFree(a->tables_dev)); a->tables_dev = nullptr; }
  FDQL_HIP(hipMalloc(&a->tables_dev, total ? total : 256));
  std::vector<char> host(total);
  size_t off = 0;
  for (Stage &s : a->stages) {
    if (s.kind == ST_GEMM) {
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.gemm) { s.flops += gemm_flops(p); s.bytes += gemm_bytes(p); }
      for (int sh = 0; sh < GEMM_NSHAPES; ++sh) {
        GemmSub &sub = s.sub[sh];
        if (sub.probs.empty()) { sub.blocks = 0; sub.dev = nullptr; continue; }
        sub.blocks = gemm_finalize(sub.probs.data(), (int)sub.probs.size(), sh);
        const size_t bytes = sub.probs.size() * sizeof(GemmProblem);
        memcpy(host.data() + off, sub.probs.data(), bytes);
        sub.dev = (char *)a->tables_dev + off;
        off += pad(bytes);
      }
    } else if (s.kind == ST_SKINNY_WGRAD) {
      s.blocks = s.stream ? stream_wgrad_finalize(s.swg.data(), (int)s.swg.size()) : skinny_wgrad_finalize(s.swg.data(), (int)s.swg.size());
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.swg) {
        s.flops += 2.0 * p.M * (double)p.K * p.Nout;
        s.bytes += 4.0 * p.M * ((double)p.K + (p.dY ? p.Nout : 0));
      }
      const size_t bytes = s.swg.size() * sizeof(SkinnyWgradProblem);
      memcpy(host.data() + off, s.swg.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    } else if (s.kind == ST_HEAD_DGRAD) {
      s.blocks = head_dgrad_finalize(s.hdg.data(), (int)s.hdg.size());
      s.flops = 0; s.bytes = 0;
      for (auto &p : s.hdg) {
        s.flops += 2.0 * p.M * (double)p.N * p.Q;
        s.bytes += 8.0 * p.M * (double)p.N;   // read h, write dpre
      }
      const size_t bytes = s.hdg.size() * sizeof(HeadDgradProblem);
      memcpy(host.data() + off, s.hdg.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    } else if (s.kind == ST_CHAIN) {
      s.blocks = chain_finalize(s.cprobs.data(), (int)s.cprobs.size(), s.chain_bm);
      size_t bytes = s.cprobs.size() * sizeof(ChainProblem);
      memcpy(host.data() + off, s.cprobs.data(), bytes);
      s.dev = (char *)a->tables_dev + off;
      off += pad(bytes);
      bytes = s.cops.size() * sizeof(ChainOp);
      memcpy(host.data() + off, s.cops.data(), bytes);
      s.cops_dev = (char *)a->tables_dev + off;
      off += pad(bytes);
    }
  }
  if (total) FDQL_HIP(hipMemcpy(a->tables_dev, host.data(), total, hipMemcpyHostToDevice));
  // three dependent single-network dgrads in a row on the row-block dgrad kernel, the first with the folded sum of the d state
  // shares (a config-2-shaped plan: joiner.dpre0, d enc, enc_obs.dpre0): one launch with the 64-row activations resident in LDS
  for (Stage &s : a->stages)   // (decided anew with every table upload)
    if (s.chained) { s.off = false; s.chained = false; }
  for (size_t i = 0; i + 2 < a->stages.size(); ++i) {
    Stage &s1 = a->stages[i], &s2 = a->stages[i + 1], &s3 = a->stages[i + 2];
    auto lone_rd = [](const Stage &s) {
      if (s.kind != ST_GEMM || s.gemm.size() != 1 || s.rows.size() != 1 || !s.rows[0].rd) return false;
      for (const GemmSub &sub : s.sub) if (!sub.probs.empty()) return false;
      return true;
    };
    if (!s1.fold_sum || !lone_rd(s1) || !lone_rd(s2) || !lone_rd(s3) || s1.phase != s2.phase || s1.phase != s3.phase) continue;
    if (s1.rd_chain_bm != s2.rd_chain_bm || s1.rd_chain_bm != s3.rd_chain_bm) continue;
    RowChainArgs c;
    if (!rowchain_from_launches(s1.rows[0].rda, s2.rows[0].rda, s3.rows[0].rda, c, s1.rd_chain_bm > 0 ? s1.rd_chain_bm : RD_BM)) continue;
    s1.rows[0].chain3 = true;
    s1.rows[0].rch = c;
    s2.off = s3.off = true;
    s2.chained = s3.chained = true;
    // the launch does their work (fdql_agent_stats counts executed flops); s1's own figures were recomputed from its
    // problems at the top of this upload, so a second upload does not add them twice
    s1.flops += s2.flops + s3.flops;
    s1.bytes += s2.bytes + s3.bytes;
  }
  // a stage planned on 32- / 16-row blocks exists only inside a chain launch (the lone kernel is the 64-row one)
  for (const Stage &s : a->stages)
    if (s.rd_chain_bm > 0 && s.rd_chain_bm < RD_BM && !s.chained && !(s.rows.size() == 1 && s.rows[0].chain3)) {
      set_error("stage %s was planned as a member of a %d-row dgrad chain that did not form", s.name.c_str(), s.rd_chain_bm);
      return FDQL_EINVAL;
    }
  // head-fusion planes: with every hidden layer of the critics on weight-stationary launches, those launches sum a tile's column
  // planes themselves, the plane-sum stage is switched off and the finish adds one plane per layer (Stage::hf_role)
  {
    Stage *fin = nullptr, *sum = nullptr;
    int nfwd = 0;
    bool all = true;
    for (Stage &s : a->stages) {
      if (s.hf_role == 2) sum = &s;
      if (s.hf_role == 3) fin = &s;
      if (s.hf_role != 1) continue;
      ++nfwd;
      size_t n = 0;
      bool ok = !s.rows.empty();
      for (const RowsLaunch &rl : s.rows) { ok = ok && rl.ws && rl.wa.hf_q > 0; n += (size_t)rl.wa.ninst; }
      all = all && ok && n == s.gemm.size();
    }
    const bool presum = fin && fin->hfin_can_presum && nfwd > 0 && nfwd == fin->hfin_presum.planes && all;
    for (Stage &s : a->stages)
      if (s.hf_role == 1)
        for (RowsLaunch &rl : s.rows)
          if (rl.ws) rl.wa.hf_presum = presum ? 1 : 0;
    if (fin) *fin->hfin = presum ? fin->hfin_presum : fin->hfin_plain;
    if (sum) sum->off = presum;
  }
  // gate masks: written by the critics' forward launches and read by their backward launches only when every one of those
  // forward layers runs weight-stationary (the mask layout is that kernel's register layout); FDQL_NO_GATE_MASKS: never
  {
    int nfwd = 0;
    bool all = plan_switches().gate_masks;
    for (Stage &s : a->stages) {
      if (s.gm_role != 1) continue;
      ++nfwd;
      size_t n = 0;
      bool ok = !s.rows.empty();
      for (const RowsLaunch &rl : s.rows) { ok = ok && rl.ws; n += (size_t)rl.wa.ninst; }
      all = all && ok && n == s.gemm.size();
    }
    const bool masks = nfwd > 0 && all;
    for (Stage &s : a->stages) {   // a stage planned on the masks and its GEMM stand-in: exactly one of them runs
      if (s.needs_masks) s.off = !masks;
      if (s.masks_fallback) s.off = masks;
    }
    for (Stage &s : a->stages) {
      for (RowsLaunch &rl : s.rows) {
        if (!rl.ws) continue;
        if (rl.wa.grad == 0) {   // forward launches: nobody reads masks written outside the scheme
          if (!masks || s.gm_role != 1)
            for (int i = 0; i < rl.wa.ninst; ++i) rl.wa.inst[i].gm_out = rl.wa.inst[i].gm_out2 = nullptr;
        } else if (s.gm_role == 2 && rl.wa.grad == 1) {   // gated dgrad forms: every instance must carry the masks it would read
          bool have = masks;
          for (int i = 0; i < rl.wa.ninst; ++i) have = have && rl.wa.inst[i].gm_ref && (!rl.wa.fz || rl.wa.inst[i].gm_fz);
          rl.wa.use_masks = have ? 1 : 0;
        }
      }
    }
  }
  return 0;
}

int build_plan(fdql_agent *a) {
  const fdql_agent_config_t &c = a->cfg;
  a->stages.clear();
  Builder b(a);
  const int N = a->N, M = a->M, B = a->B, L = c.latent, A = c.act_dim, C = c.n_critics, Q = c.n_quantiles, Nq = a->Nq;
  const fdql_batch_t &x = a->batch;
  float *params = a->params, *targets = a->targets;

  // ---- instances
  MlpInst eo = make_inst(a, a->enc_obs, "enc_obs", params, 0, N, true);
  if (c.obs_dim) eo.in.push_back({x.obs_1d, c.obs_dim, c.obs_dim});
  if (c.goal_dim) {
    eo.in.push_back({x.achieved_goal, c.goal_dim, c.goal_dim});
    eo.in.push_back({x.desired_goal, c.goal_dim, c.goal_dim});
  }
  const int nconv = (int)a->conv.size();
  const int conv_col0 = c.obs_dim + 2 * c.goal_dim;   // first column of the conv features in the obs MLP's input
  if (nconv) {
    const float *feat = a->buf("conv" + std::to_string(nconv - 1) + ".out");
    eo.in.push_back({feat, a->conv_feat, a->conv_feat});
  }
  eo.out = a->buf("enc_obs.out"); eo.ldout = c.enc_features;
  const bool gru = c.joiner_gru != 0;
  MlpInst jo;
  float *state = a->buf("state");
  if (!gru) {
    jo = make_inst(a, a->joiner, "joiner", params, 0, N, true);
    jo.in.push_back({eo.out, c.enc_features, c.enc_features});
    jo.out = state; jo.ldout = L;
  }
  const float *s_cur = state, *s_nxt = state + (int64_t)B * L;

  MlpInst at = make_inst(a, a->actor, "actor_t", targets, a->tgt_begin, M, false);
  at.in.push_back({s_nxt, L, L});
  at.out = a->buf("actor_t.out"); at.ldout = a->actor.dout;
  MlpInst ao = make_inst(a, a->actor, "actor", params, 0, M, true);
  ao.in.push_back({s_cur, L, L});
  ao.out = a->buf("actor.out"); ao.ldout = a->actor.dout;

  std::vector<MlpInst> ct, co, cf;
  for (int k = 0; k < C; ++k) {
    const std::string s = std::to_string(k);
    MlpInst t = make_inst(a, a->critic[k], "crit_t" + s, targets, a->tgt_begin, M, false);
    t.in.push_back({s_nxt, L, L});
    t.in.push_back({a->buf("next_action"), A, A});
    t.out = a->buf("next_z") + k * Q; t.ldout = Nq;
    ct.push_back(t);
    MlpInst o = make_inst(a, a->critic[k], "crit" + s, params, 0, M, true);
    o.in.push_back({s_cur, L, L});
    o.in.push_back({c.discrete ? a->buf("action_onehot") : x.action, A, A});
    o.out = a->buf("q_pred") + k * Q; o.ldout = Nq;
    co.push_back(o);
    MlpInst f = make_inst(a, a->critic[k], "crit_f" + s, params, 0, M, true);
    f.in.push_back({s_cur, L, L});
    f.in.push_back({a->buf("pi"), A, A});
    f.out = a->buf("q_frozen") + k * Q; f.ldout = Nq;
    cf.push_back(f);
  }

  DevState *dst = a->st();
  const float *log_alpha = params + a->log_alpha_off;

  // ---- stage 0: tick + prep
  bool fold_prep = false;
  PrepArgs prep_args;
  memset(&prep_args, 0, sizeof(prep_args));
  {
    const float inv_gb = 1.0f / (float)(B * (c.world_size > 0 ? c.world_size : 1));
    float *w = a->buf("w"), *ic = a->buf("is_contiguous");
    const float *td = x.task_done, *es = x.episode_step;
    const int T = a->T;
    const int burn = c.burn_in_steps;
    const int cumprod = gru ? 1 : 0;   // encoder.py:80
    // continuous policies: prep's workgroups ride in the policy-forward launch (nothing before the loss reads what it writes)
    fold_prep = !c.discrete && plan_switches().small_folds;
    prep_args = PrepArgs{td, es, T, B, burn, cumprod, inv_gb, w, ic, dst, log_alpha};
    if (!fold_prep)
      b.func_stage("prep", [=](hipStream_t s) { return prep_launch(td, es, T, B, burn, cumprod, inv_gb, w, ic, dst, log_alpha, s); });
    if (c.discrete) {   // stored action index -> one-hot critic input (deepQlearning.py:206-210)
      const float *act = x.action;
      float *oh = a->buf("action_onehot");
      b.func_stage("onehot", [=](hipStream_t s) { return onehot_launch(act, N, A, oh, s); });
    }
  }
  // ---- encoder forward (encoder.py:52-67)
  auto fwd_chain = [&](std::vector<MlpInst *> group, const std::string &name) {
    const size_t nh = group[0]->d->hid.size();
    for (size_t i = 0; i < nh; ++i) {
      Stage &gs = b.gemm_stage(name + ".fwd" + std::to_string(i));
      // a 256 -> 256 hidden layer of one or two networks with >= one 64-row tile per CU runs weight-stationary (config 4 / 5:
      // joiner.fwd0 0.083 -> 0.071 ms, actors.fwd0 0.147 -> 0.123 at config 4; measured in round 5, kept in round 6)
      gs.try_rows = true;
      for (MlpInst *m : group) gs.gemm.push_back(b.fwd_layer(*m, (int)i));
    }
    Stage &hs = b.gemm_stage(name + ".head");
    for (MlpInst *m : group) hs.gemm.push_back(b.fwd_head(*m));
  };
  for (int i = 0; i < nconv; ++i) {   // pixel encoder forward: im2col + GEMM (bias, LeakyReLU) per layer, all N images
    const fdql_agent::ConvLayer &Lc = a->conv[i];
    const ConvGeom g = Lc.g;
    const int K = g.C * g.k * g.k;
    const long long rows = (long long)N * g.OH * g.OW;
    FDQL_REQUIRE(rows < (1LL << 31), "conv layer %d: %lld im2col rows exceed the GEMM's 32-bit row index", i, rows);
    float *out = a->buf("conv" + std::to_string(i) + ".out");
    if (Lc.fast_fwd) {   // implicit GEMM (conv.hip): the image groups resident in LDS, no column matrix
      ConvFwdArgs ca;
      ca.in.base = i == 0 ? (const void *)x.obs_2d_u8 : (const void *)a->buf("conv" + std::to_string(i - 1) + ".out");
      ca.in.u8 = i == 0; ca.in.slots = i == 0 ? x.obs_2d_slots : nullptr;
      ca.W = params + Lc.w_off; ca.bias = params + Lc.b_off; ca.out = out; ca.nimg = N; ca.g = g; ca.cout = Lc.cout;
      Stage &cs = b.func_stage("conv.fwd" + std::to_string(i), [=](hipStream_t s) { return conv_fwd_launch(ca, s); });
      cs.mfma = true;
      cs.flops = 2.0 * (double)rows * K * Lc.cout;
      cs.bytes = (i == 0 ? 1.0 : 4.0) * (double)N * g.C * g.H * g.W + 4.0 * (double)rows * Lc.cout;
      continue;
    }
    FDQL_REQUIRE(i > 0 || x.obs_2d, "conv layer 0 runs on the im2col path: it needs the float32 frames (batch.obs_2d)");
    float *col = a->buf("conv" + std::to_string(i) + ".col");
    const float *in = i == 0 ? x.obs_2d : a->buf("conv" + std::to_string(i - 1) + ".out");
    const int nhwc = i > 0;
    const float scale = i == 0 ? 1.0f / 255.0f : 1.0f;
    const long long nimg = N;
    b.func_stage("conv.im2col", [=](hipStream_t s) { return im2col_launch(in, nhwc, scale, nimg, g, col, s); });
    Stage &gs = b.gemm_stage("conv.fwd" + std::to_string(i));
    GemmProblem p = Builder::new_gemm((int)rows, Lc.cout, out, Lc.cout);
    Builder::add_seg(p, col, K, 1, params + Lc.w_off, K, 1, K);
    p.bias = params + Lc.b_off;
    p.epi = EPI_LRELU;
    gs.gemm.push_back(p);
  }
  // Row-block chain (chain.hip): encoder MLP -> joiner MLP -> online actor and target actor in ONE launch, the
  // activations of a 64-row block resident in LDS from the observation to the policy logits.  Falls back to the
  // per-layer launches when a layer does not fit the kernel (see ChainBuilder).
  // FDQL_CHAIN: "0" never, "1" (default) the encoder/actor chain when the batch fills at least half the chip with
  // 64-row blocks (fewer blocks leave most CUs idle for the length of a whole chain: the per-layer launches with their
  // K-splits are faster there), "all" every eligible program incl. the critics' (measured slower than the grouped
  // launches at config 2 so far: DESIGN.md section 5), regardless of size - the parity tests run all three.
  const int chain_mode = plan_switches().chain;   // (common.h: 0 never, 1 default, 2 "enc": the encoder chain whatever the size, 3 "all")
  const bool chain_all = chain_mode == 3;
  const int chain_min_blocks = chain_mode >= 2 ? 1 : 96;
  // Rows per workgroup: 64 when that many blocks fill the chip, else 32 (twice the workgroups - one rank's share of a
  // data-parallel batch - and images of half the size: 64 rows of a 376-column observation next to a hidden image do not fit
  // the LDS, 32 do)
  const bool chain_on = chain_mode != 0;
  std::vector<int> bms;
  {
    // 32-row blocks only while they are one round of workgroups (one per CU): measured at config 4, 128 windows per GPU
    // (200 blocks) 1.172 -> 1.154 ms per step against the six per-layer launches; at 256 windows (400 blocks, 1.6 rounds) the
    // chain is the slower one (1.902 -> 1.987 ms).  While they ARE one round they come first: a block's time is the weights it
    // streams plus its MFMAs, 200 blocks of 32 rows finish in 0.10 ms where 100 blocks of 64 take 0.146 (config 2, 128 windows).
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const bool one_round32 = N >= (long long)chain_min_blocks * 32 && (N + 31) / 32 <= ncu;
    if (one_round32 && !chain_all) bms.push_back(32);
    if (chain_all || N >= (long long)chain_min_blocks * CH_BM) bms.push_back(CH_BM);
    if (chain_all) bms.push_back(32);
  }
  bool enc_chained = false;
  // Round 6: the same four networks on 16- / 32-row blocks, two workgroups per CU, weights eight groups ahead (fwdchain.hip) when
  // they have the reference's default shape (one hidden layer of 256 everywhere) - k_chain's blocks are latency-bound below a
  // dispatch round of them
  if (chain_mode == 1 && !gru && nconv == 0 && a->enc_obs.hid.size() == 1 && a->joiner.hid.size() == 1 && a->actor.hid.size() == 1 &&
      a->enc_obs.hid[0] == F3_W && a->joiner.hid[0] == F3_W && a->actor.hid[0] == F3_W && c.enc_features == F3_W && L == F3_W &&
      (int)eo.in.size() <= F3_MAX_IN) {
    Fwd3Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.N = N; fa.M = M; fa.B = B;
    fa.nin = (int)eo.in.size();
    for (int i = 0; i < fa.nin; ++i) { fa.in[i] = eo.in[i].ptr; fa.in_ld[i] = eo.in[i].ld; fa.in_w[i] = eo.in[i].width; fa.K0 += eo.in[i].width; }
    auto mlp_of = [](const MlpInst &m) { return Fwd3Mlp{m.W(0), m.Bv(0), m.HW(), m.HB()}; };
    fa.enc = mlp_of(eo); fa.joi = mlp_of(jo); fa.act = mlp_of(ao); fa.act_t = mlp_of(at);
    fa.P = a->actor.dout;
    fa.enc_h = eo.h[0]; fa.enc_out = eo.out; fa.joi_h = jo.h[0]; fa.state = state; fa.act_h = ao.h[0]; fa.act_out = ao.out; fa.act_t_out = at.out;
    // Block size by measurement (profiles/r06_fwd3_blocks.txt): 16 rows while that is at most one workgroup per CU (17 observation
    // columns, 512 / 1 600 / 3 200 rows: 0.052 / 0.064 / 0.066 ms against 0.085-0.101 for the six small-batch launches / k_chain<1>),
    // 32 rows up to one round of two workgroups per CU (6 400 / 12 800 rows: 0.087 / 0.146 ms against k_chain's 0.108 / 0.159;
    // config 4's 376 columns at 6 400 / 12 800 rows: 0.122 / 0.244 against 0.183 / 0.290); beyond one round the per-layer launches /
    // k_chain keep the plan (config 4 at 51 200 rows: 0.84 against 0.75 ms).  Wide observations need the 16-byte request form
    // (fwdchain.hip VOBS: row pitch a multiple of four floats, whole ring rounds)
    int ncu3 = 256, dev3 = 0;
    if (hipGetDevice(&dev3) == hipSuccess) (void)hipDeviceGetAttribute(&ncu3, hipDeviceAttributeMultiprocessorCount, dev3);
    auto obs_ok = [&](int bmx) { return fa.K0 <= 64 || (fa.K0 % 4 == 0 && ((fa.K0 + 15) / 16) % (bmx == 16 ? 8 : 4) == 0); };
    int bm = 0;
    if (N % 16 == 0 && B % 16 == 0 && N / 16 <= ncu3 && obs_ok(16)) bm = 16;
    else if (N % 32 == 0 && B % 32 == 0 && N / 32 <= 2 * ncu3 && obs_ok(32)) bm = 32;
    else if (N % 16 == 0 && B % 16 == 0 && N / 16 <= 2 * ncu3 && obs_ok(16)) bm = 16;
    fa.bm = bm;
    if (bm > 0 && fwd3_takes(fa)) {
      Stage &fs = b.func_stage("enc_joiner_actors", [=](hipStream_t s) { return fwd3_launch(fa, s); });
      fs.mfma = true;
      fs.prof = bm == 16 ? "fwd3<16>:" : "fwd3<32>:";
      fs.flops = fwd3_flops(fa);
      fs.bytes = 4.0 * ((double)N * (fa.K0 + 4 * F3_W) + (double)M * (F3_W + 2 * fa.P));
      enc_chained = true;
    }
  }
  for (size_t t = 0; t < bms.size() && !enc_chained && (chain_all || chain_on) && !gru; ++t) {
    Stage cs;
    cs.kind = ST_CHAIN; cs.name = "enc_joiner_actors"; cs.chain_bm = bms[t];
    ChainBuilder cb(cs);
    cb.begin(N);
    ChainImg xi = cb.load(eo.in);
    ChainImg ei = cb.mlp(eo, {xi}, true, true, {0, N, 0}, true);
    ChainImg si = cb.mlp(jo, {ei}, true, true, {0, N, 0}, true);
    cb.mlp(ao, {si}, false, true, {0, M, 0}, false);
    cb.mlp(at, {si}, true, false, {B, N, B}, false);
    cb.end();
    if (cb.ok) { a->stages.push_back(cs); enc_chained = true; }
  }
  if (enc_chained) {
    // nothing left to launch for these three networks
  } else {
  fwd_chain({&eo}, "enc_obs");
  if (!gru) {
    fwd_chain({&jo}, "joiner");
  } else {
    // GRU joiner (encoder.py:40-42, 63-65): the input projection of all T*B rows is one GEMM; the scan over t is
    // T x (recurrent GEMM [B, L] x [L, 3L] + gate kernel) - sequential by nature and latency-bound at B = 256
    const int L3 = 3 * L, F = c.enc_features, T = a->T;
    float *gi = a->buf("gru.gi"), *gh = a->buf("gru.gh"), *hprev = a->buf("gru.hprev"), *h0 = a->buf("gru.h0");
    const float *wih = params + a->gru_wih, *whh = params + a->gru_whh, *bih = params + a->gru_bih, *bhh = params + a->gru_bhh;
    {
      Stage &gs = b.gemm_stage("gru.gi");
      GemmProblem p = Builder::new_gemm(N, L3, gi, L3);
      Builder::add_seg(p, eo.out, F, 1, wih, F, 1, F);
      p.bias = bih;
      gs.gemm.push_back(p);
    }
    {
      const int mode = c.gru_state_mode;
      const float *src = mode == 1 ? x.agent_state : (mode == 2 ? params + a->gru_h0 : nullptr);
      b.func_stage("gru.h0", [=](hipStream_t s) { return gru_h0_launch(mode, src, h0, B, L, s); });
    }
    const bool scan = gru_scan_takes(B, L);   // the whole scan as ONE persistent launch (gruscan.hip) instead of T x (GEMM + gate kernel)
    if (scan) {
      GruScanArgs ga;
      memset(&ga, 0, sizeof(ga));
      float *pf = a->buf("gru.wpack_f"), *pb = a->buf("gru.wpack_b");
      ga.T = T; ga.B = B; ga.L = L; ga.W = pf; ga.bhh = bhh; ga.gi = gi; ga.h0 = h0; ga.gh = gh; ga.state = state; ga.hprev = hprev;
      b.func_stage("gru.pack", [=](hipStream_t s) { return gru_pack_launch(whh, L, pf, pb, s); });
      b.func_stage("gru.scan", [=](hipStream_t s) { return gru_scan_fwd_launch(ga, s); });
    }
    for (int t = 0; t < T && !scan; ++t) {
      const float *hp = t == 0 ? h0 : state + (int64_t)(t - 1) * B * L;
      float *gh_t = gh + (int64_t)t * B * L3, *h_t = state + (int64_t)t * B * L, *hs_t = hprev + (int64_t)t * B * L;
      const float *gi_t = gi + (int64_t)t * B * L3;
      float *ghp = a->buf("gru.ghp");
      Stage &gs = b.gemm_stage("gru.gh");
      GemmProblem p = Builder::new_gemm(B, L3, ghp, L3);
      Builder::add_seg(p, hp, L, 1, whh, L, 1, L);
      p.ksplit = GRU_KSPLIT_FWD;
      p.split_stride = (long long)B * L3;
      gs.gemm.push_back(p);
      b.func_stage("gru.cell", [=](hipStream_t s) {
        return gru_cell_fwd_launch(gi_t, gh_t, ghp, GRU_KSPLIT_FWD, bhh, hp, h_t, hs_t, B, L, s);
      });
    }
  }
  fwd_chain({&at, &ao}, "actors");
  }
  // ---- policy sampling (gaussian_mlp.py:15-39)
  {
    PolicyFwdArgs p0{at.out, nullptr, nullptr, a->buf("next_action"), a->buf("next_log_pi"), 0u, nullptr, nullptr};
    PolicyFwdArgs p1{ao.out, nullptr, a->buf("noise_actor"), a->buf("pi"), a->buf("log_pi"), 1u,
                     c.discrete ? a->buf("action_onehot") : x.action, a->buf("pi_diff")};
    fdql_agent *ag = a;
    const PrepArgs pra = prep_args;
    const bool with_prep = fold_prep;
    b.func_stage("policy_fwd", [=](hipStream_t s) {
      PolicyFwdArgs q0 = p0, q1 = p1;
      q0.noise = ag->noise_t;
      q1.noise = ag->noise_a;
      return policy_fwd_launch(q0, q1, 2, M, A, dst, ag->seed, ag->cfg.discrete, s, with_prep ? &pra : nullptr);
    });
  }
  // ---- critics forward: target(next, a'), online(cur, a), frozen(cur, pi)
  {
    // critic_frozen is a copy of critic taken when the actor loss is formed (soft_actor_critic.py:142),
    // so both read the online weights and layer 0 of q(s, a) and q(s, pi) shares s.Ws: ONE problem per
    // critic accumulates cat(s, a), stores h0 of the online pass, continues with (pi - a).Wa and stores
    // h0 of the frozen pass (GemmProblem::emit_seg) - 10 layer-0 problems instead of 15.
    const size_t nh = a->critic[0].hid.size();
    // Head fusion: each hidden layer's launch also forms its part of the skip head's dot product (GemmProblem::hf_*),
    // so the head streams only cat(s, a) instead of every hidden activation again (584 -> 197 MB at config 2).
    // Needs the Q outputs of a critic to be 1, 2, 4 or 8 (the butterfly's group size).
    bool fuse = nh > 0 && plan_switches().head_fuse && (Q == 1 || Q == 2 || Q == 4 || Q == 8);
    float *hf_parts = a->buf("hf.parts"), *hf_sum = a->buf("hf.sum");
    const long long MQ = (long long)M * Q;
    auto inst_id = [&](int k, int which) { return 3 * k + which; };   // which: 0 target, 1 online, 2 frozen
    auto plane0 = [&](int layer) { int p = 0; for (int i = 0; i < layer; ++i) p += ((a->critic[0].hid[i] + 63) / 64) * 2; return p; };
    auto set_hf = [&](GemmProblem &p, const MlpInst &m, int layer, int inst, bool second) {
      if (!fuse) return;
      p.hf_w = m.HW() + b.head_col_of_hidden(*m.d, layer);
      p.hf_ldw = m.d->head_ld();
      p.hf_q = Q;
      float *out = hf_parts + ((long long)inst * a->hf_planes + plane0(layer)) * MQ;
      if (second) p.hf_out2 = out; else p.hf_out = out;
    };
    bool crit_chained = false;
    if (chain_all) {   // every critic instance as one chain program: cat(s, a) -> hidden layers -> skip head
      Stage cs;
      cs.kind = ST_CHAIN; cs.name = "critics.fwd";
      ChainBuilder cb(cs);
      for (int k = 0; k < C && cb.ok; ++k) {
        int which = 0;
        for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
          cb.begin(M);
          ChainImg xi = cb.load(m->in);
          cb.mlp(*m, {xi}, true, which != 0, {0, M, 0}, false);   // target activations are never needed again
          cb.end();
          ++which;
        }
      }
      if (cb.ok) { a->stages.push_back(cs); crit_chained = true; }
    }
    if (crit_chained) {
      // done
    } else if (nh > 0 && plan_switches().dual) {
      Stage &gs = b.gemm_stage("critics.fwd0");
      gs.try_rows = true;
      gs.hf_role = fuse ? 1 : 0;
      gs.gm_role = 1;
      for (int k = 0; k < C; ++k) {
        GemmProblem pt = b.fwd_layer(ct[k], 0);
        pt.emit_seg = pt.nseg - 1;   // no tail: rides in the same launch as the dual problems
        set_hf(pt, ct[k], 0, inst_id(k, 0), false);
        gs.gemm.push_back(pt);
        GemmProblem p = b.fwd_layer(co[k], 0);
        Builder::add_seg(p, a->buf("pi_diff"), A, 1, co[k].W(0) + L, a->critic[k].din, 1, A);
        p.emit_seg = p.nseg - 2;
        p.C2 = cf[k].h[0];
        p.ldc2 = a->critic[k].hid[0];
        if (!cf[k].gm.empty()) p.gm_out2 = cf[k].gm[0];
        set_hf(p, co[k], 0, inst_id(k, 1), false);
        set_hf(p, cf[k], 0, inst_id(k, 2), true);
        gs.gemm.push_back(p);
      }
      for (size_t i = 1; i < nh; ++i) {
        Stage &ls = b.gemm_stage("critics.fwd" + std::to_string(i));
        ls.try_rows = true;
        ls.hf_role = fuse ? 1 : 0;
        ls.gm_role = 1;
        for (int k = 0; k < C; ++k) {
          int which = 0;
          for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
            GemmProblem p = b.fwd_layer(*m, (int)i);
            set_hf(p, *m, (int)i, inst_id(k, which++), false);
            ls.gemm.push_back(p);
          }
        }
      }
      // The head's finish.  With the hidden layers' parts already formed (head fusion), what is left per instance is
      // cat(s, a) . Wh[:, inputs] + the parts + the bias: a row-per-wave kernel for all instances (k_head_finish)
      // instead of a partial-sum reduction launch plus a head GEMM streaming cat(s, a) through padded tiles.
      bool finished = false;
      if (fuse && C * Q <= 16 && A <= 16 && C <= HEAD_FINISH_MAX_SETS) {
        HeadFinishArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.M = M; ha.L = L; ha.A = A; ha.Q = Q; ha.planes = a->hf_planes; ha.ngroups = 2;
        bool okf = true;
        for (int k = 0; k < C; ++k)
          for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) okf = okf && m->in.size() == 2 && m->in[0].width == L && m->in[1].width == A;
        if (okf) {
          HeadFinishGroup &gc = ha.g[0], &gn = ha.g[1];   // group 0: s_cur (online + frozen, one weight set per critic); 1: s_next (targets)
          gc.s = co[0].in[0].ptr; gc.lds = co[0].in[0].ld; gc.nsets = C; gc.nvar = 2; gc.ldw = a->critic[0].head_ld();
          gc.a[0] = co[0].in[1].ptr; gc.lda[0] = co[0].in[1].ld; gc.out[0] = a->buf("q_pred"); gc.ldo[0] = Nq;
          gc.a[1] = cf[0].in[1].ptr; gc.lda[1] = cf[0].in[1].ld; gc.out[1] = a->buf("q_frozen"); gc.ldo[1] = Nq;
          gn.s = ct[0].in[0].ptr; gn.lds = ct[0].in[0].ld; gn.nsets = C; gn.nvar = 1; gn.ldw = a->critic[0].head_ld();
          gn.a[0] = ct[0].in[1].ptr; gn.lda[0] = ct[0].in[1].ld; gn.out[0] = a->buf("next_z"); gn.ldo[0] = Nq;
          for (int k = 0; k < C; ++k) {
            okf = okf && co[k].HW() == cf[k].HW() && a->critic[k].head_ld() == gc.ldw;   // frozen reads the online weights
            gc.Wh[k] = co[k].HW(); gc.bias[k] = co[k].HB();
            gn.Wh[k] = ct[k].HW(); gn.bias[k] = ct[k].HB();
            gc.parts[k][0] = hf_sum + (long long)inst_id(k, 1) * MQ;
            gc.parts[k][1] = hf_sum + (long long)inst_id(k, 2) * MQ;
            gn.parts[k][0] = hf_sum + (long long)inst_id(k, 0) * MQ;
          }
        }
        // few rows (temporal_len 2): the plane sum inside the finish (its 16-row waves add the planes as they read them) instead
        // of a reduction launch in front of it - one launch less; at many rows the pair of launches is the faster one (config 2:
        // 0.0105 + 0.0153 ms against 0.0278 ms folded)
        const bool sum_in_finish = okf && M <= 4096;
        if (sum_in_finish) {
          ha.sum_planes = 1;
          const long long inst_stride = (long long)a->hf_planes * MQ;
          for (int k = 0; k < C; ++k) {
            ha.g[0].parts[k][0] = hf_parts + inst_id(k, 1) * inst_stride;
            ha.g[0].parts[k][1] = hf_parts + inst_id(k, 2) * inst_stride;
            ha.g[1].parts[k][0] = hf_parts + inst_id(k, 0) * inst_stride;
          }
        }
        if (okf) {
          const int ninst = 3 * C, planes = a->hf_planes;
          if (!sum_in_finish)
            b.func_stage("critics.head_sum", [=](hipStream_t s) { return reduce_partials_batched_launch(hf_parts, ninst, planes, MQ, hf_sum, s); }).hf_role = 2;
          // the form the finish takes when every hidden layer's launch sums its own planes (decided in upload_tables): one plane
          // per layer, added while they are read
          HeadFinishArgs hp = ha;
          bool same = true;
          for (size_t i = 1; i < nh; ++i) same = same && a->critic[0].hid[i] == a->critic[0].hid[0];
          hp.sum_planes = 1;
          hp.planes = (int)nh;
          hp.plane_step = ((a->critic[0].hid[0] + 63) / 64) * 2;
          {
            const long long inst_stride = (long long)a->hf_planes * MQ;
            for (int k = 0; k < C; ++k) {
              hp.g[0].parts[k][0] = hf_parts + inst_id(k, 1) * inst_stride;
              hp.g[0].parts[k][1] = hf_parts + inst_id(k, 2) * inst_stride;
              hp.g[1].parts[k][0] = hf_parts + inst_id(k, 0) * inst_stride;
            }
          }
          auto hap = std::make_shared<HeadFinishArgs>(ha);
          Stage &fs = b.func_stage("critics.head", [=](hipStream_t s) { return head_finish_launch(*hap, s); });
          fs.hf_role = 3; fs.hfin = hap; fs.hfin_plain = ha; fs.hfin_presum = hp;
          fs.hfin_can_presum = same && plan_switches().head_presum;
          finished = true;
        }
      }
      if (!finished) {
      if (fuse) {
        const int ninst = 3 * C, planes = a->hf_planes;
        b.func_stage("critics.head_sum", [=](hipStream_t s) { return reduce_partials_batched_launch(hf_parts, ninst, planes, MQ, hf_sum, s); });
      }
      Stage &hs = b.gemm_stage("critics.head");
      for (int k = 0; k < C; ++k) {
        int which = 0;
        for (MlpInst *m : {&ct[k], &co[k], &cf[k]}) {
          GemmProblem p = fuse ? b.fwd_head_inputs_only(*m) : b.fwd_head(*m);
          if (fuse) { p.epi = EPI_ADD_REF; p.ref = hf_sum + (long long)inst_id(k, which) * MQ; p.ldref = Q; }
          ++which;
          hs.gemm.push_back(p);
        }
      }
      }
    } else {
      std::vector<MlpInst *> g;
      for (int k = 0; k < C; ++k) { g.push_back(&ct[k]); g.push_back(&co[k]); g.push_back(&cf[k]); }
      fwd_chain(g, "critics");
    }
  }
  // ---- loss
  bool ride_finish = false;
  LossFinishArgs finish_args;
  memset(&finish_args, 0, sizeof(finish_args));
  {
    LossArgs la;
    memset(&la, 0, sizeof(la));
    la.M = M; la.B = B; la.Nq = Nq; la.Nt = a->Nt;
    int G = 8;
    while (G < Nq) G <<= 1;
    if (loss_wave_form(c.distributional, Nq)) G = 64;   // one wave per row, four rows per workgroup (kernels.hip, k_loss_wave)
    la.G = G;
    la.distributional = c.distributional; la.lowerbound = c.use_lowerbound; la.max_entropy = c.use_max_entropy;
    la.gamma = (float)c.gamma; la.target_entropy = -(float)A; la.half_inv_nq = (float)(0.5 / (double)Nq);
    la.st = dst; la.log_alpha = log_alpha;
    la.z_target = a->buf("next_z"); la.q_pred = a->buf("q_pred"); la.z_frozen = a->buf("q_frozen");
    la.logp_next = a->buf("next_log_pi"); la.logp = a->buf("log_pi");
    la.reward = x.reward; la.task_done = x.task_done; la.mc_return = c.use_lowerbound ? x.mc_return : nullptr;
    la.w = a->buf("w"); la.dz = a->buf("dz"); la.dzf = a->buf("dzf"); la.td_target = a->buf("td_target");
    la.q_loss = a->buf("q_loss"); la.pi_loss = a->buf("pi_loss"); la.alpha_loss = a->buf("alpha_loss");
    la.partials = a->buf("loss_partials");
    int nblocks = loss_blocks(M, G);
    // few workgroups (temporal_len 2): the last one to finish also sums the partial rows (kernels.hip, LossFinishArgs) - one
    // launch less; many workgroups would serialise on the arrival counter (round 2: ~80 ns per atomic)
    const bool fuse_finish = nblocks <= 64 && !c.bootstrap_nstep && plan_switches().small_folds;
    if (fuse_finish) {
      LossFinishArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.nblocks = nblocks; fa.M = M; fa.Nq = Nq; fa.st = dst; fa.scalars = a->buf("scalars");
      fa.dlog_alpha = a->buf("slabs") + a->log_alpha_off; fa.lr = c.lr; fa.b1 = c.beta1; fa.b2 = c.beta2;
      LossFinishArgs *fdev = reinterpret_cast<LossFinishArgs *>(a->buf("loss_fin_args"));
      FDQL_HIP(hipMemcpy(fdev, &fa, sizeof(fa), hipMemcpyHostToDevice));
      la.fin = fdev;
    }
    b.func_stage("loss", [=](hipStream_t s) { return loss_launch(la, s); });
    if (c.bootstrap_nstep) {   // soft_actor_critic.py:102-132: its loss value rides as one more partial row
      BootArgs ba;
      memset(&ba, 0, sizeof(ba));
      ba.T = a->T; ba.B = B; ba.Nq = Nq; ba.gamma = (float)c.gamma;
      ba.scale = (float)(1.0 / ((double)B * Nq * (c.world_size > 0 ? c.world_size : 1) * a->T));
      ba.reward = x.reward; ba.task_done = x.task_done; ba.contig = a->buf("is_contiguous");
      ba.td_target = a->buf("td_target"); ba.q_pred = a->buf("q_pred"); ba.dz = a->buf("dz");
      ba.partial_row = a->buf("loss_partials") + (int64_t)nblocks * LOSS_NPART;
      b.func_stage("boot_lowerbound", [=](hipStream_t s) { return boot_lowerbound_launch(ba, s); });
      ++nblocks;
    }
    float *scal = a->buf("scalars");
    float *dla = a->buf("slabs") + a->log_alpha_off;
    const float *parts = la.partials;
    // also the Adam bias corrections of the step about to be applied (torch.optim.Adam's Python floats)
    const double lr = c.lr, b1 = c.beta1, b2 = c.beta2;
    // single-process plans: the finish rides in the policy-backward launch (its next consumer is the optimiser); a two-bucket
    // plan needs d log_alpha in the early bucket, before that launch
    ride_finish = !fuse_finish && !a->bucketed() && plan_switches().small_folds;
    finish_args.nblocks = nblocks; finish_args.M = M; finish_args.Nq = Nq; finish_args.st = dst; finish_args.scalars = scal;
    finish_args.dlog_alpha = dla; finish_args.lr = lr; finish_args.b1 = b1; finish_args.b2 = b2;
    if (!fuse_finish && !ride_finish)
      b.func_stage("loss_finish", [=](hipStream_t s) { return loss_finish_launch(parts, nblocks, M, Nq, dst, scal, dla, lr, b1, b2, s); });
  }
  // ---- critic backward (online: wgrad + d state; frozen: d pi only)
  {
    const size_t nh = a->critic[0].hid.size();
    // Two hidden layers under a narrow head: the last layer's gradient (k_head_dgrad: a rank-Q outer product gated by
    // LeakyReLU') can be formed inside the loader of the row-block launch that consumes it (wstat.hip, k_wstat_grad<FUSE>) instead of
    // making a 2 x 131 MB round trip through HBM in a launch of its own.  Only when that launch takes the problems.
    bool fused1 = false;
    if (nh == 2 && Builder::narrow_head_last(a->critic[0], 1) && plan_switches().fuse_dpre1) {
      std::vector<GemmProblem> cand;
      for (int k = 0; k < C; ++k) {
        int which = 0;
        for (MlpInst *m : {&co[k], &cf[k]}) {
          GemmProblem p = b.bwd_dpre(*m, 0, a->buf(which == 0 ? "dz" : "dzf") + k * Q, Nq);
          p.fz_h = m->h[1];
          if (m->gm.size() > 1) p.gm_fz = m->gm[1];
          p.fz_w = m->HW() + b.head_col_of_hidden(*m->d, 1);
          p.fz_ldw = m->d->head_ld();
          p.fz_out = m->dpre[1];
          p.fz_colsum = m->dpre_cs[1];
          p.fz_discard = which == 1;   // frozen copy: its dpre1 feeds nothing but this GEMM
          cand.push_back(p);
          ++which;
        }
      }
      RowsLaunch rl;
      if (M % WS_BM == 0 && rows_launch_of(a, cand, rl)) {   // (32-row tiles: 1 568 rows - config 2 at 32 windows - are 49 of them)
        Stage &gs = b.gemm_stage("critics.dpre1+0");
        gs.try_rows = true;
        gs.gm_role = 2;
        gs.gemm = cand;
        fused1 = true;
        if (rl.ws)   // the weight-stationary launch leaves one partial row of column sums per workgroup
          for (int k = 0; k < C; ++k)
            for (MlpInst *m : {&co[k], &cf[k]}) m->dpre_cs_rows[0] = m->dpre_cs_rows[1] = wstat_colsum_rows(rl.wa);
      }
    }
    // few rows (temporal_len 2): the last layer's rank-Q product as GEMM problems on the small-batch kernel (one load round per
    // workgroup) instead of the streaming kernel's row loop: 0.022 -> 0.011 ms at 256 rows
    const bool head_dgrad_as_gemm = a->small_max_tiles > 0 && gemm_dense_shape() == GEMM_64x64 &&
                                    (long long)2 * C * ((M + 63) / 64) * ((a->critic[0].hid.empty() ? 0 : a->critic[0].hid.back()) + 63) / 64 <= 4LL * a->small_max_tiles;
    // Will the forward launches leave gate masks?  (the question upload_tables answers for good: every forward layer of the
    // critics as one weight-stationary launch; a stage that relies on the answer is checked there - Stage::needs_masks)
    bool masks_planned = plan_switches().gate_masks;
    for (const Stage &fs : a->stages) {
      if (fs.gm_role != 1 || !masks_planned) continue;
      RowsLaunch rl;
      masks_planned = fs.gemm.size() > 1 && rows_launch_of(a, fs.gemm, rl) && rl.ws;
    }
    for (int i = (int)nh - 1; i >= 0 && !fused1; --i) {
      // the last hidden layer under a head of up to 32 outputs, gated by the masks (config 4: 25 quantiles; the rank-25 product was
      // a tile launch at 25 TF that read h again): FDQL_NO_HEAD_DGRAD_MASKED keeps the GEMM stage
      if (i == (int)nh - 1 && masks_planned && Q > HEAD_DGRAD_MAXQ && Q <= HDM_MAXQ && a->critic[0].hid[i] == 256 && M % 32 == 0 && 2 * C <= HDM_MAX_INST &&
          !co[0].gm.empty() && plan_switches().head_dgrad_masked) {
        HeadDgradMaskedArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.M = M; ha.Q = Q; ha.ninst = 2 * C; ha.lddy = Nq; ha.ldw = a->critic[0].head_ld();
        for (int k = 0; k < C; ++k) {
          int w = 0;
          for (MlpInst *m : {&co[k], &cf[k]}) {
            const int n = 2 * k + w;
            ha.dY[n] = a->buf(w == 0 ? "dz" : "dzf") + k * Q;
            ha.Wh[n] = m->HW() + b.head_col_of_hidden(*m->d, i);
            ha.gm[n] = m->gm[i];
            ha.dpre[n] = m->dpre[i];
            ha.colsum[n] = m->dpre_cs[i];
            ++w;
          }
        }
        Stage &fs = b.func_stage("critics.dpre" + std::to_string(i) + "(masked)", [=](hipStream_t s) { return head_dgrad_masked_launch(ha, s); });
        fs.needs_masks = true;
        // the same product as a GEMM stage gated by h itself (same outputs, same per-64-row column sums): switched on by
        // upload_tables instead of the masked launch when the forward launches of the final plan turn out not to write masks
        Stage &fb = b.gemm_stage("critics.dpre" + std::to_string(i) + "(unmasked)");
        fb.masks_fallback = true;
        fb.off = true;
        for (int k = 0; k < C; ++k) {
          fb.gemm.push_back(b.bwd_dpre(co[k], i, a->buf("dz") + k * Q, Nq));
          fb.gemm.push_back(b.bwd_dpre(cf[k], i, a->buf("dzf") + k * Q, Nq));
        }
        continue;
      }
      if (Builder::narrow_head_last(a->critic[0], i) && !head_dgrad_as_gemm) {
        Stage st;
        st.kind = ST_HEAD_DGRAD; st.name = "critics.dpre" + std::to_string(i);
        for (int k = 0; k < C; ++k) {
          st.hdg.push_back(b.bwd_dpre_head(co[k], i, a->buf("dz") + k * Q, Nq));
          st.hdg.push_back(b.bwd_dpre_head(cf[k], i, a->buf("dzf") + k * Q, Nq));
        }
        a->stages.push_back(st);
        continue;
      }
      Stage &gs = b.gemm_stage("critics.dpre" + std::to_string(i));
      gs.try_rows = true;
      gs.gm_role = 2;
      for (int k = 0; k < C; ++k) {
        gs.gemm.push_back(b.bwd_dpre(co[k], i, a->buf("dz") + k * Q, Nq));
        gs.gemm.push_back(b.bwd_dpre(cf[k], i, a->buf("dzf") + k * Q, Nq));
      }
      {
        RowsLaunch rl;
        std::vector<GemmProblem> grp = gs.gemm;
        if (rows_launch_of(a, grp, rl) && rl.ws)
          for (int k = 0; k < C; ++k)
            for (MlpInst *m : {&co[k], &cf[k]}) m->dpre_cs_rows[i] = wstat_colsum_rows(rl.wa);
      }
    }
    // d pi: input-grad of each frozen critic's action columns as its own narrow (128x32) problem
    // -> C partials [C][M][A], summed in fixed order by the policy backward kernel
    {
      Stage &gs = b.gemm_stage("dpi");
      for (int k = 0; k < C; ++k) {
        GemmProblem p = Builder::new_gemm(M, A, a->buf("dpi_part") + (int64_t)k * M * A, A);
        b.input_grad_segs(cf[k], a->buf("dzf") + k * Q, Nq, L, p);
        gs.gemm.push_back(p);
      }
    }
  }
  // ---- data-parallel plans: the critics' weight gradients now, and their slab sum, so that the all-reduce of the arena
  // range [crit_begin, n_train) (critics + log_alpha: 2/3 of the arena at config 2) can run beside everything below
  const bool bucketed = a->bucketed();
  a->grad_bucket = bucketed ? a->crit_begin : a->n_train;
  size_t first_rest_stage = 0;
  if (bucketed) {
    Stage cn, cws;
    cn.kind = ST_GEMM; cn.name = "wgrad.critics.narrow";
    cws.kind = ST_SKINNY_WGRAD; cws.name = "colsums.critics";
    for (int k = 0; k < C; ++k) b.wgrads(co[k], a->buf("dz") + k * Q, Nq, nullptr, cn, cn, cws);
    b.flush_wgrad_stat("wgrad.critics", cn);
    Stage cnw;
    cnw.kind = ST_SKINNY_WGRAD; cnw.stream = true; cnw.name = "wgrad.critics.stream";
    b.take_stream_wgrads(cn, cnw, cws);
    b.colsums_into_stream(cws, cnw);
    if (!cn.gemm.empty()) a->stages.push_back(cn);
    if (!cnw.swg.empty()) a->stages.push_back(cnw);
    if (!cws.swg.empty()) a->stages.push_back(cws);
    const float *slabs = a->buf("slabs");
    float *grads = a->grads;
    const int S = a->nsplit;
    const long long P = a->n_train, first = a->crit_begin;
    b.func_stage("reduce_slabs.critics", [=](hipStream_t s) { return reduce_slabs_range_launch(slabs, S, P, first, P - first, grads, s); }).when = 1;
    first_rest_stage = a->stages.size();
  }
  // ---- policy backward
  bool fuse_pd = false;
  {
    const float *lo = ao.out, *nz = a->buf("noise_actor"), *pi = a->buf("pi"), *dpi = a->buf("dpi_part"), *w = a->buf("w");
    float *dlo = a->buf("dlogits"), *dpi_sum = a->buf("dpi");
    const int disc = c.discrete;
    const float *lparts = a->buf("loss_partials");
    const LossFinishArgs fa = finish_args;
    const bool ride = ride_finish;
    // the actor's last hidden layer under its narrow head: its pre-activation gradient in the same launch (FDQL_NO_POLICY_DPRE_FUSE:
    // a GEMM stage of its own, as before round 4)
    const int last = (int)a->actor.hid.size() - 1;
    fuse_pd = last >= 0 && policy_bwd_dpre_takes(disc, A, a->actor.hid[last]) && a->actor.dout == 2 * A && ao.dpre[last] &&
              plan_switches().policy_dpre_fuse;
    if (fuse_pd) {
      const float *Wh = ao.HW() + b.head_col_of_hidden(*ao.d, last), *h = ao.h[last];
      const int ldw = ao.d->head_ld();
      float *dpre = ao.dpre[last], *cs = ao.dpre_cs[last];
      b.func_stage("policy_bwd+actor.dpre" + std::to_string(last), [=](hipStream_t s) {
        return policy_bwd_dpre_launch(lo, nz, pi, dpi, C, dpi_sum, w, dst, M, A, dlo, Wh, ldw, h, dpre, cs, s, lparts, ride ? &fa : nullptr);
      });
    } else {
      b.func_stage("policy_bwd", [=](hipStream_t s) {
        return policy_bwd_launch(lo, nz, pi, dpi, C, dpi_sum, w, dst, M, A, dlo, disc, s, lparts, ride ? &fa : nullptr);
      });
    }
  }
  // ---- actor backward
  // The weight gradients of a network only need that network's own dpre/dY, so instead of one big
  // wgrad stage at the end they ride along with the small single-network dgrad launches that follow
  // (same tile shape -> same launch): those launches have only ~400 workgroups of their own.
  std::vector<size_t> hosts;  // stage indices of the dense dgrad launches after the critics' backward
  for (int i = (int)a->actor.hid.size() - 1; i >= 0; --i) {
    if (fuse_pd && i == (int)a->actor.hid.size() - 1) continue;   // formed by the policy backward's launch
    // (the rank-2A product on the streaming kernel k_head_dgrad - 12 broadcast LDS reads per element - measured 0.050 ms against
    // 0.015 ms for this K = 12 problem on MFMA tiles at config 2: it stays a GEMM problem)
    Stage &gs = b.gemm_stage("actor.dpre" + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(ao, i, a->buf("dlogits"), a->actor.dout));
    hosts.push_back(a->stages.size() - 1);
  }
  // ---- d state = sum over online critics and the actor
  {
    Stage &gs = b.gemm_stage("dstate");
    if (a->dstate_split) {
      float *parts = a->buf("dstate.parts");
      const long long ML = (long long)M * L;
      gs.try_rows = true;   // the critics' shares: weight-stationary plain dgrad form when there are enough rows
      for (int k = 0; k <= C; ++k) {
        GemmProblem p = Builder::new_gemm(M, L, parts + k * ML, L);
        if (k < C) b.input_grad_segs(co[k], a->buf("dz") + k * Q, Nq, 0, p);
        else b.input_grad_segs(ao, a->buf("dlogits"), a->actor.dout, 0, p);
        gs.gemm.push_back(p);
      }
    } else {
      GemmProblem p = Builder::new_gemm(M, L, a->buf("dstate"), L);
      for (int k = 0; k < C; ++k) b.input_grad_segs(co[k], a->buf("dz") + k * Q, Nq, 0, p);
      b.input_grad_segs(ao, a->buf("dlogits"), a->actor.dout, 0, p);
      p.colsum = a->buf("cs.dstate");
      gs.gemm.push_back(p);
    }
    hosts.push_back(a->stages.size() - 1);
  }
  const size_t idx_dstate = a->stages.size() - 1;
  // The sum of the shares inside the launch that consumes it first - the joiner's top hidden layer's dgrad on the row-block dgrad
  // kernel (rowdgrad.h, RowDgradArgs::sum_*): each 64-row workgroup adds its rows of the C + 1 shares while it stages them, writes
  // d state and its column sums; the summing launch (and one write + read of d state) goes away.  Asked here with the same
  // deterministic questions upload_tables asks, because the answer changes the column sums' row count.
  bool fold_dsum = false;
  // ... and with one hidden layer in the joiner and in the encoder the three launches behind d state become one (k_rowdgrad_chain):
  // its stages may have more blocks than one round of workgroups (784 at config 4, B = 1024: 0.379 ms against 0.397 for the summing
  // launch + three tile launches; a lone row-block dgrad launch of that size loses against the tile kernel)
  const bool chain_planned = a->joiner.hid.size() == 1 && a->enc_obs.hid.size() == 1 && c.enc_features == L && plan_switches().rowdgrad_chain;
  const int rd_max_blocks = chain_planned ? std::max(a->rowdgrad_max_blocks, 1024) : a->rowdgrad_max_blocks;
  // Round 6: the same launch on 32- or 16-row blocks (k_rowdgrad_chain<2> / <1>) where 64-row blocks are a fraction of a dispatch
  // round - one rank's share of a data-parallel batch (config 2 at 128 / 64 / 32 windows: 98 / 49 / 24.5 blocks of 64) and
  // temporal_len 2 - instead of the summing launch + three tile / small-batch launches.  chain_bm: 0 = the 64-row rules above.
  int chain_bm = 0;
  if (a->dstate_split && L % 4 == 0 && !gru && !a->joiner.hid.empty()) {
    MlpInst jq = jo;
    jq.rows = M;
    GemmProblem p = b.bwd_dpre(jq, (int)a->joiner.hid.size() - 1, a->buf("dstate"), L);
    RowDgradArgs tmp;
    fold_dsum = M / RD_BM >= a->rowdgrad_min_blocks && M / RD_BM <= rd_max_blocks && (M >= a->rowdot_min_rows ? p.N > 16 : true) &&
                rowdgrad_from_problem(p, tmp) &&
                rowdgrad_fold_sum(tmp, a->buf("dstate.parts"), C + 1, (long long)M * L, a->buf("dstate"), a->buf("cs.dstate"));
    // Rows per workgroup of the chain launch (measured, profiles/r06_rowdgrad_chain_blocks.txt): 64 while the blocks are 0.5 .. 1
    // dispatch rounds; 32 beyond one round (config 4 at B = 1024, 784 blocks of 64: 0.377 -> 0.308 ms) and at 6 272 rows; 16 at
    // 3 136 / 1 568 rows; below 1 024 rows (temporal_len 2) the small-batch kernel's launches carry the weight gradients as
    // riders and stay
    int want = RD_BM;
    if (M / RD_BM < a->rowdgrad_min_blocks) want = M < 1024 ? 0 : ((M % 16 == 0 && M / 16 <= 256) ? 16 : (M % 32 == 0 ? 32 : (M % 16 == 0 ? 16 : 0)));
    else if (M / RD_BM > 256 && M % 32 == 0) want = 32;
    if (chain_planned && plan_switches().rowdgrad && want > 0 && want < RD_BM) {
      const int bm = want;
      // the other two members, asked like the first (upload_tables builds the same three launches)
      MlpInst eq = eo;
      eq.rows = M;
      GemmProblem p2 = Builder::new_gemm(M, c.enc_features, a->buf("denc"), c.enc_features);
      b.input_grad_segs(jq, a->buf("dstate"), L, 0, p2);
      p2.colsum = a->buf("cs.denc");
      GemmProblem p3 = b.bwd_dpre(eq, (int)a->enc_obs.hid.size() - 1, a->buf("denc"), c.enc_features);
      RowDgradArgs t1, t2, t3;
      RowChainArgs tc;
      if (bm > 0 && bm < RD_BM && rowdgrad_from_problem(p, t1, bm) && rowdgrad_from_problem(p2, t2, bm) && rowdgrad_from_problem(p3, t3, bm) &&
          rowdgrad_fold_sum(t1, a->buf("dstate.parts"), C + 1, (long long)M * L, a->buf("dstate"), a->buf("cs.dstate")) &&
          rowchain_from_launches(t1, t2, t3, tc, bm)) {
        chain_bm = bm;
        fold_dsum = true;
      }
    }
  }
  if (a->dstate_split && !fold_dsum) {
    const float *parts = a->buf("dstate.parts");
    float *dsum = a->buf("dstate");
    const long long ML = (long long)M * L;
    const int np = C + 1;
    if (L % 4 == 0) {   // the sum also leaves the column sums the joiner head's bias gradient is reduced from
      float *csd = a->buf("cs.dstate");
      b.func_stage("dstate.sum", [=](hipStream_t s) { return sum_parts_colsum_launch(parts, np, M, L, dsum, csd, s); });
    } else {
      b.func_stage("dstate.sum", [=](hipStream_t s) { return reduce_partials_launch(parts, np, ML, dsum, s); });
    }
  }
  // column sums of d state: per 64 rows from the one-problem GEMM or the folded sum, per 32 rows from the summing launch, else
  // straight from d state
  const float *cs_dstate = (a->dstate_split && L % 4) ? nullptr : a->buf("cs.dstate");
  const int cs_dstate_rows = a->dstate_split ? (chain_bm ? M / chain_bm : (fold_dsum ? (M + 63) / 64 : (M + 31) / 32)) : 0;
  // ---- encoder backward over the M rows that carry gradient (next-only rows get none)
  MlpInst jb = jo, eb = eo;
  jb.rows = M; eb.rows = M;
  if (chain_bm) {   // the chain launch leaves one row of column sums per block
    jb.dpre_cs_rows[a->joiner.hid.size() - 1] = M / chain_bm;
    eb.dpre_cs_rows[a->enc_obs.hid.size() - 1] = M / chain_bm;
  }
  for (int i = (int)a->joiner.hid.size() - 1; i >= 0; --i) {
    Stage &gs = b.gemm_stage((fold_dsum && i == (int)a->joiner.hid.size() - 1 ? "dstate.sum+joiner.dpre" : "joiner.dpre") + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(jb, i, a->buf("dstate"), L));
    if (fold_dsum && i == (int)a->joiner.hid.size() - 1) {
      gs.fold_sum = true; gs.fold_parts = a->buf("dstate.parts"); gs.fold_n = C + 1; gs.fold_stride = (long long)M * L;
      gs.fold_out = a->buf("dstate"); gs.fold_cs = a->buf("cs.dstate");
      gs.rd_max_blocks = rd_max_blocks;
      gs.rd_chain_bm = chain_bm;
    }
    if (!chain_bm) hosts.push_back(a->stages.size() - 1);   // (a member of the small-block chain carries no riders: it must stay a lone problem)
  }
  if (gru) {
    // back-propagation through time over the T-1 rows that carry gradient (h_{T-1} only feeds no_grad targets):
    //   dh_t = d state[t] + dh_{t+1} * z_{t+1} + d gh_{t+1} W_hh
    const int L3 = 3 * L, T = a->T;
    float *dgi = a->buf("gru.dgi"), *dgh = a->buf("gru.dgh"), *dhw = a->buf("gru.dhw");
    float *dhz[2] = {a->buf("gru.dhz0"), a->buf("gru.dhz1")};
    const float *gi = a->buf("gru.gi"), *gh = a->buf("gru.gh"), *hprev = a->buf("gru.hprev"), *dstate = a->buf("dstate");
    const float *whh = params + a->gru_whh;
    const bool scan = gru_scan_takes(B, L);
    if (scan) {
      float *dh_init = a->buf("gru.dh_init");
      GruScanArgs ga;
      memset(&ga, 0, sizeof(ga));
      ga.T = T; ga.B = B; ga.L = L; ga.W = a->buf("gru.wpack_b"); ga.gi = gi; ga.gh = const_cast<float *>(gh); ga.hprev = const_cast<float *>(hprev);
      ga.dstate = dstate; ga.dgi = dgi; ga.dgh = dgh; ga.dh_init = c.gru_state_mode == 2 ? dh_init : nullptr;
      b.func_stage("gru.scan_bwd", [=](hipStream_t s) { return gru_scan_bwd_launch(ga, s); });
      if (c.gru_state_mode == 2) {
        float *out = a->buf("slabs") + a->gru_h0;
        b.func_stage("gru.dh0", [=](hipStream_t s) { return gru_dh0_launch(dh_init, nullptr, 0, B, L, out, s); });
      }
    }
    for (int t = T - 2; t >= 0 && !scan; --t) {
      const int64_t r3 = (int64_t)t * B * L3, r1 = (int64_t)t * B * L;
      const bool last = t == T - 2;
      const float *ca = last ? nullptr : dhz[(t + 1) & 1], *cb = last ? nullptr : dhw;
      float *out_z = dhz[t & 1];
      b.func_stage("gru.cell_bwd", [=](hipStream_t s) {
        return gru_cell_bwd_launch(dstate + r1, ca, cb, GRU_KSPLIT_BWD, gi + r3, gh + r3, hprev + r1, dgi + r3, dgh + r3, out_z,
                                   B, L, s);
      });
      Stage &gs = b.gemm_stage("gru.dh_prev");
      GemmProblem p = Builder::new_gemm(B, L, dhw, L);
      Builder::add_seg(p, dgh + r3, L3, 1, whh, L, 0, L3);
      p.ksplit = GRU_KSPLIT_BWD;
      p.split_stride = (long long)B * L;
      gs.gemm.push_back(p);
    }
    if (c.gru_state_mode == 2 && !scan) {   // learned start state: d hidden_state = sum_b d h_{-1}
      float *out = a->buf("slabs") + a->gru_h0;
      const float *za = dhz[0];
      b.func_stage("gru.dh0", [=](hipStream_t s) { return gru_dh0_launch(za, dhw, GRU_KSPLIT_BWD, B, L, out, s); });
    }
  }
  {
    Stage &gs = b.gemm_stage("denc");
    GemmProblem p = Builder::new_gemm(M, c.enc_features, a->buf("denc"), c.enc_features);
    if (gru) Builder::add_seg(p, a->buf("gru.dgi"), 3 * L, 1, params + a->gru_wih, c.enc_features, 0, 3 * L);
    else b.input_grad_segs(jb, a->buf("dstate"), L, 0, p);
    p.colsum = a->buf("cs.denc");
    gs.gemm.push_back(p);
    if (fold_dsum) gs.rd_max_blocks = rd_max_blocks;
    gs.rd_chain_bm = chain_bm;
    if (!chain_bm) hosts.push_back(a->stages.size() - 1);
  }
  const size_t idx_denc = a->stages.size() - 1;
  for (int i = (int)a->enc_obs.hid.size() - 1; i >= 0; --i) {
    Stage &gs = b.gemm_stage("enc_obs.dpre" + std::to_string(i));
    gs.gemm.push_back(b.bwd_dpre(eb, i, a->buf("denc"), c.enc_features));
    if (fold_dsum) gs.rd_max_blocks = rd_max_blocks;
    gs.rd_chain_bm = chain_bm;
    if (!chain_bm) hosts.push_back(a->stages.size() - 1);
  }
  // ---- pixel encoder backward (M images): d features -> per layer [dW, db], d col -> col2im -> previous layer
  if (nconv) {
    const int last = nconv - 1, F2 = a->conv_feat;
    {
      Stage &gs = b.gemm_stage("conv.dfeat");
      GemmProblem p = Builder::new_gemm(M, F2, a->buf("conv" + std::to_string(last) + ".dpre"), F2);
      b.input_grad_segs(eb, a->buf("denc"), c.enc_features, conv_col0, p);
      p.epi = EPI_LRELU_GRAD;
      p.ref = a->buf("conv" + std::to_string(last) + ".out");
      p.ldref = F2;
      gs.gemm.push_back(p);
    }
    for (int i = last; i > 0; --i) {
      const fdql_agent::ConvLayer &Lc = a->conv[i];
      const ConvGeom g = Lc.g;
      const int K = g.C * g.k * g.k;
      const long long rows = (long long)M * g.OH * g.OW;
      if (Lc.fast_dgrad) {   // gather-form implicit GEMM: no d col matrix, no col2im
        ConvDgradArgs da;
        da.dpre = a->buf("conv" + std::to_string(i) + ".dpre"); da.W = params + Lc.w_off;
        da.act_prev = a->buf("conv" + std::to_string(i - 1) + ".out"); da.dprev = a->buf("conv" + std::to_string(i - 1) + ".dpre");
        da.nimg = M; da.g = g; da.cout = Lc.cout;
        Stage &ds = b.func_stage("conv.dgrad" + std::to_string(i), [=](hipStream_t s) { return conv_dgrad_launch(da, s); });
        ds.mfma = true;
        ds.flops = 2.0 * (double)rows * K * Lc.cout;
        ds.bytes = 4.0 * ((double)rows * Lc.cout + 2.0 * (double)M * g.C * g.H * g.W);
        continue;
      }
      float *dcol = a->buf("conv" + std::to_string(i) + ".dcol");
      Stage &gs = b.gemm_stage("conv.dcol" + std::to_string(i));
      GemmProblem p = Builder::new_gemm((int)rows, K, dcol, K);
      Builder::add_seg(p, a->buf("conv" + std::to_string(i) + ".dpre"), Lc.cout, 1, params + Lc.w_off, K, 0, Lc.cout);
      gs.gemm.push_back(p);
      const float *act_prev = a->buf("conv" + std::to_string(i - 1) + ".out");
      float *dprev = a->buf("conv" + std::to_string(i - 1) + ".dpre");
      const long long nimg = M;
      b.func_stage("conv.col2im", [=](hipStream_t s) { return col2im_mask_launch(dcol, act_prev, nimg, g, dprev, s); });
    }
  }
  // ---- weight gradients (K-split slabs) + column sums
  {
    Stage tail, ws;
    std::vector<std::function<hipError_t(hipStream_t)>> conv_post;   // conv weight / bias partials -> slab 0
    tail.kind = ST_GEMM; tail.name = "wgrad.enc";
    ws.kind = ST_SKINNY_WGRAD; ws.name = "colsums";
    // critics: spread over the dgrad launches that follow their backward (they are ready by then)
    if (!bucketed)
      for (int k = 0; k < C; ++k) b.wgrads(co[k], a->buf("dz") + k * Q, Nq, nullptr, a->stages[hosts[k % hosts.size()]], tail, ws);
    // actor: needs d logits / its dpre -> from the d state launch on
    b.wgrads(ao, a->buf("dlogits"), a->actor.dout, nullptr, a->stages[idx_dstate], tail, ws);
    // joiner: needs d state and its dpre -> the d enc launch; encoder MLP: needs d enc and its dpre -> the tail
    if (!gru) {
      b.wgrads(jb, a->buf("dstate"), L, cs_dstate, chain_bm ? tail : a->stages[idx_denc], tail, ws, cs_dstate_rows);
    } else {   // GRU weights: dW_hh = d gh^T h_prev, dW_ih = d gi^T e, biases = column sums (all over the M rows)
      const int L3 = 3 * L, F = c.enc_features;
      float *slab = a->buf("slabs");
      Stage &host = a->stages[idx_denc];
      b.wgrad_gemm(M, a->buf("gru.dgh"), L3, L3, a->buf("gru.hprev"), L, L, slab + a->gru_whh, L, host, tail);
      b.wgrad_gemm(M, a->buf("gru.dgi"), L3, L3, eo.out, F, F, slab + a->gru_wih, F, host, tail);
      b.wgrad_bias(M, a->buf("gru.dgh"), L3, L3, nullptr, slab + a->gru_bhh, ws);
      b.wgrad_bias(M, a->buf("gru.dgi"), L3, L3, nullptr, slab + a->gru_bih, ws);
    }
    b.wgrads(eb, a->buf("denc"), c.enc_features, a->buf("cs.denc"), tail, tail, ws, chain_bm ? M / chain_bm : 0);
    for (int i = 0; i < nconv; ++i) {   // conv weights: dW = d pre^T col over the M*OH*OW rows, bias = column sums
      const fdql_agent::ConvLayer &Lc = a->conv[i];
      const int K = Lc.g.C * Lc.g.k * Lc.g.k;
      const int R = (int)((long long)M * Lc.g.OH * Lc.g.OW);
      float *slab = a->buf("slabs");
      const float *dpre = a->buf("conv" + std::to_string(i) + ".dpre");
      if (Lc.fast_wgrad) {   // output-stationary implicit GEMM: (dW, db) partials per slab, one reduction into slab 0
        ConvWgradArgs wa;
        wa.in.base = i == 0 ? (const void *)x.obs_2d_u8 : (const void *)a->buf("conv" + std::to_string(i - 1) + ".out");
        wa.in.u8 = i == 0; wa.in.slots = i == 0 ? x.obs_2d_slots : nullptr;
        wa.dpre = dpre; wa.wpart = a->buf("conv" + std::to_string(i) + ".wpart"); wa.nimg = M; wa.g = Lc.g; wa.cout = Lc.cout;
        const int nslab = conv_wgrad_slabs(Lc.g, Lc.cout, i == 0, M);
        const long long nw = (long long)Lc.cout * K + Lc.cout;
        FDQL_REQUIRE(nslab > 0 && a->named.at("conv" + std::to_string(i) + ".wpart").second >= nslab * nw, "conv layer %d: weight-gradient slabs", i);
        Stage &wst = b.func_stage("conv.wgrad" + std::to_string(i), [=](hipStream_t s) { return conv_wgrad_launch(wa, s); });
        wst.mfma = true;
        wst.flops = 2.0 * (double)R * K * Lc.cout;
        wst.bytes = (i == 0 ? 1.0 : 4.0) * (double)M * Lc.g.C * Lc.g.H * Lc.g.W + 4.0 * (double)R * Lc.cout;
        const float *wpart = wa.wpart;
        float *wdst = slab + Lc.w_off;   // (the bias follows its weights in the arena: checked in layout)
        conv_post.push_back([=](hipStream_t s) { return reduce_partials_launch(wpart, nslab, nw, wdst, s); });
        continue;
      }
      float *wpart = a->buf("conv" + std::to_string(i) + ".wpart"), *bpart = a->buf("conv" + std::to_string(i) + ".bpart");
      const int S2 = conv_wsplit(R);
      {
        GemmProblem p = Builder::new_gemm(Lc.cout, K, wpart, K);
        Builder::add_seg(p, dpre, Lc.cout, 0, a->buf("conv" + std::to_string(i) + ".col"), K, 0, R);
        p.ksplit = S2;
        p.split_stride = (long long)Lc.cout * K;
        tail.gemm.push_back(p);
      }
      float *wdst = slab + Lc.w_off, *bdst = slab + Lc.b_off;
      const long long nw = (long long)Lc.cout * K;
      const int cout = Lc.cout, nblk = colsum_tall_blocks(R);
      conv_post.push_back([=](hipStream_t s) {
        hipError_t e = reduce_partials_launch(wpart, S2, nw, wdst, s);
        if (e == hipSuccess) e = colsum_tall_launch(dpre, R, cout, cout, bpart, s);
        const int nblk2 = colsum_tall_blocks(nblk);     // second level: the [nblk, cout] partials are tall again
        float *bpart2 = bpart + (long long)nblk * cout;
        if (e == hipSuccess) e = colsum_tall_launch(bpart, nblk, cout, cout, bpart2, s);
        if (e == hipSuccess) e = reduce_partials_launch(bpart2, nblk2, cout, bdst, s);
        return e;
      });
    }
    b.flush_wgrad_stat("wgrad.dense", tail);
    Stage nws;
    nws.kind = ST_SKINNY_WGRAD; nws.stream = true; nws.name = "wgrad.stream";
    b.take_stream_wgrads(tail, nws, ws);
    b.colsums_into_stream(ws, nws);
    if (!tail.gemm.empty()) a->stages.push_back(tail);
    if (!nws.swg.empty()) a->stages.push_back(nws);
    if (!ws.swg.empty()) a->stages.push_back(ws);
    if (!conv_post.empty())
      b.func_stage("conv.wgrad_reduce", [=](hipStream_t s) {
        for (const auto &f : conv_post) { hipError_t e = f(s); if (e != hipSuccess) return e; }
        return hipSuccess;
      });
  }
  {
    const float *slabs = a->buf("slabs");
    float *grads = a->grads;
    const int S = a->nsplit;
    const long long P = a->n_train;
    // a split call (data-parallel: the all-reduce sits between the phases) sums the slabs into grads here; the
    // single-process step forms the sum inside k_adam_polyak
    const long long count = a->grad_bucket;   // a bucketed plan has summed [grad_bucket, P) already
    b.func_stage("reduce_slabs", [=](hipStream_t s) { return reduce_slabs_range_launch(slabs, S, P, 0, count, grads, s); }).when = 1;
  }
  // ---- Adam + polyak (+ frozen copy)
  {
    AdamArgs ad;
    memset(&ad, 0, sizeof(ad));
    ad.n = a->n_train; ad.params = a->params; ad.m = a->adam_m; ad.v = a->adam_v; ad.grads = a->grads;
    ad.grad_scale = 1.0f;
    ad.one_minus_b1 = (float)(1.0 - c.beta1); ad.b2 = (float)c.beta2; ad.one_minus_b2 = (float)(1.0 - c.beta2);
    ad.eps = (float)c.adam_eps; ad.st = dst; ad.targets = a->targets; ad.tgt_begin = a->tgt_begin; ad.tgt_end = a->tgt_end;
    ad.tau = (float)c.tau; ad.one_minus_tau = (float)(1.0 - c.tau); ad.hard = c.hard_updates;
    ad.frozen = c.keep_frozen_copy ? a->frozen : nullptr; ad.frozen_begin = a->crit_begin; ad.frozen_end = a->crit_end;
    b.func_stage("adam_polyak", [=](hipStream_t s) { return adam_launch(ad, s); }, FDQL_PHASE_APPLY).when = 1;
    AdamArgs af = ad;
    af.slabs = a->buf("slabs"); af.nslab = a->nsplit; af.grads_out = a->grads;
    b.func_stage("adam_polyak", [=](hipStream_t s) { return adam_launch(af, s); }, FDQL_PHASE_APPLY).when = 2;
  }
  for (size_t i = 0; i < a->stages.size(); ++i) a->stages[i].gpart = (bucketed && i < first_rest_stage) ? 0 : 1;
  int rc = upload_tables(a);
  if (rc) return rc;
  a->plan_ready = true;
  return 0;
}

}  // namespace fdql
