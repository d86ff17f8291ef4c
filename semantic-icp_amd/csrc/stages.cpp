// stages.cpp -- stage drivers of one align(): correspondence search, covariances / histograms, projections, EM weights, one
// evaluation sweep, the per-align preamble and the outer convergence test (em_icp.hpp:25-200).
#include "engine.hpp"

namespace sicp {
namespace host {

// queries: points [q_begin, q_begin+q_count) of cloud Q (device order), optionally transformed
// by M34; targets: segment `tseg` of cloud T.  Writes device indices of T (or -1) and distances.
int run_nn(sicp_context* h, int K, const Cloud& Qc, int q_begin, int q_count, const double* M34, const Cloud& Tc,
           int tseg, bool self, float gate_sq, int* out_i, float* out_d, int timer_bit, hipStream_t stream, int out_stride,
           const WeightFold* fold) {
  if (q_count <= 0) return SICP_OK;
  // the kernels run with a list of L >= K entries and write the first K (the K nearest neighbours
  // are the first K of any longer exact list): any K in 1..32 works
  const int L = sicp::nn_list_len(K);
  if (L == 0) return SICP_ERR_INVALID_ARGUMENT;
  const Cloud::SegTree& tr = Tc.trees[tseg];
  auto account = [&](double ms) {
    if (timer_bit == SICP_PROFILE_NN) { h->st.nn_kernel_ms += ms; h->st.nn_launches += 1; }
    else { h->st.cov_kernel_ms += ms; h->st.cov_launches += 1; }
  };
  if (h->params.nn_method >= 1) {
    sicp::KnnArgs a;
    a.qx = Qc.x.p; a.qy = Qc.y.p; a.qz = Qc.z.p;
    a.q_begin = q_begin; a.q_count = q_count;
    a.do_xform = M34 ? 1 : 0;
    for (int i = 0; i < 12; ++i) a.M[i] = M34 ? M34[i] : 0.0;
    a.tree.pts4 = Tc.pts4.p; a.tree.box_lo = Tc.box_lo.p; a.tree.box_hi = Tc.box_hi.p; a.tree.leaf_code = Tc.leaf_code.p;
    a.tree.lv = tr.lv; a.tree.n = tr.n; a.tree.pt_begin = tr.pt_begin; a.tree.node_begin = tr.node_begin;
    a.tree.code_begin = tr.code_begin;
    a.tree.lo[0] = tr.lo[0]; a.tree.lo[1] = tr.lo[1]; a.tree.lo[2] = tr.lo[2]; a.tree.scale = tr.scale;
    a.self = self ? 1 : 0;
    a.gate_sq = gate_sq;
    a.inv = Tc.inv.p;
    a.out_i = out_i; a.out_d = out_d;
    a.dbg = nullptr;
    a.out_stride = out_stride;
    a.k_out = K;
    a.live_cnt = nullptr;
    a.w_srec = a.w_trec = nullptr; a.w_sproj = a.w_tproj = nullptr; a.w_out = nullptr;
    a.w_one_m_eps = 0.0; a.w_C = 0; a.w_bool_probability = 0;
    if (fold && h->params.nn_method == 1) {
      a.w_srec = fold->srec; a.w_trec = fold->trec; a.w_sproj = fold->sproj; a.w_tproj = fold->tproj; a.w_out = fold->w;
      a.w_one_m_eps = fold->one_m_eps; a.w_C = fold->C; a.w_bool_probability = fold->bool_probability;
    }
    // seed hint: what the previous search of the same queries found (same clouds, same K, this align)
    a.seed_hint = (!self && h->hint_ok && out_i == h->idx.p && h->corr_K == K && h->corr_n == Qc.n) ? h->idx.p : nullptr;
    a.hint_K = K;
    a.t_begin = Tc.seg_off.empty() ? 0 : Tc.seg_off[tseg];
    static const bool want_dbg = debug_enabled() && std::getenv("SICP_KNN_STATS") != nullptr;  // developer aid (prints: needs SICP_DEBUG)
    DevBuf<int> dbg;
    if (want_dbg) { HIPCHECK(dbg.reserve((size_t)2 * q_count)); a.dbg = dbg.p; }
    static const bool lane_per_query = std::getenv("SICP_KNN_LANE_PER_QUERY") != nullptr;  // A/B aid
    // statistics: the packet kernel counts the neighbours that pass the gate as it writes them (spread over
    // kLiveCounters partial counters); every other engine leaves it to a kernel of its own (count_active)
    static const bool count_kernel = std::getenv("SICP_COUNT_KERNEL") != nullptr;  // A/B aid: always the separate kernel
    if (h->count_stats && !self && out_i == h->idx.p && !lane_per_query && !count_kernel && h->params.nn_method == 1) {
      a.live_cnt = (unsigned long long*)h->d_count.p;
      h->counted_in_search = true;
    }
    if (h->collect) {  // lock-step batch (packet search, no profiling: checked by the driver)
      JobCollector& jc = *h->collect;
      if (!jc.knn[jc.slice].empty() && jc.knn_K[jc.slice] != L) {
        h->last_error = "internal: searches of two list lengths collected into one slice";
        return SICP_ERR_INVALID_ARGUMENT;
      }
      jc.knn_K[jc.slice] = L;
      jc.knn[jc.slice].push_back(a);
      return SICP_OK;
    }
    KernelTimer kt(h, stream == h->stream ? timer_bit : 0);
    if (lane_per_query) HIPCHECK(sicp::launch_bvh_knn(L, a, stream));
    else if (h->params.nn_method == 2) HIPCHECK(sicp::launch_bvh_knn_quad(L, a, stream));
    else HIPCHECK(sicp::launch_bvh_knn_packet(L, a, stream));
    account(kt.stop());
    if (want_dbg) {
      std::vector<int> hd((size_t)2 * q_count);
      HIPCHECK(hipMemcpy(hd.data(), dbg.p, sizeof(int) * hd.size(), hipMemcpyDeviceToHost));
      double sn = 0, sl = 0; int mn = 0, ml = 0;
      for (int i = 0; i < q_count; ++i) { sn += hd[2 * i]; sl += hd[2 * i + 1]; mn = std::max(mn, hd[2 * i]); ml = std::max(ml, hd[2 * i + 1]); }
      // per-wave maxima drive the run time: a wave is as slow as its slowest lane
      double wsum_n = 0, wsum_l = 0; int nw = 0;
      for (int w0 = 0; w0 < q_count; w0 += 64, ++nw) {
        int wn = 0, wl = 0;
        for (int i = w0; i < std::min(q_count, w0 + 64); ++i) { wn = std::max(wn, hd[2 * i]); wl = std::max(wl, hd[2 * i + 1]); }
        wsum_n += wn; wsum_l += wl;
      }
      std::fprintf(stderr, "[sicp knn stats] K=%d self=%d n=%d: boxes/query avg %.1f max %d (wave-max avg %.1f), leaves/query avg %.1f max %d (wave-max avg %.1f)\n",
                   K, (int)self, q_count, sn / q_count, mn, wsum_n / nw, sl / q_count, ml, wsum_l / nw);
      // the walk is shared by the 16 queries of a packet: distribution of the packets' work (the launch
      // ends with its slowest packet)
      std::vector<int> pb, pl;
      for (int i = 0; i < q_count; i += 16) { pb.push_back(hd[2 * i]); pl.push_back(hd[2 * i + 1]); }
      std::sort(pb.begin(), pb.end()); std::sort(pl.begin(), pl.end());
      auto pct = [](const std::vector<int>& v, double f) { return v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
      std::fprintf(stderr, "[sicp knn stats]   per packet: boxes p50 %d p90 %d p99 %d p99.9 %d max %d | leaves p50 %d p90 %d p99 %d p99.9 %d max %d\n",
                   pct(pb, .5), pct(pb, .9), pct(pb, .99), pct(pb, .999), pb.back(), pct(pl, .5), pct(pl, .9), pct(pl, .99), pct(pl, .999), pl.back());
    }
    return SICP_OK;
  }
  const int t_count = tr.n;
  const int Q = sicp::nn_queries_per_thread(L);
  const int qblocks = (q_count + 256 * Q - 1) / (256 * Q);
  // >= ~8 workgroups per CU so the search fills the chip, but never chunks below one LDS tile
  int chunks = (2048 + qblocks - 1) / qblocks;
  const int max_chunks = (t_count + 1023) / 1024;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int chunk_len = (t_count + chunks - 1) / chunks;
  chunk_len = ((chunk_len + 1023) / 1024) * 1024;
  if (chunk_len < 1024) chunk_len = 1024;
  chunks = t_count > 0 ? (t_count + chunk_len - 1) / chunk_len : 1;
  const size_t need = (size_t)chunks * q_count * L;
  HIPCHECK(h->part.reserve(need));
  sicp::NNArgs a;
  a.qx = Qc.x.p; a.qy = Qc.y.p; a.qz = Qc.z.p;
  a.q_begin = q_begin; a.q_count = q_count;
  a.do_xform = M34 ? 1 : 0;
  for (int i = 0; i < 12; ++i) a.M[i] = M34 ? M34[i] : 0.0;
  a.pts4 = Tc.pts4.p;
  a.t_begin = tr.pt_begin; a.t_count = t_count;
  a.chunk_len = chunk_len;
  a.part = h->part.p;
  sicp::MergeArgs m;
  m.q_begin = q_begin; m.q_count = q_count; m.n_chunks = chunks;
  m.part = h->part.p;
  m.inv = Tc.inv.p;
  m.gate_sq = gate_sq;
  m.out_i = out_i; m.out_d = out_d;
  m.k_out = K;
  {
    KernelTimer kt(h, stream == h->stream ? timer_bit : 0);
    HIPCHECK(sicp::launch_nn_partial(L, a, chunks, stream));
    account(kt.stop());
  }
  HIPCHECK(sicp::launch_nn_merge(L, m, stream));
  return SICP_OK;
}

// ---- per-point normals (+ label histograms) ----------------------------------------------------
int ensure_hval(sicp_context* h, int k) {
  if (h->hval_k == k && h->d_hval.p) return SICP_OK;
  std::vector<double> hv(k + 1);
  const double increment = 1.0 / (double)k;  // em_icp.hpp:279
  double acc = 0.0;
  for (int c = 0; c <= k; ++c) { hv[c] = acc; acc += increment; }  // em_icp.hpp:301, repeated +=
  HIPCHECK(h->d_hval.reserve(k + 1));
  HIPCHECK(hipMemcpyAsync(h->d_hval.p, hv.data(), sizeof(double) * (k + 1), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  h->hval_k = k;
  return SICP_OK;
}

// projections of the label histograms through the confusion matrix (once per cloud per align)
int ensure_proj(sicp_context* h, Cloud& c) {
  const sicp_params& P = h->params;
  // the projections depend on the cloud's histograms, the confusion matrix and k: a cloud shared by
  // handles that hold the same matrix is projected once
  const unsigned long long want_id = h->cm_id * 1099511628211ull + (unsigned long long)P.k_cov;
  if (c.proj_valid && c.proj_cm_id == want_id) return SICP_OK;
  SICPCHECK(ensure_hval(h, P.k_cov));
  HIPCHECK(c.proj.reserve((size_t)(c.n > 0 ? c.n : 1) * sicp::proj_stride(P.num_classes)));
  sicp::ProjArgs a;
  a.n = c.n; a.C = P.num_classes;
  a.hist = c.hist.p; a.cm = h->d_cm.p; a.hval = h->d_hval.p; a.proj = c.proj.p;
  if (h->collect) h->collect->proj[h->collect->slice].push_back(a);
  else HIPCHECK(sicp::launch_proj(a, h->stream));
  c.proj_valid = true;
  c.proj_cm_id = want_id;
  return SICP_OK;
}

int compute_features(sicp_context* h, Cloud& c, bool with_hist, hipStream_t stream) {
  if (!stream) stream = h->stream;
  const sicp_params& P = h->params;
  const int k = P.k_cov, n = c.n;
  const size_t m = (size_t)(n > 0 ? n : 1);
  HIPCHECK(c.rec.reserve(m));
  // an empty cloud still has one (all-zero) record: the accumulate kernel evaluates dead slots on record 0
  // and weights them by exactly zero, which needs finite values there
  if (n == 0) HIPCHECK(hipMemsetAsync(c.rec.p, 0, sizeof(sicp::PointRec), stream));
  HIPCHECK(c.nn.reserve(m * k));
  if (with_hist) HIPCHECK(c.hist.reserve(m * (size_t)sicp::hist_stride(P.num_classes)));
  // the packet search writes the lists rank-major ([k][n]): coalesced stores there and coalesced
  // loads in the covariance kernel; the other engines keep [n][k]
  static const bool no_lane_per_query = std::getenv("SICP_KNN_LANE_PER_QUERY") == nullptr && !(debug_enabled() && std::getenv("SICP_KNN_STATS") != nullptr);
  const int nn_stride = (P.nn_method == 1 && no_lane_per_query) ? (int)m : 0;
  for (int s = 0; s < c.n_seg(); ++s) {
    const int o = c.seg_off[s], cnt = c.seg_off[s + 1] - o;
    SICPCHECK(run_nn(h, k, c, o, cnt, nullptr, c, s, true, std::numeric_limits<float>::infinity(), c.nn.p, nullptr,
                     SICP_PROFILE_COV, stream, nn_stride));
  }
  c.nn_stride = nn_stride;
  {
    // Test aid (needs SICP_DEBUG): the N-th call of this routine in the process fails HERE -- its self-search jobs are
    // queued, its covariance job is not -- with the status an exhausted arena gives.  A failure at this point cannot be
    // provoked from outside (a cloud's feature buffers are reserved when it is set), and it is the one that used to leave
    // the registration's OTHER cloud marked current with nothing computed (tests/test_gpu_stream.py).
    static const int fail_at = (debug_enabled() && std::getenv("SICP_FAULT_FEATURES_CALL")) ? std::atoi(std::getenv("SICP_FAULT_FEATURES_CALL")) : 0;
    static std::atomic<int> calls{0};
    if (fail_at > 0 && ++calls == fail_at) return SICP_ERR_OUT_OF_MEMORY;
  }
  sicp::CovArgs a;
  a.n = n; a.k = k; a.C = with_hist ? P.num_classes : 0;
  a.x = c.x.p; a.y = c.y.p; a.z = c.z.p;
  a.label = c.has_label ? c.label.p : nullptr;
  a.nn = c.nn.p;
  a.nn_stride = nn_stride;
  a.float_products = P.quirk_float_products;
  a.rec = c.rec.p;
  a.hist = with_hist ? c.hist.p : nullptr;
  // (SICP_NO_DENSE_SRC: developer switch; with it the accumulate kernel streams the 48-byte records: 814 instead of
  //  800 us per 256-pair launch, 2.18 instead of 2.22 G corr/s)
  static const bool dense_on = std::getenv("SICP_NO_DENSE_SRC") == nullptr;
  a.rec_dense = nullptr; a.rec_dense_n = 0;
  c.rec_dense_n = 0;
  if (dense_on && n > 0) {
    HIPCHECK(c.rec_dense.reserve(sicp::dense_rec_bytes(n)));
    a.rec_dense = c.rec_dense.p; a.rec_dense_n = n;
    c.rec_dense_n = n;
  }
  if (h->collect) h->collect->cov[h->collect->slice].push_back(a);
  else HIPCHECK(sicp::launch_cov(a, stream));
  c.feat_valid = true;
  c.cov_general = false;  // (what the engine computes is of its own form)
  c.proj_valid = false;
  c.feat_k = k; c.feat_C = with_hist ? P.num_classes : 0;
  c.feat_float_products = P.quirk_float_products;
  c.feat_hist = with_hist;
  c.feat_epoch = h->epoch;
  return SICP_OK;
}

bool features_current(const sicp_context* h, const Cloud& c, bool with_hist) {
  return c.feat_valid && c.feat_k == h->params.k_cov && c.feat_float_products == h->params.quirk_float_products &&
         (!with_hist || (c.feat_hist && c.feat_C == h->params.num_classes));
}

int check_ready(sicp_context* h, bool need_cm) {
  const sicp_params& P = h->params;
  if (!h->cloud(0).is_set || !h->cloud(1).is_set) return SICP_ERR_NOT_READY;
  if (!sicp::nn_k_supported(P.knn) || sicp::nn_list_len(P.k_cov) == 0) return SICP_ERR_INVALID_ARGUMENT;
  if (P.mode != SICP_MODE_GICP && (!h->cloud(0).has_label || !h->cloud(1).has_label)) return SICP_ERR_NOT_READY;
  if (P.mode == SICP_MODE_EM || need_cm) {
    if (P.num_classes < 1 || P.num_classes > 255 || h->C != P.num_classes) return SICP_ERR_NOT_READY;
    for (int wch = 0; wch < 2; ++wch) {
      const Cloud& c = h->cloud(wch);  // em_icp.hpp:301 indexes label-1
      if (c.n > 0 && (c.label_min < 1 || c.label_max > (uint32_t)P.num_classes)) return SICP_ERR_BAD_LABEL;
    }
  }
  if (P.mode != SICP_MODE_SEMANTIC && h->cloud(1).n < P.knn) return SICP_ERR_TOO_FEW_POINTS;
  return SICP_OK;
}

void fill_pose(const double* qt, sicp::Pose& p) {
  sicp::se3::rotation(qt, p.R);
  p.t[0] = qt[4]; p.t[1] = qt[5]; p.t[2] = qt[6];
}

int segment_of(const Cloud& c, uint32_t label) {
  for (int k = 0; k < c.n_seg(); ++k)
    if (c.seg_label[k] == label) return k;
  return -1;
}

// transform + kNN + gate (+ EM weight) at pose qt: the loop em_icp.hpp:46-108
int count_active(sicp_context* h);
int run_weights(sicp_context* h, const double* qt);

int run_correspondences(sicp_context* h, const double* qt, int K, bool weights) {
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  const size_t slots = (size_t)(S.n > 0 ? S.n : 1) * K;
  HIPCHECK(h->idx.reserve(slots));
  HIPCHECK(h->d2.reserve(slots));
  HIPCHECK(h->w.reserve(slots));
  double M[12];
  matrix34(qt, M);
  const bool sem = P.mode == SICP_MODE_SEMANTIC;
  h->counted_in_search = false;
  // EM-ICP, K = 4, at most 16 classes, packet search: the weights are written by the search's own epilogue
  // (knn_kernels.hip: KnnArgs::w_*; the same operations as em_weight_rows4_kernel, which then does not run) -- for a handle
  // on its own and in batches of at most 4 pairs (JobCollector::fold_weights says why not in larger ones).  Only when
  // the projections it reads are already there -- computed by an EARLIER flush, not waiting in this one (a flush launches
  // its searches first) -- and not for the developer variants that have a weight kernel of their own.
  // SICP_NO_WEIGHT_FOLD / SICP_WEIGHT_FOLD_ALWAYS (A/B aids): never / also in large batches and streams.
  WeightFold fold_args;
  const WeightFold* fold = nullptr;
  {
    static const bool no_fold = std::getenv("SICP_NO_WEIGHT_FOLD") != nullptr;
    static const bool lane_per_query = std::getenv("SICP_KNN_LANE_PER_QUERY") != nullptr;
    const unsigned long long want_id = h->cm_id * 1099511628211ull + (unsigned long long)P.k_cov;
    bool ok = weights && !no_fold && !lane_per_query && P.mode == SICP_MODE_EM && K == 4 && P.nn_method == 1 && P.profile == 0 &&
              P.num_classes >= 1 && P.num_classes <= 16 && !weights_from_histograms(P, K) && S.n_seg() == 1 &&
              S.proj_valid && T.proj_valid && S.proj_cm_id == want_id && T.proj_cm_id == want_id;
    if (ok && h->collect) {
      static const bool always = std::getenv("SICP_WEIGHT_FOLD_ALWAYS") != nullptr;  // A/B aid: also in large batches and streams
      ok = h->collect->fold_weights || always;
      for (int s = 0; s < kParts; ++s) ok = ok && h->collect->cov[s].empty() && h->collect->proj[s].empty();
    }
    if (ok) {
      fold_args.srec = S.rec.p; fold_args.trec = T.rec.p;
      fold_args.sproj = S.proj.p; fold_args.tproj = T.proj.p;
      fold_args.w = h->w.p;
      fold_args.one_m_eps = 1.0 - P.epsilon;
      fold_args.C = P.num_classes;
      fold_args.bool_probability = P.quirk_bool_probability;
      fold = &fold_args;
    }
  }
  if (sem) {  // label segments that are skipped keep (idx, d2) = (-1, +inf)
    HIPCHECK(hipMemsetAsync(h->idx.p, 0xFF, sizeof(int) * slots, h->stream));
    HIPCHECK(hipMemsetD32Async((hipDeviceptr_t)h->d2.p, 0x7f800000, slots, h->stream));
  }
  {
    const double t0 = now_ms();
    for (int s = 0; s < S.n_seg(); ++s) {
      const int so = S.seg_off[s], sn = S.seg_off[s + 1] - so;
      int ts = 0;
      if (sem) {
        ts = segment_of(T, S.seg_label[s]);
        if (ts < 0) continue;                    // semantic_icp.hpp:50
        if (!(sn > P.min_class_pts)) continue;   // semantic_icp.hpp:51
      }
      SICPCHECK(run_nn(h, K, S, so, sn, M, T, ts, false, (float)P.gate_sq, h->idx.p, h->d2.p, SICP_PROFILE_NN, h->stream, 0, fold));
      h->st.total_corr += (int64_t)sn * K;
    }
    h->st.t_nn_ms += now_ms() - t0;
  }
  h->corr_weighted = false;
  h->corr_n = S.n;
  h->corr_K = K;
  h->corr_valid = true;
  h->hint_ok = true;
  // statistics: the live slots of this search, counted right behind it (same stream / same job flush: no extra
  // host turn between two solves)
  if (h->count_stats && !h->counted_in_search) SICPCHECK(count_active(h));
  if (weights && fold) {
    h->corr_weighted = true;  // (written by the search itself)
    ++h->st.weights_in_search;
  } else if (weights) {
    SICPCHECK(run_weights(h, qt));
  }
  return SICP_OK;
}

// SICP_WEIGHTS_FROM_HIST (developer switch; K = 4 and at most 16 classes): the weight kernel reads the 16-byte label
// histograms themselves and forms the projections it needs (feature_kernels.hip: em_weight_hist4_body; same bits), no
// projection array is computed for align().  Built and measured in round 4 (profiles/r04/weights_from_histograms.json): the
// kernel alone takes 37 us instead of 13 (121 multiply-adds per slot against a 96-byte gather), and a 256-pair step takes the
// SAME time (186.3-186.6 against 184.0-186.6 ms) -- a kernel bound by the vector pipe runs beside the accumulate launches for
// free, the gather-bound one competes with them for the L1 miss queue.  Not the default: one pair alone pays the 24 us per search.
bool weights_from_histograms(const sicp_params& P, int K) {
  static const bool on = std::getenv("SICP_WEIGHTS_FROM_HIST") != nullptr;
  return on && P.mode == SICP_MODE_EM && K == 4 && P.num_classes >= 1 && P.num_classes <= 16;
}

// the EM weights of the current correspondences (em_icp.hpp:62-107): what run_correspondences(..., true) ends with
int run_weights(sicp_context* h, const double* qt) {
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  const int K = h->corr_K;
  if (P.mode == SICP_MODE_EM) {
    KernelTimer kt(h, SICP_PROFILE_WEIGHT);
    const double t0 = now_ms();
    sicp::WeightArgs a{};
    a.n_s = S.n; a.K = K; a.C = P.num_classes;
    a.idx = h->idx.p;
    a.srec = S.rec.p; a.trec = T.rec.p;
    if (weights_from_histograms(P, K)) {  // the kernel forms the projections it needs from the 16-byte count rows
      SICPCHECK(ensure_hval(h, P.k_cov));
      a.s_hist = S.hist.p; a.t_hist = T.hist.p;
      a.cm = h->d_cm.p; a.hval = h->d_hval.p;
    } else {
      SICPCHECK(ensure_proj(h, S));
      SICPCHECK(ensure_proj(h, T));
      a.s_proj = S.proj.p; a.t_proj = T.proj.p;
    }
    fill_pose(qt, a.pose);
    a.one_m_eps = 1.0 - P.epsilon;
    a.bool_probability = P.quirk_bool_probability;
    a.w = h->w.p;
    if (h->collect) h->collect->weight[h->collect->slice].push_back(a);
    else HIPCHECK(sicp::launch_em_weight(a, h->stream));
    h->st.weight_launches += 1;
    h->st.weight_kernel_ms += kt.stop();
    h->st.t_weight_ms += now_ms() - t0;
    h->corr_weighted = true;
  }
  return SICP_OK;
}

void fill_acc(sicp_context* h, sicp::AccArgs& a) {
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  a.n_s = h->corr_n; a.K = h->corr_K;
  a.idx = h->idx.p;
  a.w = h->corr_weighted ? h->w.p : nullptr;
  a.srec = S.rec.p; a.trec = T.rec.p;
  a.srec_dense = (S.rec_dense_n == S.n && S.n > 0 && S.n == h->corr_n) ? S.rec_dense.p : nullptr;
  a.lm = nullptr;
  a.lm_step = nullptr;
  a.one_m_eps = 1.0 - P.epsilon;
  a.loss.cauchy_a = P.cauchy_a;
  a.loss.use_sqloss = P.use_sqloss;
  a.partials = h->partials.p;
}

// pairs one launch evaluates (12 bytes of LDS each in the accumulate kernel).  SICP_MAX_ACTIVE: tuning aid.
const int kMaxActivePairs = [] { const char* e = std::getenv("SICP_MAX_ACTIVE"); const int v = e ? std::atoi(e) : 256; return std::min(std::max(v, 1), 512); }();

// One evaluation sweep at pose qt: the batched kernel on a batch of one (every path -- a pair alone, a
// lock-step batch, the host-loop solve, this hook -- runs the SAME accumulate kernel, so they agree bit
// for bit), then the fixed-order sum of the chunk partials.
bool general_covariances(const sicp_context* h) { return h->cloud(0).cov_general || h->cloud(1).cov_general; }

int eval28(sicp_context* h, const double* qt, double* out28) {
  SICPCHECK(batch_reserve(h, 1));
  const int nb = sicp::accumulate_blocks(h->corr_n * h->corr_K, h->corr_K);
  HIPCHECK(h->partials.reserve((size_t)nb * 28));
  h->ts[0].tick_valid = false;
  sicp::BatchArgs& B = h->ts[0].h_batch[0];
  std::memset(&B, 0, sizeof B);
  fill_acc(h, B.a);
  fill_pose(qt, B.a.pose);
  B.nb = nb;
  *h->ts[0].h_bhdr = sicp::BatchHeader{1, h->ts[0].epoch_host, {0, 0}};
  HIPCHECK(hipMemcpyAsync(h->ts[0].d_bhdr.p, h->ts[0].h_bhdr, sizeof(sicp::BatchHeader), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->ts[0].d_batch.p, h->ts[0].h_batch, sizeof(sicp::BatchArgs), hipMemcpyHostToDevice, h->stream));
  if (general_covariances(h)) {
    // caller covariances of general form: the literal cost function on full 3x3 matrices, the same columns for the same sum
    sicp::GenAccArgs g;
    g.a = B.a;
    g.scov6 = h->cloud(0).cov_general ? h->cloud(0).cov6.p : nullptr;
    g.tcov6 = h->cloud(1).cov_general ? h->cloud(1).cov6.p : nullptr;
    g.n_chunks = nb;
    KernelTimer kt(h, SICP_PROFILE_ACC);
    HIPCHECK(sicp::launch_accumulate_general(g, h->stream));
    h->st.acc_launches += 1;
    h->st.acc_kernel_ms += kt.stop();
  } else {
    KernelTimer kt(h, SICP_PROFILE_ACC);  // the accumulate kernel alone
    HIPCHECK(sicp::launch_accumulate_batch(h->corr_K, h->params.use_sqloss, h->ts[0].d_bhdr.p, h->ts[0].d_batch.p, std::min(h->ts[0].cap, kMaxActivePairs), h->stream));
    h->st.acc_launches += 1;
    h->st.acc_kernel_ms += kt.stop();
  }
  HIPCHECK(sicp::launch_finalize_batch(h->ts[0].d_batch.p, 1, h->d_bout28.p, h->stream));
  HIPCHECK(hipMemcpyAsync(h->h_bout28, h->d_bout28.p, sizeof(double) * 28, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  std::memcpy(out28, h->h_bout28, sizeof(double) * 28);
  h->st.total_evals++;
  return SICP_OK;
}

// per-align preamble: counters, cloud layout, covariances / histograms (asynchronous)
int align_begin(sicp_context* h, bool want_stats) {
  const sicp_params& P = h->params;
  const bool em = P.mode == SICP_MODE_EM, sem = P.mode == SICP_MODE_SEMANTIC;
  std::memset(&h->st, 0, sizeof h->st);
  h->hint_ok = false;  // every align() starts its first search from the curve position, like a first call would
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  HIPCHECK(h->d_count.reserve(sicp::kLiveCounters));
  if (want_stats) HIPCHECK(hipMemsetAsync(h->d_count.p, 0, sizeof(long long) * sicp::kLiveCounters, h->stream));
  h->count_stats = want_stats;
  SICPCHECK(prepare_cloud(h, S));
  SICPCHECK(prepare_cloud(h, T));
  // em_icp.hpp:28-29 / gicp.hpp:33-34 recompute the covariances on every align(); for
  // SemanticICP they belong to cloud construction (semantic_point_cloud.hpp:25-84)
  const double t0 = now_ms();
  // the two clouds' feature kernels are independent and latency bound: run them side by side
  // (not with brute force, which shares one scratch buffer, nor while those kernels are timed)
  const bool side_by_side = P.nn_method >= 1 && !(P.profile & SICP_PROFILE_COV);
  // A cloud is searched at most once per align() / align_batch() call (it may be shared by two
  // handles of a batch: one scan is the source of a pair and the target of the next), and not at
  // all when reuse_features is set and the features already belong to this cloud, k and C.
  auto stale = [&](const Cloud& c) {
    if (!features_current(h, c, em)) return true;
    if (sem || P.reuse_features) return false;
    return c.feat_epoch != h->epoch;
  };
  if (stale(S)) SICPCHECK(compute_features(h, S, em));
  if (stale(T)) {
    SICPCHECK(compute_features(h, T, em, side_by_side ? h->stream2 : h->stream));
    if (side_by_side && !h->collect) {
      HIPCHECK(hipEventRecord(h->ev_join, h->stream2));
      HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
  }
  if (em && !weights_from_histograms(P, P.knn)) {  // label distributions through the confusion matrix: same phase as the features they read
    SICPCHECK(ensure_proj(h, S));
    SICPCHECK(ensure_proj(h, T));
  }
  if (P.profile) HIPCHECK(hipStreamSynchronize(h->stream));
  h->st.t_cov_ms = now_ms() - t0;
  return SICP_OK;
}

// outer convergence test: em_icp.hpp:179-187 / gicp.hpp:153-161 / semantic_icp.hpp:151-158
void outer_finish(const sicp_params& P, OuterState& o) {
  double inv[7], rel[7], lg[6];
  sicp::se3::inverse(o.cur, inv);
  sicp::se3::mul(inv, o.est, rel);
  sicp::se3::log(rel, lg);
  double mse = 0;
  for (int i = 0; i < 6; ++i) mse += lg[i] * lg[i];
  if (P.mode == SICP_MODE_SEMANTIC) {
    if (mse < P.outer_tol || o.count > P.max_outer) o.converged = true;
    std::memcpy(o.cur, o.est, sizeof o.cur);
  } else {
    if (mse < P.outer_tol || o.outer > P.max_outer) o.converged = true;
    std::memcpy(o.cur, o.est, sizeof o.cur);
    o.outer++;
  }
}

int align_end(sicp_context* h, const OuterState& o, double t_begin, int32_t* outer_iters, sicp_stats* stats) {
  h->count_stats = false;
  if (stats) {
    HIPCHECK(hipMemcpyAsync(h->h_count, h->d_count.p, sizeof(long long) * sicp::kLiveCounters, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->st.total_active = 0;
    for (int k = 0; k < sicp::kLiveCounters; ++k) h->st.total_active += h->h_count[k];
  }
  h->st.outer_iters = h->params.mode == SICP_MODE_SEMANTIC ? o.count : o.outer;
  h->st.t_total_ms = now_ms() - t_begin;
  if (outer_iters) *outer_iters = h->st.outer_iters;
  if (stats) *stats = h->st;
  return SICP_OK;
}

// slice of the batch a pair belongs to: SICP_BATCH_PARTS contiguous slices of >= 2 pairs (default: 2 for
// K > 1, 4 for K = 1).  Measured at the end of round 2 (100K-point pairs; G corr/s at 1 / 2 / 3 / 4 slices):
// EM-ICP K = 4: 16 pairs 1.22 / 1.26 / 1.09 / 1.04, 64 pairs 1.49 / 1.55 / 1.55 / 1.46, 256 pairs 1.71 / 1.81 /
// 1.81 / 1.74 -- two streams of job launches overlap one slice's small kernels with the other's search
// tails, more of them only split the search launches into smaller, tail-bound ones; SE3-GICP K = 1 (cheap
// searches, short accumulate launches) at 256 pairs: 0.71 with 2 slices, 0.78 with 4.
int batch_slice(int p, int n, int knn) {
  static const int env = [] { const char* e = std::getenv("SICP_BATCH_PARTS"); return e ? std::atoi(e) : 0; }();
  const int want = env > 0 ? env : knn <= 1 ? 4 : 2;
  const int parts = std::max(1, std::min(std::min(want, kParts), n / 2));
  return (int)((long long)p * parts / n);
}

// launches what the pairs' stages collected: searches, then the kernels that consume them
int flush_jobs(sicp_context* h, JobCollector& jc, hipStream_t base) {
  if (!base) base = h->stream;
  // The slices of the batch run their stage sequences (searches -> covariances -> projections ->
  // weights -> counts) on their own streams: the small kernels and the search tails of one slice
  // overlap the searches of the others.
  bool used[kParts];
  int n_used = 0;
  for (int s = 0; s < kParts; ++s) {
    used[s] = !jc.knn[s].empty() || !jc.cov[s].empty() || !jc.proj[s].empty() || !jc.weight[s].empty() || !jc.count[s].empty();
    n_used += used[s];
  }
  const bool fork = n_used > 1 || (n_used == 1 && !used[0]);
  if (fork) {
    for (int s = 1; s < kParts; ++s)
      if (used[s] && !h->part_stream[s]) {
        HIPCHECK(hipStreamCreateWithFlags(&h->part_stream[s], hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&h->part_done[s], hipEventDisableTiming));
      }
    if (!h->part_fork) HIPCHECK(hipEventCreateWithFlags(&h->part_fork, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(h->part_fork, base));
  }
  for (int s = 0; s < kParts; ++s) {
    if (!used[s]) continue;
    hipStream_t st = s ? h->part_stream[s] : base;
    if (s) HIPCHECK(hipStreamWaitEvent(st, h->part_fork, 0));
    if (!jc.knn[s].empty()) HIPCHECK(sicp::launch_bvh_knn_packet_jobs(jc.knn_K[s], jc.knn[s].data(), (int)jc.knn[s].size(), st));
    if (!jc.cov[s].empty()) HIPCHECK(sicp::launch_cov_jobs(jc.cov[s].data(), (int)jc.cov[s].size(), st));
    if (!jc.proj[s].empty()) HIPCHECK(sicp::launch_proj_jobs(jc.proj[s].data(), (int)jc.proj[s].size(), st));
    if (!jc.weight[s].empty()) HIPCHECK(sicp::launch_em_weight_jobs(jc.weight[s].data(), (int)jc.weight[s].size(), st));
    if (!jc.count[s].empty()) HIPCHECK(sicp::launch_count_active_jobs(jc.count[s].data(), (int)jc.count[s].size(), st));
    jc.knn[s].clear(); jc.cov[s].clear(); jc.proj[s].clear(); jc.weight[s].clear(); jc.count[s].clear();
    if (s) {
      HIPCHECK(hipEventRecord(h->part_done[s], st));
      HIPCHECK(hipStreamWaitEvent(base, h->part_done[s], 0));
    }
  }
  return SICP_OK;
}

// getFusedLabels (em_icp.hpp:202-268): the K = 4 search at the final pose ...
int labels_search(sicp_context* h, const double* qt) {
  // the label kernel reads the projections of both clouds; align() no longer computes them when its weights come straight
  // from the histograms: queued here, with the search (collected into the same job flush inside a batch / stream)
  SICPCHECK(ensure_proj(h, h->cloud(0)));
  SICPCHECK(ensure_proj(h, h->cloud(1)));
  SICPCHECK(run_correspondences(h, qt, 4, false));  // K = 4 is a literal there (em_icp.hpp:221)
  h->corr_valid = false;                            // K may differ from params.knn
  h->hint_ok = false;
  return SICP_OK;
}

// ... and arg-max of the label posterior over the classes (em_icp.hpp:230-266), one lane per source point
int labels_launch(sicp_context* h, const double* qt, hipStream_t st) {
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  sicp::WeightArgs a{};
  a.n_s = S.n; a.K = 4; a.C = P.num_classes;
  a.idx = h->idx.p;
  a.srec = S.rec.p; a.trec = T.rec.p;
  SICPCHECK(ensure_proj(h, S));
  SICPCHECK(ensure_proj(h, T));
  a.s_proj = S.proj.p; a.t_proj = T.proj.p;
  fill_pose(qt, a.pose);
  a.one_m_eps = 1.0 - P.epsilon;
  a.bool_probability = P.quirk_bool_probability;
  a.w = nullptr;
  HIPCHECK(h->tmpl.reserve((size_t)(S.n > 0 ? S.n : 1)));
  HIPCHECK(sicp::launch_fused_labels(a, h->tmpl.p, st));
  return SICP_OK;
}

// statistics only: add the number of live slots of the current search to the device counter (run_correspondences)
int count_active(sicp_context* h) {
  if (h->collect) {
    h->collect->count[h->collect->slice].push_back(sicp::CountJob{h->idx.p, h->corr_n * h->corr_K, (unsigned long long*)h->d_count.p});
    return SICP_OK;
  }
  HIPCHECK(sicp::launch_count_active(h->idx.p, h->corr_n * h->corr_K, (unsigned long long*)h->d_count.p, h->stream));
  return SICP_OK;
}

}  // namespace host
}  // namespace sicp
