// solve.cpp -- the inner solve (em_icp.hpp:162-177) and the continuous batching of many pairs: ticks of [accumulate, LM step],
// the persistent launch of the last pair still iterating, BatchRun::turn, and the closed batch behind sicp_align_batch.
#include "engine.hpp"

namespace sicp {
namespace host {

constexpr size_t kSoloFallbackOffset = 64;  // bytes between the persistent solve's polled word and its fallback copy's landing area (one cache line)

sicp::LmOptions lm_options(const sicp_params& P) {
  sicp::LmOptions o;
  o.max_iterations = P.max_lm_iterations;
  o.gradient_tolerance = P.gradient_tolerance;
  o.function_tolerance = P.function_tolerance;
  o.parameter_tolerance = P.parameter_tolerance;
  o.initial_radius = P.initial_radius;
  o.max_radius = P.max_radius;
  o.min_radius = P.min_radius;
  o.min_relative_decrease = P.min_relative_decrease;
  o.min_lm_diagonal = P.min_lm_diagonal;
  o.max_lm_diagonal = P.max_lm_diagonal;
  o.max_consecutive_invalid_steps = P.max_consecutive_invalid_steps;
  o.jacobi_scaling = P.jacobi_scaling != 0 ? 1 : 0;
  return o;
}

// may the next launch of this leader be a persistent one?  (after a timed-out launch a number of them are not)
bool solo_allowed(sicp_context* h) {
  if (h->solo_skip > 0) { --h->solo_skip; return false; }
  return true;
}

// the inner ceres::Solve (em_icp.hpp:162-177) on the current correspondences
int run_solve(sicp_context* h, const double* init_qt, double* out_qt, SolveResult* res) {
  const sicp_params& P = h->params;
  if (!P.lm_on_device || general_covariances(h)) {
    // host loop: one kernel pair + one 224-byte read-back + one synchronisation per evaluation (also the path of a pair with
    // caller covariances of general form: eval28 evaluates it with the full-matrix kernel)
    sicp::LmState s;
    sicp::lm_init(s, lm_options(P), init_qt);
    while (s.status == sicp::LM_RUNNING) {
      double o[28];
      SICPCHECK(eval28(h, s.pose, o));
      sicp::lm_feed(s, o);
    }
    std::memcpy(out_qt, s.x, sizeof s.x);
    res->status = s.status; res->iterations = s.iterations; res->evaluations = s.evaluations; res->cost = s.cost;
    return SICP_OK;
  }
  // device-resident: the trust-region state stays in HBM; ticks of lm_batch evaluations (accumulate
  // kernel + one-wave LM step each) as ONE graph launch; the host looks at the state once per tick.
  // The very machinery of sicp_align_batch, with a batch of one.
  SICPCHECK(batch_reserve(h, 1));
  h->ts[0].tick_valid = false;
  const int len = std::min(P.lm_batch > 0 ? P.lm_batch : 12, sicp::kMaxBatchLen);
  sicp_handle self = h;
  double start[1][7];
  std::memcpy(start[0], init_qt, sizeof start[0]);
  std::vector<int> act(1, 0), joining(1, 0);
  for (;;) {
    const bool solo = P.lm_on_device != 2 && solo_allowed(h) && sicp::solve_one_fits(h->corr_n * h->corr_K, h->corr_K);
    SICPCHECK(run_tick(h, h->stream, &self, 1, act, joining, start, len, solo ? sicp::kSoloMaxEvals : 0));
    if (solo && h->solo_failed) continue;  // nothing has happened: the same step again as a tick
    joining.clear();
    h->st.acc_launches += solo ? 1 : len;
    if (h->h_bstates[0].status != sicp::LM_RUNNING) break;
  }
  const sicp::LmState& s = h->h_bstates[0];
  std::memcpy(out_qt, s.x, sizeof s.x);
  res->status = s.status; res->iterations = s.iterations; res->evaluations = s.evaluations; res->cost = s.cost;
  h->st.total_evals += s.evaluations;
  return SICP_OK;
}

// ---- lock-step batch -----------------------------------------------------------------------------
bool same_solver(const sicp_params& a, const sicp_params& b) {
  return a.mode == b.mode && a.knn == b.knn && a.lm_batch == b.lm_batch && a.use_sqloss == b.use_sqloss &&
         a.nn_method == b.nn_method && a.lm_on_device == b.lm_on_device && a.profile == b.profile && a.k_cov == b.k_cov;
}

int tickset_reserve(sicp_context* h, TickSet& S, int n) {
  n = std::max(32, (n + 31) / 32 * 32);  // capacity in steps of 32: the tick graph is keyed on it
  HIPCHECK(S.d_batch.reserve(n));
  HIPCHECK(S.d_bhdr.reserve(1));
  if (!S.h_bhdr) HIPCHECK(hipHostMalloc((void**)&S.h_bhdr, sizeof(sicp::BatchHeader), hipHostMallocDefault));
  HIPCHECK(S.d_join.reserve(n));
  if (S.cap < n) {
    if (S.h_batch) (void)hipHostFree(S.h_batch);
    if (S.h_join) (void)hipHostFree(S.h_join);
    S.h_batch = nullptr; S.h_join = nullptr; S.cap = 0;
    HIPCHECK(hipHostMalloc((void**)&S.h_batch, sizeof(sicp::BatchArgs) * n, hipHostMallocDefault));
    HIPCHECK(hipHostMalloc((void**)&S.h_join, sizeof(sicp::LmJoin) * n, hipHostMallocDefault));
    S.cap = n;
  }
  return SICP_OK;
}

// buffers for a batch of n pairs: per-pair LM states and sums (indexed by pair), tick set 0
int batch_reserve(sicp_context* h, int n) {
  SICPCHECK(tickset_reserve(h, h->ts[0], n));
  n = std::max(32, (n + 31) / 32 * 32);
  HIPCHECK(h->d_bstates.reserve(n));
  if (h->d_ein.cap < (size_t)2 * n) {  // (zeroed: epoch 0 never is a launch's epoch)
    HIPCHECK(h->d_ein.reserve((size_t)2 * n));
    HIPCHECK(hipMemsetAsync(h->d_ein.p, 0, sizeof(sicp::EvalIn) * h->d_ein.cap, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  }
  HIPCHECK(h->d_bout28.reserve((size_t)28 * n));
  if (h->h_batch_cap < n) {
    if (h->h_bstates) (void)hipHostFree(h->h_bstates);
    if (h->h_bout28) (void)hipHostFree(h->h_bout28);
    h->h_bstates = nullptr; h->h_bout28 = nullptr; h->h_batch_cap = 0;
    HIPCHECK(hipHostMalloc((void**)&h->h_bstates, sizeof(sicp::LmState) * n, hipHostMallocCoherent));  // (also written by the persistent solve's master)
    HIPCHECK(hipHostMalloc((void**)&h->h_bout28, sizeof(double) * 28 * n, hipHostMallocDefault));
    h->h_batch_cap = n;
  }
  return SICP_OK;
}

// Build-time experiment of round 6 (-DSICP_LM_STEP_IN_LAUNCH, then SICP_LM_STEP_IN_LAUNCH=1): the LM step of a tick inside its
// accumulate launches (solve_kernels.hip says what it measured: slower).  The product's tick is [accumulate, lm_step_batch] x len.
// The searches / weights / feature kernels of the pairs between two solves run on a stream of their own beside the ticks.
// SICP_SIDE_PRIORITY=low|high (developer A/B aid): that stream with the device's lowest / highest queue priority.
hipError_t create_side_stream(hipStream_t* st) {
  static const int mode = [] { const char* e = std::getenv("SICP_SIDE_PRIORITY"); return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'h' ? 2 : 0)); }();
  if (mode == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  int least = 0, greatest = 0;  // (numerically: greatest priority is the smaller number)
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) return e;
  return hipStreamCreateWithPriority(st, hipStreamNonBlocking, mode == 1 ? least : greatest);
}

bool lm_step_in_launch() {
#if defined(SICP_LM_STEP_IN_LAUNCH)
  static const bool on = std::getenv("SICP_LM_STEP_IN_LAUNCH") != nullptr;
  return on;
#else
  return false;
#endif
}

// One TICK of a batch: `len` LM evaluations of every pair in `act` (pair indices), in one graph launch:
// the accumulate kernel evaluates all of them at their current LM poses, lm_step_batch_kernel
// advances every pair's trust-region machine (csrc/lm.hpp, the same code and the same bits as for a
// pair alone).  `joining` pairs start their inner solve with this tick (their LM state is initialised
// and uploaded first).  tick_launch only queues work on stream M (ending with the read-back of the
// states of pairs [lo, hi) into h_bstates); the caller synchronises M when it wants the result.
int tick_launch(sicp_context* h, TickSet& S, hipStream_t M, sicp_handle* hs, int lo, int hi, const std::vector<int>& act,
                const std::vector<int>& joining, const double (*start)[7], int len, int solo_evals) {
  if (!joining.empty() && solo_evals <= 0) {  // their LM states are initialised on the device: one upload + one tiny kernel
    int k = 0;
    for (int p : joining) {
      sicp::LmJoin& J = S.h_join[k++];
      J.pair = p; J.pad_ = 0;
      std::memcpy(J.start, start[p], sizeof J.start);
      J.opt = lm_options(hs[p]->params);
    }
    HIPCHECK(hipMemcpyAsync(S.d_join.p, S.h_join, sizeof(sicp::LmJoin) * joining.size(), hipMemcpyHostToDevice, M));
    HIPCHECK(sicp::launch_lm_init(S.d_join.p, (int)joining.size(), h->d_bstates.p, M));
  }
  if (solo_evals > 0) {
    // The ONLY pair still iterating: (up to solo_evals evaluations of) its inner solve as one persistent launch
    // (solve_kernels.hip: solve_one_kernel) -- the chunk data stays in registers, two fence-free hand-offs per
    // evaluation instead of two kernel boundaries, one host look per launch instead of per tick.  Everything travels
    // in the kernel arguments: no argument upload, no state-initialisation kernel, no memset.
    const int p = act[0];
    sicp_context* g = hs[p];
    const int nb = sicp::accumulate_blocks(g->corr_n * g->corr_K, g->corr_K);
    if (g->partials.reserve((size_t)nb * 28) != hipSuccess) return SICP_ERR_OUT_OF_MEMORY;
    const int evals = std::min(solo_evals, sicp::kSoloMaxEvals);
    if (!h->d_solo_sync.p || h->solo_tag > 0xF0000000u) {
      HIPCHECK(h->d_solo_sync.reserve((size_t)sicp::kSoloSyncWords + 16));  // (+ the phase timers of a -DSICP_SOLO_TIMING build)
      HIPCHECK(hipMemsetAsync(h->d_solo_sync.p, 0, sizeof(unsigned) * (sicp::kSoloSyncWords + 16), M));
      h->solo_tag = 0;
    }
    sicp::SoloArgs A;
    std::memset(&A, 0, sizeof A);
    fill_acc(g, A.a);
    A.a.lm = A.a.lm_step = h->d_bstates.p + p;
    A.sync = h->d_solo_sync.p;
    A.max_evals = evals;
    A.wait_ticks = sicp::solo_wait_ticks();
    A.tag_base = h->solo_tag;
    h->solo_tag += (unsigned)evals + 1u;
    A.init = joining.empty() ? 0 : 1;
    static std::atomic<int> launches{0};  // process-wide: a recycled state buffer cannot hold the number by accident
    A.seq = h->solo_seq = ++launches;
    if (A.init) std::memcpy(A.start, start[p], sizeof A.start);
    A.opt = lm_options(g->params);
    // the state also lands in the pinned mirror by the master's own stores, and the launch number in a pinned word the host
    // polls (solo_wait); the copy queued behind the kernel stays as the fallback (a launch that gave up writes neither)
    if (!h->h_solo_flag) {
      // fine-grained: visible to the CPU while the kernel runs.  Behind the word: the landing area of the fallback copy
      HIPCHECK(hipHostMalloc((void**)&h->h_solo_flag, kSoloFallbackOffset + sizeof(sicp::LmState), hipHostMallocCoherent));
      *h->h_solo_flag = 0;
    }
    static const bool no_poll = std::getenv("SICP_SOLO_NO_HOST_POLL") != nullptr;  // A/B aid: wait for the read-back copy as before
    A.host_state = no_poll ? nullptr : static_cast<sicp::LmCore*>(h->h_bstates + p);
    A.host_flag = no_poll ? nullptr : h->h_solo_flag;
    h->solo_pair = p;
    h->solo_was_init = A.init != 0;
    S.tick_valid = false;  // (the argument array in HBM was not refreshed)
    HIPCHECK(sicp::launch_solve_one(g->corr_K, h->params.use_sqloss, A, nb, M));
    // The fallback read-back lands in an area of its OWN: the host reads h_bstates[p] as soon as the polled word has changed,
    // while this copy may still be in flight behind the kernel -- it must not rewrite what the host is reading (solo_wait
    // moves it over when the poll did not see the word: a launch that gave up, polling switched off).
    HIPCHECK(hipMemcpyAsync(reinterpret_cast<char*>(h->h_solo_flag) + kSoloFallbackOffset, h->d_bstates.p + p, sizeof(sicp::LmState), hipMemcpyDeviceToHost, M));
    return SICP_OK;
  }
  // the argument array in HBM only changes when the set of pairs inside a solve does
  const bool same_set = S.tick_valid && joining.empty() && S.tick_act == act;
  int k = 0;
  for (int p : act) {
    if (same_set) break;
    sicp_context* g = hs[p];
    sicp::BatchArgs& B = S.h_batch[k++];
    std::memset(&B, 0, sizeof B);
    const int nb = sicp::accumulate_blocks(g->corr_n * g->corr_K, g->corr_K);
    if (g->partials.reserve((size_t)nb * 28) != hipSuccess) return SICP_ERR_OUT_OF_MEMORY;
    fill_acc(g, B.a);
    B.a.lm = B.a.lm_step = h->d_bstates.p + p;
    B.a.ein = lm_step_in_launch() ? h->d_ein.p + 2 * (size_t)p : nullptr;
    B.nb = nb;
  }
  if (!same_set) {
    *S.h_bhdr = sicp::BatchHeader{(int)act.size(), S.epoch_host, {0, 0}};
    HIPCHECK(hipMemcpyAsync(S.d_bhdr.p, S.h_bhdr, sizeof(sicp::BatchHeader), hipMemcpyHostToDevice, M));
    HIPCHECK(hipMemcpyAsync(S.d_batch.p, S.h_batch, sizeof(sicp::BatchArgs) * act.size(), hipMemcpyHostToDevice, M));
    S.tick_act = act;
    S.tick_valid = true;
  }
  // [accumulate, lm_step_batch] x len as an explicit graph with fixed grids: the kernels read the
  // number of pairs from the header and their status from the LM states, so the graph is instantiated once per tick set
  // (buffer addresses) and never touched when pairs come and go or batches differ in size.
  // SICP_NO_GRAPH: the same 2 x len kernels as plain launches.  For profilers: rocprofv3 (ROCm 7.2) dies with a segmentation
  // fault when a thread other than the main one launches graphs while it traces kernels (tools/r04/two_thread_dispatch.hip
  // reproduces it without this library), and a stream's ticks are launched by its worker thread.  Same kernels, same bits.
  static const bool no_graph = std::getenv("SICP_NO_GRAPH") != nullptr;
  const bool fold = lm_step_in_launch();
  if (no_graph) {
    const int cap = std::min(S.cap, kMaxActivePairs);
    if (fold) HIPCHECK(sicp::launch_tick_prepare(S.d_bhdr.p, S.d_batch.p, M));
    for (int b = 0; b < len; ++b) {
      HIPCHECK(sicp::launch_accumulate_batch(hs[act[0]]->corr_K, h->params.use_sqloss, S.d_bhdr.p, S.d_batch.p, cap, M, b | (S.static_ranges ? sicp::kAccStaticRanges : 0)));
      if (!fold) HIPCHECK(sicp::launch_lm_step_batch(S.d_bhdr.p, S.d_batch.p, cap, M));
    }
  } else {
    int built = 0;
    sicp::BatchGraph& graph = S.graph[S.static_ranges ? 1 : 0];
    HIPCHECK(sicp::batch_graph_prepare(graph, hs[act[0]]->corr_K, h->params.use_sqloss, S.d_bhdr.p, S.d_batch.p, std::min(S.cap, kMaxActivePairs),
                                       len, &built, fold ? 1 : 0, S.static_ranges ? 1 : 0));
    h->st.graph_builds += built;
    HIPCHECK(hipGraphLaunch(graph.exec, M));
  }
  if (fold) S.epoch_host += (unsigned)sicp::kMaxBatchLen;  // (what tick_prepare_kernel has just been queued to do)
  HIPCHECK(hipMemcpyAsync(h->h_bstates + lo, h->d_bstates.p + lo, sizeof(sicp::LmState) * (hi - lo), hipMemcpyDeviceToHost, M));
  return SICP_OK;
}

// Wait for the work queued on M so far.  (Polling with hipStreamQuery before blocking, to shorten the
// wake-up of the host thread, measured no different: 3.76 vs 3.78 ms for one pair alone.)
int tick_wait(sicp_context* h, hipStream_t M) {
  HIPCHECK(hipStreamSynchronize(M));
  return SICP_OK;
}

// Wait for a persistent solve: its master writes the state into the pinned mirror and then the launch number into a pinned
// word (SoloArgs::host_flag) -- the host polls that word (bounded) instead of waiting for the read-back copy queued behind
// the kernel; whatever the poll does not see (a launch that gave up, polling switched off) the stream wait catches.
int solo_wait(sicp_context* h, hipStream_t M) {
  if (h->h_solo_flag) {
    volatile int* flag = h->h_solo_flag;
    for (int spins = 0; *flag != h->solo_seq; ++spins) {
      // every ~20 us: has everything queued on M ended without the word changing?  Then the launch gave up (or the
      // word is not coming): stop polling.
      if ((spins & 1023) == 1023 && hipStreamQuery(M) != hipErrorNotReady) break;
      // a solve of a few hundred evaluations lasts milliseconds: after ~0.2 ms of pure spinning the poll gives its core
      // to whoever else wants it between looks (a stream's worker shares the host with the submitting threads)
      if (spins > 16384 && (spins & 255) == 255) std::this_thread::yield();
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (*flag == h->solo_seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return SICP_OK;  // (h_bstates[solo_pair] holds the LmCore the master wrote; the options behind it never change)
    }
  }
  HIPCHECK(hipStreamSynchronize(M));
  if (h->h_solo_flag && h->solo_pair >= 0)
    std::memcpy(static_cast<void*>(h->h_bstates + h->solo_pair), reinterpret_cast<const char*>(h->h_solo_flag) + kSoloFallbackOffset, sizeof(sicp::LmState));
  return SICP_OK;
}

// after a persistent launch has been waited for: did it run to its regular end?
int solo_check(sicp_context* h) {
#if defined(SICP_SOLO_TIMING)  // developer aid: cycles per phase of the master and of worker 0, per evaluation of the launch that just ended
  {
    unsigned w[14];
    if (debug_enabled() && hipMemcpy(w, h->d_solo_sync.p + sicp::kSoloSyncWords, sizeof w, hipMemcpyDeviceToHost) == hipSuccess) {
      const int ev = std::max(1, h->h_bstates[h->solo_pair].evaluations);
      const char* nm[7] = {"master: wait", "reduce", "lm_feed", "publish", "| worker 0: compute", "publish", "wait"};
      std::fprintf(stderr, "[solo timing] %d evaluations, cycles per evaluation:", ev);
      for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %s %.0f", nm[i], (double)(((unsigned long long)w[2 * i + 1] << 32) | w[2 * i]) / ev);
      std::fprintf(stderr, "\n");
    }
  }
#endif
  // The master echoes the launch's sequence number in the state's pad_ word when it writes the state back.  Anything
  // else means a wait timed out -- the grid was not resident as a whole: something else holds CUs for longer than
  // the limit -- and the launch has left the state in HBM as it was: the solve continues (or starts) as
  // [accumulate, LM step] ticks, and this handle stays with them.
  h->solo_failed = h->h_bstates[h->solo_pair].pad_ != h->solo_seq;
  static const bool log = debug_enabled() && std::getenv("SICP_SOLO_LOG") != nullptr;  // developer aid (SICP_DEBUG gate: engine.hpp)
  if (log)
    std::fprintf(stderr, "[solo] launch %d pair %d init %d -> %s, evaluations %d, status %d, t %.3f ms\n", h->solo_seq, h->solo_pair, (int)h->solo_was_init,
                 h->solo_failed ? "TIMED OUT" : "ok", h->h_bstates[h->solo_pair].evaluations, h->h_bstates[h->solo_pair].status, now_ms());
  if (h->solo_failed) {
    h->solo_penalty = std::min(std::max(2 * h->solo_penalty, 8), 4096);
    h->solo_skip = h->solo_penalty;
    HIPCHECK(hipMemsetAsync(h->d_solo_sync.p, 0, sizeof(unsigned) * sicp::kSoloSyncWords, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->solo_tag = 0;
  }
  return SICP_OK;
}

int run_tick(sicp_context* h, hipStream_t M, sicp_handle* hs, int n, const std::vector<int>& act, const std::vector<int>& joining,
             const double (*start)[7], int len, int solo_evals) {
  SICPCHECK(tick_launch(h, h->ts[0], M, hs, 0, n, act, joining, start, len, solo_evals));
  if (solo_evals > 0) SICPCHECK(solo_wait(h, M)); else SICPCHECK(tick_wait(h, M));
  if (solo_evals > 0) SICPCHECK(solo_check(h));
  return SICP_OK;
}

// What the host does between two ticks of group G: finish the previous tick (if any), queue the searches
// of the pairs that are between two solves, and launch the next tick.
int BatchRun::turn(TickGroup& G, JobCollector& jc) {
  sicp_context* h = L;
  if (G.pending) {
    const double tw0 = now_ms();
    if (solo_now) SICPCHECK(solo_wait(h, G.M)); else SICPCHECK(tick_wait(h, G.M));
    dbg_wait_ms += now_ms() - tw0; ++dbg_ticks; dbg_act += (long long)G.act.size();
    G.pending = false;
    G.finished.clear();
    if (solo_now) {
      SICPCHECK(solo_check(h));
      if (h->solo_failed) {  // nothing has happened: the pair takes the same step again with the ticks
        const int p = G.act[0];
        if (h->solo_was_init) { phase[p] = PAIR_JOINING; search_round[p] = 0; }
        G.act.clear();
      }
    }
    for (int p : G.act) {
      sicp_context* g = hs[p];
      const sicp::LmState& st = h->h_bstates[p];
      // evaluation launches the pair sat through: the tick's, or -- one persistent launch -- its own evaluations
      g->st.lockstep_slots += solo_now ? st.evaluations - evals_seen[p] : len;
      g->st.acc_launches += solo_now ? 1 : len;
      evals_seen[p] = st.status == sicp::LM_RUNNING ? st.evaluations : 0;
      if (st.status == sicp::LM_RUNNING) continue;
      std::memcpy(o[p].est, st.x, sizeof st.x);
      g->st.total_lm_iters += st.iterations;
      g->st.final_cost = st.cost;
      g->st.total_evals += st.evaluations;
      G.finished.push_back(p);
    }
    for (int p : G.finished) {
      jc.slice = 0;
      outer_finish(P, o[p]);
      phase[p] = o[p].converged ? PAIR_DONE : PAIR_NEED_SEARCH;
    }
  }
  if (live(G) == 0) return SICP_OK;
  if (!chunk_ev.empty()) {
    // nothing in flight and nobody ready: wait for the next start-up chunk instead of spinning
    bool idle = !G.pending;
    for (int p = G.lo; p < G.hi && idle; ++p) idle = phase[p] != PAIR_NEED_SEARCH && phase[p] != PAIR_JOINING && phase[p] != PAIR_SOLVING;
    if (promote_started(G, idle) < 0) { h->last_error = "start-up pipeline: event query failed"; return SICP_ERR_HIP; }
  }
  ++G.round;
  if (G.force_wait) {  // (streams.cpp: a feature rewrite of a cloud that a solving pair reads was queued on the side stream)
    if (G.side_recorded) HIPCHECK(hipStreamWaitEvent(G.M, G.side_done, 0));
    G.force_wait = false;
  }
  const double dbg_t_search0 = now_ms();
  // (1) searches of the pairs between two inner solves -> side stream
  bool any_search = false;
  for (int p = G.lo; p < G.hi; ++p) {
    if (phase[p] != PAIR_NEED_SEARCH) continue;
    std::memcpy(o[p].est, o[p].cur, sizeof o[p].est);
    if (P.mode == SICP_MODE_SEMANTIC) o[p].count++;
    jc.slice = batch_slice(p - G.lo, G.hi - G.lo, P.knn);
    SICPCHECK(run_correspondences(hs[p], o[p].cur, P.knn, true));
    phase[p] = PAIR_JOINING;
    search_round[p] = G.round;
    any_search = true;
  }
  // (2) the tick: pairs inside a solve, plus (up to the capacity) the pairs whose search was queued
  // during the previous tick.  When nobody is inside a solve there is nothing for the fresh
  // searches to run beside: they are queued first and their pairs join at once.
  G.act.clear(); G.joining.clear();
  for (int p = G.lo; p < G.hi; ++p)
    if (phase[p] == PAIR_SOLVING) G.act.push_back(p);
  const bool join_fresh = G.act.empty() || !one_launch;
  if (any_search && one_launch && join_fresh) {
    SICPCHECK(flush_jobs(h, jc, side));
    HIPCHECK(hipEventRecord(G.side_done, side));
    G.side_recorded = true;
    any_search = false;
  }
  bool waited = false;
  for (int p = G.lo; p < G.hi && (int)G.act.size() < kMaxActivePairs; ++p) {
    if (phase[p] != PAIR_JOINING || (search_round[p] == G.round && !join_fresh)) continue;
    G.joining.push_back(p); G.act.push_back(p);
    if (!waited && one_launch && G.side_recorded) { HIPCHECK(hipStreamWaitEvent(G.M, G.side_done, 0)); waited = true; }
    if (!one_launch && hs[p]->stream != G.M) {  // the pair's own stream produced its correspondences
      HIPCHECK(hipEventRecord(hs[p]->ev_join, hs[p]->stream));
      HIPCHECK(hipStreamWaitEvent(G.M, hs[p]->ev_join, 0));
    }
  }
  // this round's searches run beside the tick (SICP_TICK_FIRST, tuning aid: they are queued AFTER the tick's launch)
  static const bool tick_first = std::getenv("SICP_TICK_FIRST") != nullptr;
  auto flush_round = [&]() -> int {
    if (any_search && one_launch) {
      SICPCHECK(flush_jobs(h, jc, side));
      HIPCHECK(hipEventRecord(G.side_done, side));
      G.side_recorded = true;
      any_search = false;
    }
    return SICP_OK;
  };
  if (!tick_first || G.act.empty()) SICPCHECK(flush_round());
  if (G.act.empty()) return SICP_OK;
  // the tick reads its pairs' arguments in ascending slot order (the order of the argument array)
  std::sort(G.act.begin(), G.act.end());
  for (int p : G.joining) { phase[p] = PAIR_SOLVING; std::memcpy(starts[p].q, o[p].est, sizeof starts[p].q); }
  // The only pair of the whole run that still iterates -- a run of one, or the tail of a batch -- has the chip to
  // itself: its solve continues as persistent launches (of at most 64 evaluations when other slots may fill up
  // meanwhile: a stream's new registrations are admitted between launches).
  solo_now = false;
  if (solo && G.act.size() == 1) {
    int live_all = 0;
    for (int p = 0; p < n; ++p) live_all += phase[p] != PAIR_FREE && phase[p] != PAIR_DONE;
    const sicp_context* g = hs[G.act[0]];
    solo_now = live_all == 1 && sicp::solve_one_fits(g->corr_n * g->corr_K, g->corr_K) && solo_allowed(L);
  }
  const double dbg_t_launch0 = now_ms();
  dbg_search_ms += dbg_t_launch0 - dbg_t_search0;
  G.S->static_ranges = acc_static;
  int rc = tick_launch(h, *G.S, G.M, hs, G.lo, G.hi, G.act, G.joining, reinterpret_cast<const double(*)[7]>(starts.data()), len,
                       solo_now ? (n == 1 ? sicp::kSoloMaxEvals : 64) : 0);
  dbg_launch_ms += now_ms() - dbg_t_launch0;
  if (rc != SICP_OK) return rc;
  SICPCHECK(flush_round());
  G.pending = true;
  return SICP_OK;
}

// sicp_align_batch (and sicp_align with the device-resident solve: a batch of one)
int align_batch(sicp_handle* hs, int32_t n, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats) {
  if (!hs || n < 1 || !init_qt || !out_qt) return SICP_ERR_INVALID_ARGUMENT;
  for (int p = 0; p < n; ++p)
    if (!hs[p]) return SICP_ERR_INVALID_ARGUMENT;
  sicp_context* L = hs[0];  // the leader owns the batch buffers and runs the batched kernels on its stream
  {
    sicp_context* h = L;
    SICPCHECK(set_device(h));
  }
  for (int p = 0; p < n; ++p) {
    sicp_context* h = hs[p];
    for (int q = 0; q < p; ++q)
      if (hs[q] == h) return SICP_ERR_INVALID_ARGUMENT;  // every pair needs its own handle
    // one launch evaluates every pair: they must agree on what a launch does
    if (h->device != L->device || !same_solver(h->params, L->params)) {
      h->last_error = "sicp_align_batch: handles differ in device, mode, knn, k_cov, nn_method, lm_on_device, lm_batch, profile or loss";
      return SICP_ERR_INVALID_ARGUMENT;
    }
    SICPCHECK(check_ready(h, false));
    if (general_covariances(h)) {
      if (n == 1) return align_host_loop(h, init_qt, out_qt, outer_iters, stats);
      h->last_error = "sicp_align_batch: a handle with caller covariances of general form (not I - (1 - epsilon) n n^T) registers one pair at a time (sicp_align)";
      return SICP_ERR_INVALID_ARGUMENT;
    }
  }
  const sicp_params& P = L->params;
  const double t_begin = now_ms();
  // One launch per kind of kernel for ALL pairs (searches, covariances, projections, weights):
  // their long tails overlap inside the launch.  (With profiling on, or another search engine
  // selected, every pair launches its own kernels on its own stream instead.)
  bool one_launch = true;
  for (int p = 0; p < n; ++p) {
    one_launch = one_launch && hs[p]->params.profile == 0 && hs[p]->params.nn_method == 1;
    sicp_context* h = hs[p];
    HIPCHECK(hipStreamSynchronize(h->stream));  // earlier work of the handle on its own stream
  }
  JobCollector jc;
  jc.fold_weights = n <= 4;
  BatchGuard guard(hs, n, one_launch ? &jc : nullptr, L->stream);
  const unsigned long long epoch = next_epoch();
  // A large batch starts PIPELINED: the per-align features (self-searches, covariances, projections) and the first
  // search + weights of the pairs are queued chunk by chunk on a stream of their own, and a chunk's pairs join the
  // ticks as soon as its event has completed -- the solves of the first pairs run beside the features of the later
  // ones instead of ~50 ms of features for all 512 clouds before the first tick.  The GPU is work-bound, so this only
  // fills the ramp: 2.09 -> 2.12 G corr/s at 256 pairs (chunks of 8 ... 32 alike, 64 and more lose it again).
  const bool staged = one_launch && n > 48;
  const bool early_first = one_launch && n <= 4;  // (never staged: that starts at 49 pairs)
  static const int kStartChunk = [] { const char* e = std::getenv("SICP_START_CHUNK"); const int v = e ? std::atoi(e) : 32; return v > 0 ? v : 32; }();  // tuning aid
  if (!staged) {  // (a staged batch queues its start-up pipelines below, once the run exists)
    for (int p = 0; p < n; ++p) {
      hs[p]->epoch = epoch;
      jc.slice = batch_slice(p, n, P.knn);
      SICPCHECK(align_begin(hs[p], stats != nullptr));
    }
    if (one_launch) {
      if (early_first) {
        // A few pairs alone: the first search (pose = the initial guess) needs none of the features, only the weights
        // behind it do -- it is collected into a slice of its own, i.e. runs on its own stream beside the
        // self-searches / covariances / projections; the weights follow once both have been queued.
        for (int p = 0; p < n; ++p) {
          jc.slice = kParts - 1;
          SICPCHECK(run_correspondences(hs[p], init_qt + 7 * p, P.knn, false));
        }
      }
      SICPCHECK(flush_jobs(L, jc));
      if (early_first) {
        for (int p = 0; p < n; ++p) {
          jc.slice = 0;
          SICPCHECK(run_weights(hs[p], init_qt + 7 * p));
        }
        SICPCHECK(flush_jobs(L, jc));
      }
    } else {
      // per-pair launches on the pairs' own streams: a cloud shared by two pairs has just been given
      // its features on ONE of them
      for (int p = 0; p < n; ++p) {
        sicp_context* h = hs[p];
        HIPCHECK(hipStreamSynchronize(h->stream));
      }
    }
  }
  // ---- the outer loops of all pairs, CONTINUOUSLY batched.  Every pair runs its own sequence
  //   search (transform + kNN + weights) -> inner solve -> convergence test -> search -> ...
  // and the batch advances in ticks of lm_batch LM evaluations: one graph launch evaluates every pair
  // that is inside an inner solve, while the searches of the pairs that have just finished one run
  // on a second stream beside it; those pairs rejoin at the next tick.  No pair waits for another
  // pair's solve or outer loop -- only for the end of the current tick.
  {
    sicp_context* h = L;
    SICPCHECK(batch_reserve(h, n));
    h->ts[0].tick_valid = false;
    if (!h->side_stream) {
      HIPCHECK(create_side_stream(&h->side_stream));
      HIPCHECK(hipEventCreateWithFlags(&h->side_done, hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&h->main_done, hipEventDisableTiming));
    }
  }
  BatchRun run;
  run.L = L; run.hs = hs; run.P = P; run.one_launch = one_launch; run.want_stats = stats != nullptr;
  run.solo = one_launch && P.lm_on_device != 2;
  run.resize(n);
  for (int p = 0; p < n; ++p) run.start_pair(p, init_qt + 7 * p);
  if (early_first)  // (their first search is on its way: what BatchRun::turn does for a pair between two solves)
    for (int p = 0; p < n; ++p) {
      OuterState& o = run.o[p];
      std::memcpy(o.est, o.cur, sizeof o.est);
      if (P.mode == SICP_MODE_SEMANTIC) o.count++;
      run.phase[p] = PAIR_JOINING;
      run.search_round[p] = 0;
    }
  if (staged) {
    sicp_context* h = L;
    if (!h->feat_stream) HIPCHECK(hipStreamCreateWithFlags(&h->feat_stream, hipStreamNonBlocking));
    const int n_chunks = (n + kStartChunk - 1) / kStartChunk;
    while ((int)h->chunk_ev.size() < n_chunks) {
      hipEvent_t e = nullptr;
      HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->chunk_ev.push_back(e);
    }
    run.chunk_ev.assign(h->chunk_ev.begin(), h->chunk_ev.begin() + n_chunks);
    guard.retarget(h->feat_stream);
    for (int c = 0; c < n_chunks; ++c) {
      const int p0 = c * kStartChunk, p1 = std::min(n, p0 + kStartChunk);
      for (int p = p0; p < p1; ++p) {
        hs[p]->epoch = epoch;
        jc.slice = batch_slice(p - p0, p1 - p0, P.knn);
        SICPCHECK(align_begin(hs[p], stats != nullptr));
      }
      SICPCHECK(flush_jobs(h, jc, h->feat_stream));
      for (int p = p0; p < p1; ++p) {  // the first search of the pair (what BatchRun::turn does for a pair between two solves)
        OuterState& o = run.o[p];
        std::memcpy(o.est, o.cur, sizeof o.est);
        if (P.mode == SICP_MODE_SEMANTIC) o.count++;
        jc.slice = batch_slice(p - p0, p1 - p0, P.knn);
        SICPCHECK(run_correspondences(hs[p], o.cur, P.knn, true));
        run.phase[p] = PAIR_FIRST;
        run.first_chunk[p] = c;
      }
      SICPCHECK(flush_jobs(h, jc, h->feat_stream));
      HIPCHECK(hipEventRecord(run.chunk_ev[c], h->feat_stream));
    }
  }
  // Tick length: lm_batch evaluations; twice that for up to 4 pairs, where the host's turn-around between
  // ticks (not the idle tail of a tick: a launch over finished pairs costs ~7 us) is what hurts.
  run.len = std::min((P.lm_batch > 0 ? P.lm_batch : 12) * (n <= 4 ? 2 : 1), sicp::kMaxBatchLen);
  // Two halves of the batch alternate (from 8 pairs on): while the host reads back one half's tick,
  // tests its pairs for convergence and queues their searches, the other half's tick is running, so the
  // GPU does not idle through the host's turn-around (~60 us per tick).  Each half has its own stream
  // and argument set; the LM states are indexed by pair.
  // (Measured, 100K-point EM pairs: +1...4 % at 32 pairs; at 64 pairs one tick over all pairs is 4 %
  // faster again -- its launches are long enough to hide the turn-around, and half-size launches pay
  // the fixed ~20 us of a launch boundary + LM step twice as often.)
  const int n_groups = (one_launch && n >= 8 && n <= 48) ? 2 : 1;
  TickGroup grp[2];
  {
    sicp_context* h = L;
    for (int g = 0; g < n_groups; ++g) {
      grp[g].lo = g == 0 ? 0 : n / 2;
      grp[g].hi = (g == n_groups - 1) ? n : n / 2;
      grp[g].M = g == 0 ? guard.s1[0] : guard.s2[0];  // the leader's own two streams
      grp[g].S = &h->ts[g];
      SICPCHECK(tickset_reserve(h, h->ts[g], grp[g].hi - grp[g].lo));
      h->ts[g].tick_valid = false;
      if (g == 1 && !h->side_done2) HIPCHECK(hipEventCreateWithFlags(&h->side_done2, hipEventDisableTiming));
      grp[g].side_done = g == 0 ? h->side_done : h->side_done2;
    }
  }
  // (one pair alone: its searches and its solves alternate anyway -- one stream, no cross-stream event per outer iteration)
  const hipStream_t side = (one_launch && n > 1) ? L->side_stream : grp[0].M;
  run.side = side;
  {  // the tick streams and the side stream start after everything queued so far (features)
    sicp_context* h = L;
    HIPCHECK(hipEventRecord(h->main_done, guard.s1[0]));
    HIPCHECK(hipStreamWaitEvent(side, h->main_done, 0));
    if (n_groups == 2) HIPCHECK(hipStreamWaitEvent(grp[1].M, h->main_done, 0));
  }
  JobCollector gjc[2];
  gjc[0].fold_weights = gjc[1].fold_weights = jc.fold_weights;
  if (one_launch) {
    guard.retarget(side);
    for (int g = 0; g < n_groups; ++g)  // from here on a pair's stages collect into its group's job lists
      for (int p = grp[g].lo; p < grp[g].hi; ++p) hs[p]->collect = &gjc[g];
  }
  for (;;) {
    bool all_done = true;
    for (int g = 0; g < n_groups; ++g) {
      int rc = run.turn(grp[g], gjc[g]);
      if (rc != SICP_OK) return rc;
      all_done = all_done && run.live(grp[g]) == 0 && !grp[g].pending;
    }
    if (all_done) break;
  }
  {
    sicp_context* h = L;
    if (one_launch) HIPCHECK(hipStreamSynchronize(side));
  }
  for (int p = 0; p < n; ++p) {
    std::memcpy(out_qt + 7 * p, run.o[p].cur, sizeof run.o[p].cur);
    SICPCHECK(align_end(hs[p], run.o[p], t_begin, outer_iters ? outer_iters + p : nullptr, stats ? stats + p : nullptr));
  }
  return SICP_OK;
}

}  // namespace host
}  // namespace sicp
