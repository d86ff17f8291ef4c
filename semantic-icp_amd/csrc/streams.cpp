// streams.cpp -- registration streams (sicp_stream_*): the continuous batching of solve.cpp without the closed batch.
#include "engine.hpp"

namespace sicp {
namespace host {

void stream_fail(sicp_stream_ctx* S, int rc, const std::string& msg) {
  std::lock_guard<std::mutex> lock(S->m);
  if (S->error == SICP_OK) { S->error = rc; S->error_msg = msg; }
  S->cv_done.notify_all();
  S->cv_space.notify_all();
}

// What compute_features / ensure_proj record about a cloud's feature buffers at the moment they queue the kernels.
struct FeatureMarks {
  bool feat_valid = false, proj_valid = false, feat_hist = false;
  int rec_dense_n = 0, feat_k = 0, feat_C = 0, feat_float_products = 0, nn_stride = 0;
  unsigned long long feat_epoch = 0, proj_cm_id = 0;
  explicit FeatureMarks(const Cloud* c) {
    if (!c) return;
    feat_valid = c->feat_valid; proj_valid = c->proj_valid; feat_hist = c->feat_hist; rec_dense_n = c->rec_dense_n;
    feat_k = c->feat_k; feat_C = c->feat_C; feat_float_products = c->feat_float_products; nn_stride = c->nn_stride;
    feat_epoch = c->feat_epoch; proj_cm_id = c->proj_cm_id;
  }
  void restore(Cloud* c) const {
    if (!c) return;
    c->feat_valid = feat_valid; c->proj_valid = proj_valid; c->feat_hist = feat_hist; c->rec_dense_n = rec_dense_n;
    c->feat_k = feat_k; c->feat_C = feat_C; c->feat_float_products = feat_float_products; c->nn_stride = nn_stride;
    c->feat_epoch = feat_epoch; c->proj_cm_id = proj_cm_id;
  }
};

// The worker: admit queued registrations into free slots, one turn of the continuous batching
// (BatchRun::turn: finish the tick in flight, queue the searches of the pairs between two solves, launch
// the next tick), retire the pairs that have converged.  One iteration per tick.
static void stream_worker_loop(sicp_stream_ctx* S) {
  if (hipSetDevice(S->device) != hipSuccess) { stream_fail(S, SICP_ERR_NO_DEVICE, "hipSetDevice"); return; }
  sicp_context* L = S->slots[0];
  BatchRun run;
  run.L = L; run.hs = S->slots.data(); run.P = S->params; run.one_launch = true; run.want_stats = false;
  run.resize(S->cap);
  run.len = std::min(S->params.lm_batch > 0 ? S->params.lm_batch : 8, sicp::kMaxBatchLen);
  run.side = L->side_stream;
  TickGroup G;
  G.lo = 0; G.hi = S->cap; G.M = S->own1[0]; G.S = &L->ts[0]; G.side_done = L->side_done;
  JobCollector jc;
  for (sicp_context* g : S->slots) g->collect = &jc;
  std::vector<int> free_slots;
  for (int p = S->cap - 1; p >= 0; --p) free_slots.push_back(p);
  std::vector<sicp_stream_ctx::Submission> fresh;
  std::vector<int> fresh_slot;
  // Registrations in flight per cloud.  A SICP_SUBMIT_FRESH_FEATURES registration REWRITES the feature buffers of its
  // two clouds (records, dense records, neighbour lists, histograms, projections) on the side stream, while the tick
  // of the pairs already solving -- stream M -- reads those very buffers when another live registration shares a
  // cloud.  Nothing used to order the two; now:
  //   * such a registration first waits in the queue for the live one to retire (others overtake it: free), and after
  //     kMaxOvertaken worker turns it is admitted WITH ORDER: the side stream waits for the tick in flight (an event on
  //     M), and the next tick waits for the side stream's rewrite (TickGroup::force_wait) -- one tick without the
  //     overlap, so a pair that is re-submitted faster than it converges does not become a serial chain;
  //   * two registrations of ONE admission round never rewrite the same cloud (their feature jobs would land in the
  //     same launch): the later one waits a round, later registrations that touch its clouds stay behind it, others
  //     overtake it (results carry tickets, not positions).
  // A registration that reuses features only reads: its reads follow any rewrite queued earlier on the side stream.
  std::unordered_map<const Cloud*, int> users;
  constexpr int kMaxOvertaken = 48;  // worker turns (a turn = one tick of ~4 LM evaluations)
  std::unordered_set<const Cloud*> held, round_clouds;  // (hash sets: the scan below runs under S->m once per worker turn)
  bool round_rewrites = false, order_needed = false;
  hipEvent_t tick_ev = nullptr;  // recorded on M behind the tick in flight when a rewrite has to wait for it
  auto release_users = [&](const Cloud* a, const Cloud* b) {
    for (const Cloud* c : {a, b}) {
      auto it = users.find(c);
      if (it != users.end() && --it->second <= 0) users.erase(it);
    }
  };
  std::vector<std::pair<const Cloud*, const Cloud*>> slot_clouds((size_t)S->cap, {nullptr, nullptr});
  std::vector<sicp_stream_result> out;
  std::vector<std::array<double, 11>> dbg_log;
  double dbg_admit_ms = 0, dbg_flush_ms = 0, dbg_turn_ms = 0;
  for (;;) {
    // ---- admit
    fresh.clear(); fresh_slot.clear();
    round_rewrites = false; order_needed = false;
    {
      std::unique_lock<std::mutex> lock(S->m);
      S->cv_work.wait(lock, [&] { return S->stop || !S->queue.empty() || S->in_flight > 0; });
      if (S->stop) {
        if (tick_ev) (void)hipEventDestroy(tick_ev);
        for (size_t i = 0; i < dbg_log.size(); i += std::max<size_t>(1, dbg_log.size() / 40))
          std::fprintf(stderr, "[stream] t %.1f ms completed %.0f ticks %.0f waited %.1f ms pairs-per-tick %.1f solo-allowed %.0f | host ms: admit %.1f flush %.1f turn %.1f (of which waited; searches %.1f, tick launch %.1f)\n",
                       dbg_log[i][0], dbg_log[i][1], dbg_log[i][2], dbg_log[i][3], dbg_log[i][4], dbg_log[i][5], dbg_log[i][6], dbg_log[i][7], dbg_log[i][8], dbg_log[i][9], dbg_log[i][10]);
        return;
      }
      held.clear();
      round_clouds.clear();
      for (auto it = S->queue.begin(); it != S->queue.end() && !free_slots.empty();) {
        const Cloud* a = it->src.get();
        const Cloud* b = it->tgt.get();
        const bool behind = held.count(a) || held.count(b);
        const bool rewrites = (it->flags & SICP_SUBMIT_FRESH_FEATURES) != 0;
        const bool in_round = round_clouds.count(a) || round_clouds.count(b);
        const bool shared_live = rewrites && (users.count(a) || users.count(b)) && !in_round;  // with a registration admitted earlier
        // a rewrite of a cloud a live registration reads first WAITS for that registration (free: others go ahead); only
        // when it has waited kMaxOvertaken rounds is it admitted with order (one tick without overlap) -- a pair that is
        // re-submitted faster than it converges must not become a serial chain
        const bool wait_for_live = shared_live && it->overtaken < kMaxOvertaken;
        if (behind || wait_for_live || (in_round && (rewrites || round_rewrites))) {
          if (wait_for_live) ++it->overtaken;
          held.insert(a); held.insert(b);
          ++it;
          continue;
        }
        if (shared_live) order_needed = true;
        if (rewrites) round_rewrites = true;
        round_clouds.insert(a); round_clouds.insert(b);
        ++users[a]; ++users[b];
        slot_clouds[(size_t)free_slots.back()] = {a, b};
        fresh.push_back(std::move(*it));
        it = S->queue.erase(it);
        fresh_slot.push_back(free_slots.back());
        free_slots.pop_back();
        ++S->in_flight;
      }
      if (!fresh.empty()) S->cv_space.notify_all();
      // The last registration of a stream that is being drained may run its solves as persistent launches.  Not
      // otherwise: a stream that has just begun is alone for a moment too, and the next registrations' feature
      // kernels would then compete with the persistent grid for the CUs it needs all at once.
      run.solo = S->params.lm_on_device != 2 && S->draining > 0 && S->queue.empty();
    }
    out.clear();
    const double t_admit0 = now_ms();
    for (size_t k = 0; k < fresh.size(); ++k) {
      const int p = fresh_slot[k];
      sicp_context* h = S->slots[p];
      // the slot lets go of its previous pair's clouds and takes this pair's
      h->cl[0] = fresh[k].src;
      h->cl[1] = fresh[k].tgt;
      h->corr_valid = false;
      h->epoch = next_epoch();
      S->slot_ticket[p] = fresh[k].ticket;
      S->slot_t0[p] = now_ms();
      S->slot_flags[p] = fresh[k].flags;
      jc.slice = batch_slice(p, S->cap, S->params.knn);
      size_t mark[5][kParts];  // what the collector held before this registration queued anything
      for (int q = 0; q < kParts; ++q) {
        mark[0][q] = jc.knn[q].size(); mark[1][q] = jc.cov[q].size(); mark[2][q] = jc.proj[q].size();
        mark[3][q] = jc.weight[q].size(); mark[4][q] = jc.count[q].size();
      }
      FeatureMarks before[2] = {FeatureMarks(h->cl[0].get()), FeatureMarks(h->cl[1].get())};
      int rc = check_ready(h, false);
      if (rc == SICP_OK && general_covariances(h)) rc = SICP_ERR_INVALID_ARGUMENT;  // (caller covariances of general form: sicp_align, one pair at a time)
      // SICP_SUBMIT_FRESH_FEATURES: this registration recomputes the features of both its clouds, like an align() of
      // the reference (the slot's epoch is new, so align_begin finds them stale)
      h->params.reuse_features = (fresh[k].flags & SICP_SUBMIT_FRESH_FEATURES) ? 0 : 1;
      if (rc == SICP_OK) rc = align_begin(h, false);
      h->params.reuse_features = 1;
      if (rc != SICP_OK) {  // this registration cannot run (too few points, bad labels ...): report it, free the slot
        sicp_stream_result r;
        std::memset(&r, 0, sizeof r);
        r.ticket = fresh[k].ticket; r.status = rc;
        std::memcpy(r.qt, fresh[k].init, sizeof r.qt);
        out.push_back(r);
        free_slots.push_back(p);
        if (rc == SICP_ERR_HIP) { stream_fail(S, rc, h->last_error); return; }
        // nothing of it runs: the jobs it had queued are dropped, its clouds go back (a released cloud's memory must not
        // stay pinned by an idle slot), and the registrations waiting for those clouds may come in
        for (int q = 0; q < kParts; ++q) {
          jc.knn[q].resize(mark[0][q]); jc.cov[q].resize(mark[1][q]); jc.proj[q].resize(mark[2][q]);
          jc.weight[q].resize(mark[3][q]); jc.count[q].resize(mark[4][q]);
        }
        // ... and with the jobs goes what they would have written: compute_features / ensure_proj mark a cloud's records,
        // histograms and projections current when they QUEUE the kernels, so a registration that failed on its second cloud
        // (an arena that is full: SICP_ERR_OUT_OF_MEMORY) would leave its first cloud marked current with nothing computed,
        // and every later registration sharing it would solve on stale or uninitialised features.  The clouds get back the
        // marks they had before this registration queued anything (what an earlier registration of this round queued for
        // them is still in the collector and still runs).
        for (int c = 0; c < 2; ++c) before[c].restore(h->cl[c].get());
        release_users(slot_clouds[(size_t)p].first, slot_clouds[(size_t)p].second);
        slot_clouds[(size_t)p] = {nullptr, nullptr};
        h->cl[0] = acquire_cloud(S->device);
        h->cl[1] = acquire_cloud(S->device);
        continue;
      }
      run.start_pair(p, fresh[k].init);
    }
    // the new pairs' features (self-searches, covariances, projections): one launch per kind, on the side
    // stream, beside the tick in flight and ahead of the pairs' first searches
    const double t_flush0 = now_ms();
    dbg_admit_ms += t_flush0 - t_admit0;
    if (!fresh.empty()) {
      if (order_needed) {  // the rewrite waits for the tick that may be reading the shared cloud ...
        hipError_t e = tick_ev ? hipSuccess : hipEventCreateWithFlags(&tick_ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(tick_ev, G.M);
        if (e == hipSuccess) e = hipStreamWaitEvent(run.side, tick_ev, 0);
        if (e != hipSuccess) { stream_fail(S, SICP_ERR_HIP, std::string("stream: ordering a feature rewrite: ") + hipGetErrorString(e)); return; }
      }
      const int rc = flush_jobs(L, jc, run.side);
      if (rc != SICP_OK) { stream_fail(S, rc, L->last_error); return; }
      if (order_needed) {  // ... and the next tick waits for the rewrite
        const hipError_t e = hipEventRecord(G.side_done, run.side);
        if (e != hipSuccess) { stream_fail(S, SICP_ERR_HIP, std::string("stream: ordering a feature rewrite: ") + hipGetErrorString(e)); return; }
        G.side_recorded = true;
        G.force_wait = true;
      }
    }
    const double t_turn0 = now_ms();
    dbg_flush_ms += t_turn0 - t_flush0;
    // ---- one turn
    {
      // scans still arriving: their tree-build kernels (sorts that need LDS) run beside the ticks, and only get onto a CU when
      // an accumulate workgroup leaves it -- which the dynamic split of a large launch postpones to the launch's very end
      run.acc_static = now_ms() - S->last_add_ms.load(std::memory_order_relaxed) < 5.0;
      const int rc = run.turn(G, jc);
      if (rc != SICP_OK) { stream_fail(S, rc, L->last_error); return; }
    }
    dbg_turn_ms += now_ms() - t_turn0;
    // ---- fused labels of the pairs that have just converged (SICP_SUBMIT_FUSED_LABELS): the K = 4 searches of all of
    //      them in one job launch on the side stream, then per pair the label kernel, the read-back into the slot's
    //      pinned buffer and an event; the slot stays taken (PAIR_LABELS) until that event has completed
    {
      bool any = false;
      for (int p = 0; p < S->cap; ++p) {
        if (run.phase[p] != PAIR_DONE || !(S->slot_flags[p] & SICP_SUBMIT_FUSED_LABELS)) continue;
        sicp_context* h = S->slots[p];
        jc.slice = 0;
        const sicp_stats keep = h->st;  // (the registration's own counters: the label pass is not part of its align())
        const int rc = labels_search(h, run.o[p].cur);
        h->st = keep;
        if (rc != SICP_OK) { stream_fail(S, rc, h->last_error); return; }
        run.phase[p] = PAIR_LABELS;
        S->slot_flags[p] |= 0x80000000u;  // queued in this turn
        any = true;
      }
      if (any) {
        int rc = flush_jobs(L, jc, run.side);
        if (rc != SICP_OK) { stream_fail(S, rc, L->last_error); return; }
        for (int p = 0; p < S->cap; ++p) {
          if (!(S->slot_flags[p] & 0x80000000u)) continue;
          S->slot_flags[p] &= ~0x80000000u;
          sicp_context* h = S->slots[p];
          const int n = h->cloud(0).n;
          rc = labels_launch(h, run.o[p].cur, run.side);
          hipError_t e = rc == SICP_OK ? h->h_labels.resize((size_t)(n > 0 ? n : 1)) : hipSuccess;
          if (rc == SICP_OK && e == hipSuccess && n > 0)
            e = hipMemcpyAsync(h->h_labels.data(), h->tmpl.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, run.side);
          if (rc == SICP_OK && e == hipSuccess && !S->slot_ev[p]) e = hipEventCreateWithFlags(&S->slot_ev[p], hipEventDisableTiming);
          if (rc == SICP_OK && e == hipSuccess) e = hipEventRecord(S->slot_ev[p], run.side);
          if (rc != SICP_OK || e != hipSuccess) { stream_fail(S, rc != SICP_OK ? rc : SICP_ERR_HIP, rc != SICP_OK ? h->last_error : std::string("fused labels: ") + hipGetErrorString(e)); return; }
        }
      }
    }
    // ---- retire
    long long busy = 0, slots_sat = 0;
    int labels_waiting = -1;
    for (int p = 0; p < S->cap; ++p) {
      if (run.phase[p] == PAIR_LABELS) {
        const hipError_t q = hipEventQuery(S->slot_ev[p]);
        if (q == hipErrorNotReady) { if (labels_waiting < 0) labels_waiting = p; continue; }
        if (q != hipSuccess) { stream_fail(S, SICP_ERR_HIP, std::string("fused labels: ") + hipGetErrorString(q)); return; }
        // device order -> the caller's order (a point that never went to the device has no correspondences: label 0)
        sicp_context* h = S->slots[p];
        const Cloud& C0 = h->cloud(0);
        std::vector<uint32_t> lab((size_t)C0.n_caller, 0u);
        for (int d = 0; d < C0.n; ++d) lab[(size_t)C0.caller_index(d)] = h->h_labels[d];
        {
          std::lock_guard<std::mutex> lock(S->m);
          S->labels[S->slot_ticket[p]] = std::move(lab);  // kept until sicp_stream_take_labels(ticket) (or the stream's end)
        }
        run.phase[p] = PAIR_DONE;
        S->slot_flags[p] &= ~(unsigned)SICP_SUBMIT_FUSED_LABELS;
      }
      if (run.phase[p] != PAIR_DONE) continue;
      sicp_context* h = S->slots[p];
      sicp_stream_result r;
      std::memset(&r, 0, sizeof r);
      r.ticket = S->slot_ticket[p];
      r.status = SICP_OK;
      h->st.outer_iters = S->params.mode == SICP_MODE_SEMANTIC ? run.o[p].count : run.o[p].outer;
      h->st.t_total_ms = now_ms() - S->slot_t0[p];
      r.outer_iters = h->st.outer_iters;
      std::memcpy(r.qt, run.o[p].cur, sizeof r.qt);
      r.stats = h->st;
      busy += h->st.total_evals; slots_sat += h->st.lockstep_slots;
      out.push_back(r);
      run.phase[p] = PAIR_FREE;
      free_slots.push_back(p);
      release_users(slot_clouds[(size_t)p].first, slot_clouds[(size_t)p].second);
      slot_clouds[(size_t)p] = {nullptr, nullptr};
    }
    {
      static const bool slog = debug_enabled() && std::getenv("SICP_STREAM_LOG") != nullptr;  // developer aid (needs SICP_DEBUG): kept in memory, printed when the stream ends
      if (slog && !out.empty())
        dbg_log.push_back({now_ms(), (double)(S->completed + (long long)out.size()), (double)run.dbg_ticks, run.dbg_wait_ms,
                           run.dbg_ticks ? (double)run.dbg_act / run.dbg_ticks : 0.0, (double)run.solo, dbg_admit_ms, dbg_flush_ms, dbg_turn_ms, run.dbg_search_ms, run.dbg_launch_ms});
    }
    // nothing left to advance but label read-backs: wait for the first instead of spinning through empty turns
    if (out.empty() && labels_waiting >= 0 && run.live(G) == 0 && !G.pending) (void)hipEventSynchronize(S->slot_ev[labels_waiting]);
    if (!out.empty()) {
      std::lock_guard<std::mutex> lock(S->m);
      for (const sicp_stream_result& r : out) S->done.push_back(r);
      S->in_flight -= (int)out.size();
      S->completed += (long long)out.size();
      S->busy_evals += busy; S->slot_evals += slots_sat;
      S->cv_done.notify_all();
    }
  }
}

// The thread function: an exception that left it would end the process (std::terminate) from inside the library;
// it becomes the stream's fatal error instead -- every caller blocked in submit / poll wakes up with a status.
void stream_worker(sicp_stream_ctx* S) {
  const char* what = nullptr;
  int rc = SICP_ERR_INTERNAL;
  try {
    stream_worker_loop(S);
    return;
  } catch (const std::bad_alloc&) {
    rc = SICP_ERR_OUT_OF_MEMORY; what = "stream worker: out of host memory";
  } catch (const std::exception& e) {
    try { stream_fail(S, SICP_ERR_INTERNAL, std::string("stream worker: ") + e.what()); return; } catch (...) { what = "stream worker: exception"; }
  } catch (...) {
    what = "stream worker: unknown exception";
  }
  try {
    stream_fail(S, rc, what);
  } catch (...) {  // (not even the message could be stored: the status alone)
    std::lock_guard<std::mutex> lock(S->m);
    if (S->error == SICP_OK) S->error = rc;
    S->cv_done.notify_all();
    S->cv_space.notify_all();
  }
}

}  // namespace host
}  // namespace sicp

using namespace sicp::host;

extern "C" {

int sicp_stream_create(int device_id, const sicp_params* params, int32_t max_in_flight, sicp_stream* out) {
  return abi_guard([&]() -> int {
    if (!out || !params || max_in_flight < 1 || max_in_flight > 4096) return SICP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (params->nn_method != 1 || params->lm_on_device == 0 || params->profile != 0) return SICP_ERR_INVALID_ARGUMENT;
    std::unique_ptr<sicp_stream_ctx> S(new (std::nothrow) sicp_stream_ctx());
    if (!S) return SICP_ERR_OUT_OF_MEMORY;
    S->device = device_id;
    S->cap = max_in_flight;
    S->params = *params;
    S->params.reuse_features = 1;  // a stream's cloud keeps its normals / histograms: computed with its first registration
    S->params.lm_on_device = params->lm_on_device == 2 ? 2 : 1;  // (2: never the persistent solve)
    auto cleanup = [&](int rc) {
      for (size_t k = 0; k < S->slots.size(); ++k) {
        sicp_context* g = S->slots[k];
        g->collect = nullptr; g->stream = S->own1[k]; g->stream2 = S->own2[k];
        sicp_destroy(g);
      }
      if (S->uploader) sicp_destroy(S->uploader);
      return rc;
    };
    int rc = sicp_create(device_id, &S->uploader);
    if (rc != SICP_OK) return cleanup(rc);
    rc = sicp_set_params(S->uploader, &S->params);
    if (rc != SICP_OK) return cleanup(rc);
    for (int p = 0; p < S->cap; ++p) {
      sicp_context* g = nullptr;
      rc = sicp_create(device_id, &g);
      if (rc != SICP_OK) return cleanup(rc);
      S->slots.push_back(g);
      S->own1.push_back(g->stream);
      S->own2.push_back(g->stream2);
      rc = sicp_set_params(g, &S->params);
      if (rc != SICP_OK) return cleanup(rc);
    }
    {  // the leader's batch machinery (what sicp_align_batch sets up per call)
      sicp_context* h = S->slots[0];
      rc = batch_reserve(h, S->cap);
      if (rc != SICP_OK) return cleanup(rc);
      h->ts[0].tick_valid = false;
      if (!h->side_stream) {
        if (create_side_stream(&h->side_stream) != hipSuccess ||
            hipEventCreateWithFlags(&h->side_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->main_done, hipEventDisableTiming) != hipSuccess)
          return cleanup(SICP_ERR_HIP);
      }
      // every slot's own launches (memsets of the semantic search, cloud waits) go to the side stream
      for (sicp_context* g : S->slots) { g->stream = h->side_stream; g->stream2 = h->side_stream; g->wait_on_device = true; }
    }
    S->slot_ticket.assign(S->cap, 0);
    S->slot_t0.assign(S->cap, 0.0);
    S->slot_flags.assign(S->cap, 0u);
    S->slot_ev.assign(S->cap, nullptr);
    S->worker = std::thread(stream_worker, S.get());
    *out = S.release();
    return SICP_OK;
  });
}

int sicp_stream_destroy(sicp_stream S) {
  return abi_guard(S, [&]() -> int {
    if (!S) return SICP_OK;
    {
      std::lock_guard<std::mutex> lock(S->m);
      S->stop = true;
      S->cv_work.notify_all();
      S->cv_space.notify_all();
      S->cv_done.notify_all();
    }
    if (S->worker.joinable()) S->worker.join();
    (void)hipSetDevice(S->device);
    if (!S->slots.empty() && S->slots[0]->side_stream) (void)hipStreamSynchronize(S->slots[0]->side_stream);
    for (size_t k = 0; k < S->slots.size(); ++k) (void)hipStreamSynchronize(S->own1[k]);
    if (S->uploader) (void)hipStreamSynchronize(S->uploader->stream);
    // the clouds go back to the pool once nothing refers to them: settle their uploads while the upload stream exists
    for (auto& kv : S->clouds) settle_cloud(*kv.second);
    for (auto& q : S->queue) { settle_cloud(*q.src); settle_cloud(*q.tgt); }
    for (size_t k = 0; k < S->slots.size(); ++k) {
      sicp_context* g = S->slots[k];
      g->collect = nullptr; g->stream = S->own1[k]; g->stream2 = S->own2[k];
      sicp_destroy(g);
    }
    for (hipEvent_t e : S->slot_ev)
      if (e) (void)hipEventDestroy(e);
    S->clouds.clear();
    S->queue.clear();
    if (S->uploader) sicp_destroy(S->uploader);
    delete S;
    return SICP_OK;
  });
}

// The worker (error_msg) or another caller thread (api_error) may be writing its string: what is handed out is a copy
// taken under the lock, stable until the next call of this function on the same stream.
const char* sicp_stream_last_error(sicp_stream S) {
  if (!S) return "";
  try {
    std::lock_guard<std::mutex> lock(S->m);
    S->error_copy = !S->error_msg.empty() ? S->error_msg : S->api_error;
    return S->error_copy.c_str();
  } catch (...) {
    return "";
  }
}

int sicp_stream_set_confusion(sicp_stream S, int32_t C, const double* cm) {
  return abi_guard(S, [&]() -> int {
    if (!S || C < 1 || C > 255 || !cm) return SICP_ERR_INVALID_ARGUMENT;
    {
      std::lock_guard<std::mutex> lock(S->m);
      if (S->submitted > 0) return SICP_ERR_INVALID_ARGUMENT;  // before the first registration
    }
    for (size_t k = 0; k < S->slots.size(); ++k) {
      // (sicp_set_confusion uploads on the handle's stream and waits for it: the slot's own stream, not the side stream)
      sicp_context* g = S->slots[k];
      hipStream_t keep = g->stream;
      g->stream = S->own1[k];
      int rc = sicp_set_confusion(g, C, cm);
      if (rc == SICP_OK) rc = ensure_hval(g, S->params.k_cov);  // (one small upload + wait per slot, here rather than in the worker)
      g->stream = keep;
      if (rc != SICP_OK) return rc;
    }
    return sicp_set_confusion(S->uploader, C, cm);
  });
}

static int stream_add_common(sicp_stream S, int32_t n, const StridedCloud& in, int64_t* cloud_id) {
  if (S->params.mode != SICP_MODE_GICP && !in.label) return SICP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> up(S->up_m);
  sicp_context* h = S->uploader;
  SICPCHECK(set_device(h));
  std::shared_ptr<Cloud> c = acquire_cloud(S->device);
  SICPCHECK(stage_cloud(h, *c, n, in));
  SICPCHECK(prepare_cloud(h, *c));  // H2D + search-tree build queued on the upload stream; ready_ev recorded
  SICPCHECK(reserve_features(h, *c));  // (here, on the submitting thread: never inside the worker's turn)
  S->last_add_ms.store(now_ms(), std::memory_order_relaxed);
  std::lock_guard<std::mutex> lock(S->m);
  const long long id = S->next_cloud++;
  S->clouds.emplace(id, std::move(c));
  *cloud_id = id;
  return SICP_OK;
}

int sicp_stream_add_cloud(sicp_stream S, int32_t n, const float* x, const float* y, const float* z, const uint32_t* label, int64_t* cloud_id) {
  return abi_guard(S, [&]() -> int {
    if (!S || !cloud_id || n < 0 || (n > 0 && (!x || !y || !z))) return SICP_ERR_INVALID_ARGUMENT;
    const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
    return stream_add_common(S, n, in, cloud_id);
  });
}

int sicp_stream_add_cloud_strided(sicp_stream S, int32_t n, const void* xyz, int64_t stride_bytes, const void* label, int64_t label_stride_bytes,
                                  int64_t* cloud_id) {
  return abi_guard(S, [&]() -> int {
    if (!S || !cloud_id || n < 0) return SICP_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!xyz || stride_bytes < 12 || (label && label_stride_bytes < 4))) return SICP_ERR_INVALID_ARGUMENT;
    const char* b = (const char*)xyz;
    const StridedCloud in = {b, b + 4, b + 8, (const char*)label, stride_bytes, label_stride_bytes};
    return stream_add_common(S, n, in, cloud_id);
  });
}

int sicp_stream_release_cloud(sicp_stream S, int64_t cloud_id) {
  return abi_guard(S, [&]() -> int {
    if (!S) return SICP_ERR_INVALID_ARGUMENT;
    // the reference is dropped AFTER the lock is gone: a cloud's deleter may wait for its upload event and, beyond the
    // pool's cap, free ~26 device buffers -- not something the worker and every submit / poll caller should queue behind
    std::shared_ptr<Cloud> dead;
    {
      std::lock_guard<std::mutex> lock(S->m);
      auto it = S->clouds.find(cloud_id);
      if (it == S->clouds.end()) return SICP_ERR_INVALID_ARGUMENT;
      dead = std::move(it->second);
      S->clouds.erase(it);
    }
    return SICP_OK;
  });
}

int sicp_stream_submit(sicp_stream S, int64_t source_id, int64_t target_id, const double init_qt[7], int64_t* ticket) {
  return abi_guard(S, [&]() -> int { return sicp_stream_submit_ex(S, source_id, target_id, init_qt, 0u, ticket); });
}

int sicp_stream_take_labels(sicp_stream S, int64_t ticket, int32_t n, uint32_t* out_labels) {
  return abi_guard(S, [&]() -> int {
    if (!S || n < 0 || (n > 0 && !out_labels)) return SICP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(S->m);
    auto it = S->labels.find(ticket);
    if (it == S->labels.end()) return SICP_ERR_NOT_READY;  // not submitted with SICP_SUBMIT_FUSED_LABELS, not finished yet, or taken already
    if ((size_t)n != it->second.size()) return SICP_ERR_INVALID_ARGUMENT;
    if (n > 0) std::memcpy(out_labels, it->second.data(), sizeof(uint32_t) * (size_t)n);
    S->labels.erase(it);
    return SICP_OK;
  });
}

int sicp_stream_submit_ex(sicp_stream S, int64_t source_id, int64_t target_id, const double init_qt[7], uint32_t flags, int64_t* ticket) {
  return abi_guard(S, [&]() -> int {
    if (!S || !init_qt) return SICP_ERR_INVALID_ARGUMENT;
    if (flags & ~(uint32_t)(SICP_SUBMIT_FUSED_LABELS | SICP_SUBMIT_FRESH_FEATURES)) return SICP_ERR_INVALID_ARGUMENT;
    if ((flags & SICP_SUBMIT_FUSED_LABELS) && S->params.mode != SICP_MODE_EM) return SICP_ERR_INVALID_ARGUMENT;  // getFusedLabels is EmIterativeClosestPoint's
    std::unique_lock<std::mutex> lock(S->m);
    if (S->error != SICP_OK) return S->error;
    // The clouds are taken (as shared references) BEFORE the back-pressure wait: the wait releases the lock, and a
    // concurrent sicp_stream_release_cloud of either id, or an add_cloud that rehashes the map, would invalidate
    // iterators held across it.  A cloud released meanwhile still takes part in this registration.
    std::shared_ptr<Cloud> src, tgt;
    {
      auto a = S->clouds.find(source_id), b = S->clouds.find(target_id);
      if (a == S->clouds.end() || b == S->clouds.end()) return SICP_ERR_INVALID_ARGUMENT;
      src = a->second; tgt = b->second;
    }
    S->cv_space.wait(lock, [&] { return S->stop || S->error != SICP_OK || (int)S->queue.size() < S->cap; });
    if (S->error != SICP_OK) return S->error;
    if (S->stop) return SICP_ERR_INVALID_ARGUMENT;
    sicp_stream_ctx::Submission q;
    q.ticket = S->next_ticket++;
    q.src = std::move(src); q.tgt = std::move(tgt);
    std::memcpy(q.init, init_qt, sizeof q.init);
    q.flags = flags;
    S->queue.push_back(std::move(q));
    ++S->submitted;
    if (ticket) *ticket = S->next_ticket - 1;
    S->cv_work.notify_all();
    return SICP_OK;
  });
}

int sicp_stream_poll(sicp_stream S, int32_t wait, int32_t max_results, sicp_stream_result* results, int32_t* n_results) {
  return abi_guard(S, [&]() -> int {
    if (!S || !n_results || max_results < 0 || (max_results > 0 && !results)) return SICP_ERR_INVALID_ARGUMENT;
    *n_results = 0;
    std::unique_lock<std::mutex> lock(S->m);
    const long long want = S->submitted;
    if (wait == 1)
      S->cv_done.wait(lock, [&] { return S->stop || S->error != SICP_OK || !S->done.empty() || S->completed >= S->submitted; });
    else if (wait >= 2) {
      ++S->draining;
      S->cv_done.wait(lock, [&] { return S->stop || S->error != SICP_OK || S->completed >= want; });
      --S->draining;
    }
    int k = 0;
    while (k < max_results && !S->done.empty()) {
      results[k++] = S->done.front();
      S->done.pop_front();
    }
    *n_results = k;
    return S->error;
  });
}

int sicp_stream_counters(sicp_stream S, int64_t* submitted, int64_t* completed, int64_t* busy_evals, int64_t* slot_evals) {
  return abi_guard(S, [&]() -> int {
    if (!S) return SICP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(S->m);
    if (submitted) *submitted = S->submitted;
    if (completed) *completed = S->completed;
    if (busy_evals) *busy_evals = S->busy_evals;
    if (slot_evals) *slot_evals = S->slot_evals;
    return SICP_OK;
  });
}

}  // extern "C"
