// sicp_api.cpp -- the C ABI of include/sicp.h (handles, clouds, align, test / bench hooks) on top of the engine
// (engine.hpp).  There is no CPU fallback: every stage runs on the GPU and every entry point fails with
// SICP_ERR_NO_DEVICE / SICP_ERR_HIP if it cannot.  Registration streams: streams.cpp.
#include "engine.hpp"

using namespace sicp::host;

// align() with the outer AND the inner loop on the host (lm_on_device = 0; a pair with caller covariances of general form)
int sicp::host::align_host_loop(sicp_context* h, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats) {
  const sicp_params& P = h->params;
  const double t_begin = now_ms();
  h->epoch = next_epoch();
  SICPCHECK(align_begin(h, stats != nullptr));
  OuterState o;
  std::memcpy(o.cur, init_qt, sizeof o.cur);
  while (!o.converged) {
    std::memcpy(o.est, o.cur, sizeof o.est);
    if (P.mode == SICP_MODE_SEMANTIC) o.count++;  // semantic_icp.hpp:47
    SICPCHECK(run_correspondences(h, o.cur, P.knn, true));
    {
      const double t0 = now_ms();
      SolveResult r;
      SICPCHECK(run_solve(h, o.est, o.est, &r));
      h->st.total_lm_iters += r.iterations;
      h->st.final_cost = r.cost;
      h->st.t_solve_ms += now_ms() - t0;
    }
    outer_finish(P, o);
  }
  std::memcpy(out_qt, o.cur, sizeof o.cur);
  SICPCHECK(align_end(h, o, t_begin, outer_iters, stats));
  return SICP_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

#define SICP_STR2(x) #x
#define SICP_STR(x) SICP_STR2(x)
const char* sicp_version(void) { return "semantic-icp_amd " SICP_STR(SICP_VERSION_MAJOR) "." SICP_STR(SICP_VERSION_MINOR) " (gfx950)"; }

const char* sicp_strerror(int s) {
  switch (s) {
    case SICP_OK: return "ok";
    case SICP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SICP_ERR_NO_DEVICE: return "no usable HIP device";
    case SICP_ERR_HIP: return "HIP runtime error (see sicp_last_error)";
    case SICP_ERR_NOT_READY: return "clouds / labels / confusion matrix not set for this mode";
    case SICP_ERR_TOO_FEW_POINTS: return "target cloud has fewer points than correspondences requested";
    case SICP_ERR_BAD_LABEL: return "EM label outside 1..C";
    case SICP_ERR_OUT_OF_MEMORY: return "out of memory";
    case SICP_ERR_INTERNAL: return "internal error caught at the ABI boundary (see sicp_last_error)";
    default: return "unknown status";
  }
}

const char* sicp_last_error(sicp_handle h) { return h ? h->last_error.c_str() : ""; }

int sicp_device_count(int* count) {
  return abi_guard([&]() -> int {
    if (!count) return SICP_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return SICP_ERR_NO_DEVICE; }
    *count = n;
    return SICP_OK;
  });
}

int sicp_default_params(int mode, sicp_params* p) {
  return abi_guard([&]() -> int {
    if (!p || mode < SICP_MODE_GICP || mode > SICP_MODE_SEMANTIC) return SICP_ERR_INVALID_ARGUMENT;
    std::memset(p, 0, sizeof *p);
    p->mode = mode;
    p->k_cov = 20;          // em_icp.h:42, gicp.h:34, semantic_point_cloud.h:31
    p->epsilon = 0.001;     // em_icp.h:43
    p->gate_sq = 250.0;     // em_icp.hpp:65, gicp.hpp:70, semantic_icp.hpp:69
    p->min_class_pts = 400; // semantic_icp.hpp:51
    p->max_lm_iterations = 400;         // em_icp.hpp:169
    p->gradient_tolerance = 0.1 * 1e-10; // 0.1 * Sophus::Constants<double>::epsilon(), em_icp.hpp:163
    p->function_tolerance = 0.1 * 1e-10; // em_icp.hpp:164
    p->parameter_tolerance = 1e-8;
    p->initial_radius = 1e4;
    p->max_radius = 1e16;
    p->min_radius = 1e-32;
    p->min_relative_decrease = 1e-3;
    p->min_lm_diagonal = 1e-6;
    p->max_lm_diagonal = 1e32;
    p->max_consecutive_invalid_steps = 5;
    p->jacobi_scaling = 1;
    p->quirk_bool_probability = 1;
    p->quirk_float_products = 1;
    p->nn_method = 1;  // exact box-tree search, packet walk; 2 = per-query walk; 0 = LDS-tiled brute force (same results)
    p->lm_on_device = 1;
    p->lm_batch = 8;   // 16 kernel nodes per graph: longer graphs replay with a ~50 us bubble every 16 nodes
    if (mode == SICP_MODE_EM) {
      p->knn = 4; p->cauchy_a = 3.0; p->use_sqloss = 1;  // em_icp.hpp:60,111,115
      p->outer_tol = 1e-5; p->max_outer = 50;            // em_icp.hpp:180
    } else if (mode == SICP_MODE_GICP) {
      p->knn = 1; p->cauchy_a = 3.0; p->use_sqloss = 1;  // gicp.hpp:69,100,102
      p->outer_tol = 1e-5; p->max_outer = 50;            // gicp.hpp:154
    } else {
      p->knn = 1; p->cauchy_a = 1.5; p->use_sqloss = 0;  // semantic_icp.hpp:68,96
      p->outer_tol = 0.001; p->max_outer = 35;           // semantic_icp.hpp:152
    }
    return SICP_OK;
  });
}

// Handles are recycled like clouds: sicp_destroy parks a handle -- streams, events, pinned mirrors, device buffers, tick
// graphs and all -- in a per-device pool and sicp_create hands it out again, reset to what a new handle is.  The
// reference's drivers construct their registration objects PER PAIR (exec/kitti_eval.cc:176,198, exec/nyu_eval.cc:140,183), and a
// handle made from nothing costs ~4 ms to create, ~5 ms of first-use allocations inside its first align() and ~9 ms to
// destroy (every device buffer given back waits for the device): 49.6 ms per KITTI pair through the class shims against
// ~2 ms of align() (profiles/r06/dropin_calls_before.json).  sicp_release_pool frees the parked handles too.
namespace {
struct HandlePool {
  std::mutex m;
  std::vector<sicp_context*> parked[sicp::host::kPoolDevices];
};
HandlePool& handle_pool() {
  static HandlePool* p = new HandlePool;  // never destroyed: it may outlive the HIP runtime at process exit
  return *p;
}
constexpr size_t kHandlePoolCap = 32;  // per device (a parked handle keeps its buffers: ~12 MB at 100K x 4 slots)

// what a new handle is, for one that has been used
void reset_handle(sicp_context* h) {
  h->collect = nullptr;
  sicp_default_params(SICP_MODE_GICP, &h->params);
  h->epoch = 0;
  h->C = 0; h->cm.clear(); h->cm_id = 0;
  h->corr_n = h->corr_K = 0;
  h->corr_valid = h->corr_weighted = h->hint_ok = false;
  for (TickSet& S : h->ts) { S.tick_valid = false; S.tick_act.clear(); }
  h->solo_skip = h->solo_penalty = 0;
  h->solo_was_init = h->solo_failed = false;
  h->count_stats = h->counted_in_search = false;
  h->wait_on_device = false;
  h->last_error.clear();
  std::memset(&h->st, 0, sizeof h->st);
}

int destroy_for_real(sicp_context* h);
}  // namespace

int sicp_create(int device_id, sicp_handle* out) {
  return abi_guard([&]() -> int {
    if (!out) return SICP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SICP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return SICP_ERR_INVALID_ARGUMENT;
    {
      HandlePool& pool = handle_pool();
      sicp_context* h = nullptr;
      {
        std::lock_guard<std::mutex> lock(pool.m);
        auto& v = pool.parked[device_id % kPoolDevices];
        for (size_t i = v.size(); i-- > 0;)
          if (v[i]->device == device_id) { h = v[i]; v.erase(v.begin() + (long)i); break; }
      }
      if (h) {
        h->cl[0] = acquire_cloud(device_id);
        h->cl[1] = acquire_cloud(device_id);
        *out = h;
        return SICP_OK;
      }
    }
    sicp_context* h = new (std::nothrow) sicp_context();
    if (!h) return SICP_ERR_OUT_OF_MEMORY;
    h->device = device_id;
    h->cl[0] = acquire_cloud(device_id);
    h->cl[1] = acquire_cloud(device_id);
    sicp_default_params(SICP_MODE_GICP, &h->params);
    std::memset(&h->st, 0, sizeof h->st);
    bool ok = hipSetDevice(device_id) == hipSuccess &&
              hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreate(&h->ev0) == hipSuccess && hipEventCreate(&h->ev1) == hipSuccess &&
              hipHostMalloc((void**)&h->h_out28, sizeof(double) * 28, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&h->h_count, sizeof(long long) * sicp::kLiveCounters, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&h->h_lm, sizeof(sicp::LmState), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
      destroy_for_real(h);
      return SICP_ERR_NO_DEVICE;
    }
    *out = h;
    return SICP_OK;
  });
}

int sicp_destroy(sicp_handle h) {
  return abi_guard(h, [&]() -> int {
    if (!h) return SICP_OK;
    // park it: everything it queued has ended, its clouds go back to their pool (or stay with whoever shares them), and
    // what remains is a handle as sicp_create makes them -- with its streams, mirrors, buffers and graphs in place
    static const bool no_pool = std::getenv("SICP_NO_HANDLE_POOL") != nullptr;  // A/B aid
    bool clean = !no_pool && h->stream && h->stream2 && hipSetDevice(h->device) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess &&
                 hipStreamSynchronize(h->stream2) == hipSuccess;
    if (clean && h->side_stream) clean = hipStreamSynchronize(h->side_stream) == hipSuccess;
    if (clean && h->feat_stream) clean = hipStreamSynchronize(h->feat_stream) == hipSuccess;
    for (int s2 = 1; clean && s2 < kParts; ++s2)
      if (h->part_stream[s2]) clean = hipStreamSynchronize(h->part_stream[s2]) == hipSuccess;
    if (clean) {
      for (auto& c : h->cl)
        if (c) { settle_cloud(*c); c.reset(); }
      reset_handle(h);
      HandlePool& pool = handle_pool();
      std::lock_guard<std::mutex> lock(pool.m);
      auto& v = pool.parked[h->device % kPoolDevices];
      if (v.size() < kHandlePoolCap) { v.push_back(h); return SICP_OK; }
    }
    return destroy_for_real(h);
  });
}

namespace {
int destroy_for_real(sicp_context* h) {
  {
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& c : h->cl)
      if (c) settle_cloud(*c);  // before the streams go (see settle_cloud)
    if (h->h_out28) (void)hipHostFree(h->h_out28);
    if (h->h_count) (void)hipHostFree(h->h_count);
    if (h->h_lm) (void)hipHostFree(h->h_lm);
    for (TickSet& S : h->ts) {
      if (S.h_batch) (void)hipHostFree(S.h_batch);
      if (S.h_join) (void)hipHostFree(S.h_join);
      if (S.h_bhdr) (void)hipHostFree(S.h_bhdr);
      sicp::batch_graph_destroy(S.graph[0]);
      sicp::batch_graph_destroy(S.graph[1]);
    }
    if (h->h_bstates) (void)hipHostFree(h->h_bstates);
    if (h->h_solo_flag) (void)hipHostFree(h->h_solo_flag);
    if (h->h_bout28) (void)hipHostFree(h->h_bout28);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side_done) (void)hipEventDestroy(h->side_done);
    if (h->side_done2) (void)hipEventDestroy(h->side_done2);
    if (h->main_done) (void)hipEventDestroy(h->main_done);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->feat_stream) (void)hipStreamDestroy(h->feat_stream);
    for (hipEvent_t e : h->chunk_ev) (void)hipEventDestroy(e);
    if (h->part_fork) (void)hipEventDestroy(h->part_fork);
    for (int s = 1; s < kParts; ++s) {
      if (h->part_done[s]) (void)hipEventDestroy(h->part_done[s]);
      if (h->part_stream[s]) (void)hipStreamDestroy(h->part_stream[s]);
    }
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    {
      DevArena::FreeScope once(h->device);  // one wait for the device, not one per buffer of the handle
      delete h;
    }
    return SICP_OK;
  }
}
}  // namespace

int sicp_release_pool(int device_id) {
  return abi_guard([&]() -> int {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return SICP_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n_dev) return SICP_ERR_INVALID_ARGUMENT;
    CloudPool& pool = cloud_pool();
    std::vector<Cloud*> dead;
    {
      std::lock_guard<std::mutex> lock(pool.m);
      dead.swap(pool.free_list[device_id % kPoolDevices]);
    }
    if (hipSetDevice(device_id) != hipSuccess) return SICP_ERR_NO_DEVICE;
    {  // the parked handles of this device (their buffers are part of what the pool holds)
      std::vector<sicp_context*> idle;
      {
        HandlePool& hp = handle_pool();
        std::lock_guard<std::mutex> lock(hp.m);
        auto& v = hp.parked[device_id % kPoolDevices];
        for (size_t i = v.size(); i-- > 0;)
          if (v[i]->device == device_id) { idle.push_back(v[i]); v.erase(v.begin() + (long)i); }
      }
      for (sicp_context* g : idle) (void)destroy_for_real(g);
    }
    for (Cloud* c : dead) delete c;
    (void)hipDeviceSynchronize();   // nothing may still be running out of a block that goes back to the driver
    dev_arena().release(device_id);  // the slabs no live buffer is carved from
    return SICP_OK;
  });
}

int sicp_set_memory_limit(int device_id, int64_t bytes) {
  return abi_guard([&]() -> int {
    if (device_id < 0 || device_id >= kArenaDevices || bytes < 0) return SICP_ERR_INVALID_ARGUMENT;
    DevArena& A = dev_arena();
    std::lock_guard<std::mutex> lock(A.m);
    A.dev[device_id].limit = (size_t)bytes;
    return SICP_OK;
  });
}

int sicp_memory_reserved(int device_id, int64_t* bytes) {
  return abi_guard([&]() -> int {
    if (device_id < 0 || device_id >= kArenaDevices || !bytes) return SICP_ERR_INVALID_ARGUMENT;
    DevArena& A = dev_arena();
    std::lock_guard<std::mutex> lock(A.m);
    *bytes = (int64_t)A.dev[device_id].reserved;
    return SICP_OK;
  });
}

int sicp_set_params(sicp_handle h, const sicp_params* p) {
  return abi_guard(h, [&]() -> int {
    if (!h || !p) return SICP_ERR_INVALID_ARGUMENT;
    if (p->mode < SICP_MODE_GICP || p->mode > SICP_MODE_SEMANTIC) return SICP_ERR_INVALID_ARGUMENT;
    if (!sicp::nn_k_supported(p->knn)) {
      h->last_error = "sicp_set_params: knn (correspondences per source point) must be 1, 4 or 20";
      return SICP_ERR_INVALID_ARGUMENT;
    }
    if (sicp::nn_list_len(p->k_cov) == 0) {
      h->last_error = "sicp_set_params: k_cov (covariance neighbourhood, the constructors' k) must be in 1..32";
      return SICP_ERR_INVALID_ARGUMENT;
    }
    if (!(p->epsilon > 0) || !(p->cauchy_a > 0) || p->nn_method < 0 || p->nn_method > 2) return SICP_ERR_INVALID_ARGUMENT;
    // engine knobs (profiling, batching) do not invalidate the correspondences held on the device
    sicp_params a = h->params, b = *p;
    a.profile = b.profile = 0; a.lm_batch = b.lm_batch = 0; a.lm_on_device = b.lm_on_device = 0;
    if (std::memcmp(&a, &b, sizeof a) != 0) { h->corr_valid = false; h->hint_ok = false; }
    h->params = *p;
    return SICP_OK;
  });
}

int sicp_get_params(sicp_handle h, sicp_params* p) {
  return abi_guard(h, [&]() -> int {
    if (!h || !p) return SICP_ERR_INVALID_ARGUMENT;
    *p = h->params;
    return SICP_OK;
  });
}

int sicp_set_cloud(sicp_handle h, int which, int32_t n, const float* x, const float* y, const float* z,
                   const uint32_t* label) {
  return abi_guard(h, [&]() -> int {
    if (!h || (which != SICP_SOURCE && which != SICP_TARGET) || n < 0) return SICP_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!x || !y || !z)) return SICP_ERR_INVALID_ARGUMENT;
    const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
    return set_cloud_common(h, which, n, in);
  });
}

int sicp_set_cloud_strided(sicp_handle h, int which, int32_t n, const void* xyz, int64_t stride_bytes, const void* label,
                           int64_t label_stride_bytes) {
  return abi_guard(h, [&]() -> int {
    if (!h || (which != SICP_SOURCE && which != SICP_TARGET) || n < 0) return SICP_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!xyz || stride_bytes < 12 || (label && label_stride_bytes < 4))) return SICP_ERR_INVALID_ARGUMENT;
    const char* b = (const char*)xyz;
    const StridedCloud in = {b, b + 4, b + 8, (const char*)label, stride_bytes, label_stride_bytes};
    return set_cloud_common(h, which, n, in);
  });
}

int sicp_set_cloud_device(sicp_handle h, int which, int32_t n, const float* xd, const float* yd, const float* zd,
                          const uint32_t* ld) {
  return abi_guard(h, [&]() -> int {
    if (!h || n < 0 || (n > 0 && (!xd || !yd || !zd))) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    // the host keeps a copy of every cloud (regrouping for SICP_MODE_SEMANTIC and the final
    // un-permutation need it), so a device-resident input is mirrored once
    std::vector<float> x(n), y(n), z(n);
    std::vector<uint32_t> l(ld ? n : 0);
    if (n > 0) {
      HIPCHECK(hipMemcpy(x.data(), xd, sizeof(float) * n, hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(y.data(), yd, sizeof(float) * n, hipMemcpyDeviceToHost));
      HIPCHECK(hipMemcpy(z.data(), zd, sizeof(float) * n, hipMemcpyDeviceToHost));
      if (ld) HIPCHECK(hipMemcpy(l.data(), ld, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    }
    return sicp_set_cloud(h, which, n, x.data(), y.data(), z.data(), ld ? l.data() : nullptr);
  });
}

int sicp_share_cloud(sicp_handle h, int which, sicp_handle from, int from_which) {
  return abi_guard(h, [&]() -> int {
    if (!h || !from || (which != SICP_SOURCE && which != SICP_TARGET) || (from_which != SICP_SOURCE && from_which != SICP_TARGET))
      return SICP_ERR_INVALID_ARGUMENT;
    if (h->device != from->device) {
      h->last_error = "sicp_share_cloud: the handles are on different devices";
      return SICP_ERR_INVALID_ARGUMENT;
    }
    if (!from->cloud(from_which).is_set) return SICP_ERR_NOT_READY;
    if (h->cl[which] && h->cl[which] != from->cl[from_which]) settle_cloud(*h->cl[which]);
    h->cl[which] = from->cl[from_which];
    h->corr_valid = false;
    h->hint_ok = false;
    return SICP_OK;
  });
}

int sicp_cloud_size(sicp_handle h, int which, int32_t* n_points, int32_t* n_indexed) {
  return abi_guard(h, [&]() -> int {
    if (!h || (which != SICP_SOURCE && which != SICP_TARGET)) return SICP_ERR_INVALID_ARGUMENT;
    const Cloud& c = h->cloud(which);
    if (!c.is_set) return SICP_ERR_NOT_READY;
    if (n_points) *n_points = c.n_caller;
    if (n_indexed) *n_indexed = c.n;
    return SICP_OK;
  });
}

int sicp_set_confusion(sicp_handle h, int32_t C, const double* cm) {
  return abi_guard(h, [&]() -> int {
    if (!h || C < 1 || C > 255 || !cm) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    h->C = C;
    h->cm.assign(cm, cm + (size_t)C * C);
    {  // FNV-1a of the contents: handles holding the same matrix share the projections of a shared cloud
      unsigned long long id = 1469598103934665603ull ^ (unsigned long long)C;
      const unsigned char* b = reinterpret_cast<const unsigned char*>(h->cm.data());
      for (size_t i = 0; i < sizeof(double) * h->cm.size(); ++i) id = (id ^ b[i]) * 1099511628211ull;
      h->cm_id = id;
    }
    HIPCHECK(h->d_cm.reserve((size_t)C * C));
    HIPCHECK(hipMemcpyAsync(h->d_cm.p, h->cm.data(), sizeof(double) * C * C, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return SICP_OK;
  });
}

int sicp_align(sicp_handle h, const double init_qt[7], double out_qt[7], int32_t* outer_iters, sicp_stats* stats) {
  return abi_guard(h, [&]() -> int {
    if (!h || !init_qt || !out_qt) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    SICPCHECK(check_ready(h, false));
    const sicp_params& P = h->params;
    // device-resident solve: a lock-step batch of one (the same kernels, hence the same bits, as any batch)
    if (P.lm_on_device && !general_covariances(h)) return sicp_align_batch(&h, 1, init_qt, out_qt, outer_iters, stats);
    return align_host_loop(h, init_qt, out_qt, outer_iters, stats);
  });
}

int sicp_align_batch(sicp_handle* hs, int32_t n, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats) {
  return abi_guard((hs && n > 0) ? hs[0] : nullptr, [&]() -> int {
    return align_batch(hs, n, init_qt, out_qt, outer_iters, stats);
  });
}

int sicp_accumulate_batch(sicp_handle* hs, int32_t n, const double* qt, double* out28, int32_t repeat, double* kernel_ms) {
  return abi_guard((hs && n > 0) ? hs[0] : nullptr, [&]() -> int {
    if (!hs || n < 1 || !qt || !out28) return SICP_ERR_INVALID_ARGUMENT;
    for (int p = 0; p < n; ++p)
      if (hs[p] && general_covariances(hs[p])) {
        hs[p]->last_error = "sicp_accumulate_batch: a handle with caller covariances of general form evaluates one pair at a time (sicp_accumulate)";
        return SICP_ERR_INVALID_ARGUMENT;
      }
    for (int p = 0; p < n; ++p)
      if (!hs[p] || !hs[p]->corr_valid || hs[p]->device != hs[0]->device || hs[p]->corr_K != hs[0]->corr_K) return SICP_ERR_NOT_READY;
    sicp_context* h = hs[0];
    if (n > kMaxActivePairs) return SICP_ERR_INVALID_ARGUMENT;  // one launch holds at most this many pairs
    SICPCHECK(set_device(h));
    SICPCHECK(batch_reserve(h, n));
    h->ts[0].tick_valid = false;
    for (int p = 0; p < n; ++p) {
      sicp_context* g = hs[p];
      const int nb = sicp::accumulate_blocks(g->corr_n * g->corr_K, g->corr_K);
      if (g->partials.reserve((size_t)nb * 28) != hipSuccess) return SICP_ERR_OUT_OF_MEMORY;
      std::memset(&h->ts[0].h_batch[p], 0, sizeof(sicp::BatchArgs));
      fill_acc(g, h->ts[0].h_batch[p].a);
      fill_pose(qt + 7 * p, h->ts[0].h_batch[p].a.pose);
      h->ts[0].h_batch[p].nb = nb;
      HIPCHECK(hipStreamSynchronize(g->stream));  // the pair's correspondences are complete
    }
#ifdef SICP_DEV_PROBES
    // (developer build only: SICP_ACC_INNER_REPEAT makes the kernel repeat its range that many times inside one launch)
    static const int inner = [] { const char* e = std::getenv("SICP_ACC_INNER_REPEAT"); return e ? std::atoi(e) : 0; }();
#else
    const int inner = 0;
#endif
    *h->ts[0].h_bhdr = sicp::BatchHeader{n, h->ts[0].epoch_host, {inner, 0}};
    HIPCHECK(hipMemcpyAsync(h->ts[0].d_bhdr.p, h->ts[0].h_bhdr, sizeof(sicp::BatchHeader), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(h->ts[0].d_batch.p, h->ts[0].h_batch, sizeof(sicp::BatchArgs) * n, hipMemcpyHostToDevice, h->stream));
    if (repeat < 1) repeat = 1;
    HIPCHECK(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < repeat; ++r)
      HIPCHECK(sicp::launch_accumulate_batch(h->corr_K, h->params.use_sqloss, h->ts[0].d_bhdr.p, h->ts[0].d_batch.p, std::min(h->ts[0].cap, kMaxActivePairs),
                                             h->stream));
    HIPCHECK(hipEventRecord(h->ev1, h->stream));
    HIPCHECK(sicp::launch_finalize_batch(h->ts[0].d_batch.p, n, h->d_bout28.p, h->stream));
    HIPCHECK(hipMemcpyAsync(h->h_bout28, h->d_bout28.p, sizeof(double) * 28 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    std::memcpy(out28, h->h_bout28, sizeof(double) * 28 * n);
    if (kernel_ms) {
      float ms = 0.f;
      HIPCHECK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      *kernel_ms = (double)ms / repeat;
    }
    return SICP_OK;
  });
}

int sicp_search_batch(sicp_handle* hs, int32_t n, const double* qt, int32_t what, int32_t use_hint, int32_t repeat, double* kernel_ms) {
  return abi_guard((hs && n > 0) ? hs[0] : nullptr, [&]() -> int {
    if (!hs || n < 1 || what < 0 || what > 3 || ((what == 0 || what == 3) && !qt)) return SICP_ERR_INVALID_ARGUMENT;
    sicp_context* h = hs[0];
    SICPCHECK(set_device(h));
    for (int p = 0; p < n; ++p) {
      if (!hs[p] || hs[p]->device != h->device || !same_solver(hs[p]->params, h->params) || hs[p]->params.nn_method != 1) return SICP_ERR_INVALID_ARGUMENT;
      sicp_context* g = hs[p];
      {
        sicp_context* h = g;  // (HIPCHECK / SICPCHECK report into the handle they run on)
        SICPCHECK(check_ready(h, false));
        SICPCHECK(prepare_cloud(h, h->cloud(0)));
        SICPCHECK(prepare_cloud(h, h->cloud(1)));
        if (what == 3) {  // the search as an align() launches it, EM weights included: the features it reads must be there
          if (h->params.mode != SICP_MODE_EM) return SICP_ERR_INVALID_ARGUMENT;
          SICPCHECK(check_ready(h, true));
          for (int c = 0; c < 2; ++c)
            if (!features_current(h, h->cloud(c), true)) { h->epoch = next_epoch(); SICPCHECK(compute_features(h, h->cloud(c), true)); }
          if (!weights_from_histograms(h->params, h->params.knn)) { SICPCHECK(ensure_proj(h, h->cloud(0))); SICPCHECK(ensure_proj(h, h->cloud(1))); }
        }
        HIPCHECK(hipStreamSynchronize(h->stream));
      }
    }
    // the handles' own stage drivers collect the jobs exactly as sicp_align_batch makes them
    JobCollector jc;
    jc.fold_weights = what == 3;
    {
      BatchGuard guard(hs, n, &jc, h->stream);
      for (int p = 0; p < n; ++p) {
        sicp_context* g = hs[p];
        jc.slice = 0;
        if (what == 0 || what == 3) {
          if (!use_hint) g->hint_ok = false;
          const int rc = run_correspondences(g, qt + 7 * p, g->params.knn, what == 3);
          if (rc != SICP_OK) return rc;
        } else {
          g->epoch = next_epoch();
          const int rc = compute_features(g, g->cloud(what == 1 ? SICP_SOURCE : SICP_TARGET), g->params.mode == SICP_MODE_EM);
          if (rc != SICP_OK) return rc;
        }
      }
    }
    if (repeat < 1) repeat = 1;
    const int L = jc.knn_K[0];
    HIPCHECK(hipEventRecord(h->ev0, h->stream));
    for (int r = 0; r < repeat; ++r)
      if (!jc.knn[0].empty()) HIPCHECK(sicp::launch_bvh_knn_packet_jobs(L, jc.knn[0].data(), (int)jc.knn[0].size(), h->stream));
    HIPCHECK(hipEventRecord(h->ev1, h->stream));
    // whatever consumes the searches (covariances, histograms, projections) runs once, so the handles stay consistent
    jc.knn[0].clear();
    SICPCHECK(flush_jobs(h, jc, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (kernel_ms) {
      float ms = 0.f;
      HIPCHECK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      *kernel_ms = (double)ms / repeat;
    }
    return SICP_OK;
  });
}

int sicp_transform_source(sicp_handle h, const double qt[7], float* ox, float* oy, float* oz) {
  return abi_guard(h, [&]() -> int {
    if (!h || !qt || !ox || !oy || !oz) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    Cloud& S = h->cloud(0);
    SICPCHECK(prepare_cloud(h, S));
    const int n = S.n;
    const size_t m = (size_t)(n > 0 ? n : 1);
    HIPCHECK(h->tmpx.reserve(m)); HIPCHECK(h->tmpy.reserve(m)); HIPCHECK(h->tmpz.reserve(m));
    double M[12];
    matrix34(qt, M);
    sicp::Mat4f Mf;
    for (int i = 0; i < 12; ++i) Mf.m[i] = (float)M[i];  // (trans.matrix()).cast<float>(), em_icp.hpp:193
    Mf.m[12] = Mf.m[13] = Mf.m[14] = 0.f; Mf.m[15] = 1.f;
    HIPCHECK(sicp::launch_transform_float(n, S.x.p, S.y.p, S.z.p, Mf, h->tmpx.p, h->tmpy.p, h->tmpz.p, h->stream));
    std::vector<float> bx(n), by(n), bz(n);
    if (n > 0) {
      HIPCHECK(hipMemcpyAsync(bx.data(), h->tmpx.p, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(by.data(), h->tmpy.p, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(bz.data(), h->tmpz.p, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (int d = 0; d < n; ++d) {
      const int i = S.caller_index(d);
      ox[i] = bx[d]; oy[i] = by[d]; oz[i] = bz[d];
    }
    // the non-finite points that never went to the device: the same float arithmetic, here
    for (size_t k = 0; k < S.drop_i.size(); ++k) {
      const int i = S.drop_i[k];
      const float px = S.drop_xyz[3 * k], py = S.drop_xyz[3 * k + 1], pz = S.drop_xyz[3 * k + 2];
      const float* m = Mf.m;
      // (this file is compiled with -ffp-contract=off: separately rounded products and sums, like transform_float_kernel)
      auto row = [&](int r) { return ((m[4 * r] * px + m[4 * r + 1] * py) + m[4 * r + 2] * pz) + m[4 * r + 3]; };
      ox[i] = row(0); oy[i] = row(1); oz[i] = row(2);
    }
    return SICP_OK;
  });
}

int sicp_set_covariances(sicp_handle h, int which, const double* cov9) {
  return abi_guard(h, [&]() -> int {
    if (!h || !cov9 || (which != SICP_SOURCE && which != SICP_TARGET)) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    const sicp_params& P = h->params;
    Cloud& c = h->cloud(which);
    if (!c.is_set) return SICP_ERR_NOT_READY;
    SICPCHECK(prepare_cloud(h, c));
    SICPCHECK(cloud_wait(h, c));  // (perm is read below)
    const double kappa = 1.0 - P.epsilon;
    if (!(kappa > 0.0)) { h->last_error = "sicp_set_covariances: params.epsilon must be below 1"; return SICP_ERR_INVALID_ARGUMENT; }
    const int n = c.n;
    const size_t m = (size_t)(n > 0 ? n : 1);
    std::vector<double> nrm(m * 3, 0.0), c6;
    // The engine's own form, C = I - kappa n n^T  <=>  (I - C) / kappa = n n^T: symmetric, rank one, trace one.  The normal is
    // the column with the largest diagonal entry, scaled; what is left after taking n n^T away is the distance from the form.
    // A cloud whose matrices ALL have it keeps its normals and runs the product kernels.  Otherwise the matrices are kept as
    // they are (symmetric and finite is all the closed-form Jacobian of SURVEY 8a / a7 needs) and the cloud's registrations
    // run through the full-matrix evaluation, one pair at a time (solve_kernels.hip: accumulate_general_kernel).
    constexpr double tol = 1e-8;
    bool normal_form = true;
    int first_general = -1;
    for (int d = 0; d < n; ++d) {
      const int ci = c.caller_index(d);
      const double* C9 = cov9 + (size_t)ci * 9;
      bool finite = true, symmetric = true;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          finite = finite && std::isfinite(C9[3 * a + b]);
          symmetric = symmetric && std::fabs(C9[3 * a + b] - C9[3 * b + a]) <= tol * (1.0 + std::fabs(C9[3 * a + b]));
        }
      if (!finite || !symmetric) {
        char msg[256];
        std::snprintf(msg, sizeof msg, "sicp_set_covariances: the covariance of point %d is not a finite symmetric 3x3 matrix; nothing was changed", ci);
        h->last_error = msg;
        return SICP_ERR_INVALID_ARGUMENT;
      }
      if (!normal_form) continue;
      double M[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) M[a][b] = ((a == b ? 1.0 : 0.0) - C9[3 * a + b]) / kappa;
      int col = 0;
      if (M[1][1] > M[col][col]) col = 1;
      if (M[2][2] > M[col][col]) col = 2;
      bool ok = M[col][col] > 0.0;
      double nv[3] = {0, 0, 0};
      if (ok) {
        const double sc = 1.0 / std::sqrt(M[col][col]);
        for (int a = 0; a < 3; ++a) nv[a] = M[a][col] * sc;
        const double len = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
        ok = std::fabs(len - 1.0) <= tol;
        for (int a = 0; a < 3 && ok; ++a)
          for (int b = 0; b < 3; ++b) ok = ok && std::fabs(M[a][b] - nv[a] * nv[b]) <= tol;
        if (ok) for (int a = 0; a < 3; ++a) nv[a] /= len;
      }
      if (!ok) { normal_form = false; first_general = ci; continue; }
      nrm[3 * (size_t)d] = nv[0]; nrm[3 * (size_t)d + 1] = nv[1]; nrm[3 * (size_t)d + 2] = nv[2];
    }
    if (!normal_form) {
      if (P.mode == SICP_MODE_EM) {  // (its align() recomputes covariances and label histograms together: impl/em_icp.hpp:28-29)
        char msg[256];
        std::snprintf(msg, sizeof msg, "sicp_set_covariances: the covariance of point %d is not I - (1 - epsilon) n n^T (epsilon = %g); general matrices are taken "
                                       "in SICP_MODE_GICP / SICP_MODE_SEMANTIC only; nothing was changed", first_general, P.epsilon);
        h->last_error = msg;
        return SICP_ERR_INVALID_ARGUMENT;
      }
      std::fill(nrm.begin(), nrm.end(), 0.0);  // (the records of such a cloud carry positions only)
      c6.assign(m * 6, 0.0);
      for (int d = 0; d < n; ++d) {
        const double* C9 = cov9 + (size_t)c.caller_index(d) * 9;
        double* o = &c6[6 * (size_t)d];
        o[0] = C9[0]; o[1] = 0.5 * (C9[1] + C9[3]); o[2] = 0.5 * (C9[2] + C9[6]); o[3] = C9[4]; o[4] = 0.5 * (C9[5] + C9[7]); o[5] = C9[8];
      }
    }
    HIPCHECK(c.rec.reserve(m));
    HIPCHECK(c.rec_dense.reserve(sicp::dense_rec_bytes(n)));
    if (!normal_form) HIPCHECK(c.cov6.reserve(m * 6));
    DevBuf<double> d_nrm;
    HIPCHECK(d_nrm.reserve(m * 3));
    if (n == 0) HIPCHECK(hipMemsetAsync(c.rec.p, 0, sizeof(sicp::PointRec), h->stream));  // (compute_features: the record dead slots are evaluated on)
    if (n > 0) HIPCHECK(hipMemcpyAsync(d_nrm.p, nrm.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->stream));
    if (n > 0 && !normal_form) HIPCHECK(hipMemcpyAsync(c.cov6.p, c6.data(), sizeof(double) * 6 * n, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(sicp::launch_set_normals(n, c.x.p, c.y.p, c.z.p, d_nrm.p, c.rec.p, c.rec_dense.p, n, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));  // (nrm / c6 / d_nrm go out of scope)
    c.rec_dense_n = n;
    c.feat_valid = true;
    c.cov_general = !normal_form;
    c.proj_valid = false;
    c.feat_k = P.k_cov; c.feat_C = 0;
    c.feat_float_products = P.quirk_float_products;
    c.feat_hist = false;
    c.feat_epoch = h->epoch;
    h->corr_valid = false;
    return SICP_OK;
  });
}

int sicp_covariances(sicp_handle h, int which, double* cov9, double* normal3, uint8_t* hist, int32_t* nn_idx) {
  return abi_guard(h, [&]() -> int {
    if (!h || (which != SICP_SOURCE && which != SICP_TARGET)) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    const sicp_params& P = h->params;
    if (sicp::nn_list_len(P.k_cov) == 0) return SICP_ERR_INVALID_ARGUMENT;
    Cloud& c = h->cloud(which);
    SICPCHECK(prepare_cloud(h, c));
    const bool with_hist = P.mode == SICP_MODE_EM && c.has_label && P.num_classes >= 1;
    if (hist && !with_hist) return SICP_ERR_NOT_READY;
    if (with_hist && c.n > 0 && (c.label_min < 1 || c.label_max > (uint32_t)P.num_classes)) return SICP_ERR_BAD_LABEL;
    // what align() left behind is returned as it is (getSourceCovariances(), gicp.h:72-90)
    if (!features_current(h, c, with_hist) || (nn_idx && !c.nn.p)) SICPCHECK(compute_features(h, c, with_hist));
    const int n = c.n, k = P.k_cov;
    if (cov9 && !normal3 && !hist && !nn_idx && c.keep.empty() && n > 0) {
      // the covariances alone of a cloud without dropped points (getSourceCovariances() / getTargetCovariances(), gicp.h:72-90,
      // called once per pair by exec/kitti_eval.cc:225-226): formed in the caller's order on the device and copied straight
      // into the caller's array -- one pass instead of a record read-back, a scatter on the host and the caller's own copy
      HIPCHECK(h->tmp9.reserve((size_t)n * 9));
      HIPCHECK(sicp::launch_cov9_caller_order(n, c.rec.p, c.cov_general ? c.cov6.p : nullptr, c.d_perm.p, 1.0 - P.epsilon, h->tmp9.p, h->stream));
      HIPCHECK(hipMemcpyAsync(cov9, h->tmp9.p, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipStreamSynchronize(h->stream));
      return SICP_OK;
    }
    std::vector<sicp::PointRec> rec(n);
    if (n > 0) HIPCHECK(hipMemcpyAsync(rec.data(), c.rec.p, sizeof(sicp::PointRec) * n, hipMemcpyDeviceToHost, h->stream));
    std::vector<double> g6;  // caller covariances of general form: what is there is what comes back
    if (c.cov_general && n > 0) {
      g6.resize((size_t)n * 6);
      HIPCHECK(hipMemcpyAsync(g6.data(), c.cov6.p, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, h->stream));
    }
    std::vector<uint8_t> hh;
    std::vector<int> nn;
    const size_t HS = (size_t)sicp::hist_stride(P.num_classes);  // rows are padded to 16 bytes on the device
    if (hist && n > 0) {
      hh.resize((size_t)n * HS);
      HIPCHECK(hipMemcpyAsync(hh.data(), c.hist.p, hh.size(), hipMemcpyDeviceToHost, h->stream));
    }
    if (nn_idx && n > 0) {
      nn.resize((size_t)n * k);
      HIPCHECK(hipMemcpyAsync(nn.data(), c.nn.p, sizeof(int) * nn.size(), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHECK(hipStreamSynchronize(h->stream));
    const double ome = 1.0 - P.epsilon;
    for (int i : c.drop_i) {  // points outside the index have no neighbourhood (the reference's values for them are undefined)
      const double nan = std::numeric_limits<double>::quiet_NaN();
      if (normal3) for (int a = 0; a < 3; ++a) normal3[3 * (size_t)i + a] = nan;
      if (cov9) for (int a = 0; a < 9; ++a) cov9[9 * (size_t)i + a] = nan;
      if (hist) std::memset(hist + (size_t)i * P.num_classes, 0, P.num_classes);
      if (nn_idx) for (int j = 0; j < k; ++j) nn_idx[(size_t)i * k + j] = -1;
    }
    for (int d = 0; d < n; ++d) {
      const int i = c.caller_index(d);
      const double v[3] = {rec[d].nx, rec[d].ny, rec[d].nz};
      if (!g6.empty()) {  // (no normal belongs to a general matrix)
        const double* q = &g6[6 * (size_t)d];
        const double full[9] = {q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5]};
        if (cov9) std::memcpy(cov9 + 9 * (size_t)i, full, sizeof full);
        if (normal3) for (int a = 0; a < 3; ++a) normal3[3 * (size_t)i + a] = std::numeric_limits<double>::quiet_NaN();
      } else {
      if (normal3) { normal3[3 * (size_t)i] = v[0]; normal3[3 * (size_t)i + 1] = v[1]; normal3[3 * (size_t)i + 2] = v[2]; }
      if (cov9)  // the covariance the kernels use: C = I - (1-eps) n n^T  (== em_icp.hpp:331-338); the upper triangle, mirrored:
        for (int a = 0; a < 3; ++a)  // bit-symmetric, so the rows a caller reads are also the columns Eigen stores
          for (int b = a; b < 3; ++b) {
            const double e = (a == b ? 1.0 : 0.0) - ome * v[a] * v[b];
            cov9[9 * (size_t)i + 3 * a + b] = e;
            cov9[9 * (size_t)i + 3 * b + a] = e;
          }
      }
      if (hist) std::memcpy(hist + (size_t)i * P.num_classes, hh.data() + (size_t)d * HS, P.num_classes);
      if (nn_idx)
        for (int j = 0; j < k; ++j) {
          const int g = c.nn_stride > 0 ? nn[(size_t)j * c.nn_stride + d] : nn[(size_t)d * k + j];
          nn_idx[(size_t)i * k + j] = g < 0 ? -1 : c.caller_index(g);
        }
    }
    return SICP_OK;
  });
}

int sicp_correspondences(sicp_handle h, const double qt[7], int32_t* idx, float* d2, double* w) {
  return abi_guard(h, [&]() -> int {
    if (!h || !qt) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    SICPCHECK(check_ready(h, false));
    const sicp_params& P = h->params;
    const bool em = P.mode == SICP_MODE_EM;
    Cloud &S = h->cloud(0), &T = h->cloud(1);
    SICPCHECK(prepare_cloud(h, S));
    SICPCHECK(prepare_cloud(h, T));
    if (!features_current(h, S, em)) SICPCHECK(compute_features(h, S, em));
    if (!features_current(h, T, em)) SICPCHECK(compute_features(h, T, em));
    if (em && !weights_from_histograms(P, P.knn)) {  // as align_begin: the projections belong to the feature phase
      SICPCHECK(ensure_proj(h, S));
      SICPCHECK(ensure_proj(h, T));
    }
    SICPCHECK(run_correspondences(h, qt, P.knn, true));
    const int n = S.n, K = P.knn;
    const size_t slots = (size_t)n * K;
    std::vector<int> hi(slots);
    std::vector<float> hd(slots);
    std::vector<double> hw(slots);
    if (slots) {
      HIPCHECK(hipMemcpyAsync(hi.data(), h->idx.p, sizeof(int) * slots, hipMemcpyDeviceToHost, h->stream));
      HIPCHECK(hipMemcpyAsync(hd.data(), h->d2.p, sizeof(float) * slots, hipMemcpyDeviceToHost, h->stream));
      if (h->corr_weighted) HIPCHECK(hipMemcpyAsync(hw.data(), h->w.p, sizeof(double) * slots, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (int i : S.drop_i)  // a non-finite source point has no correspondences
      for (int c = 0; c < K; ++c) {
        const size_t o = (size_t)i * K + c;
        if (idx) idx[o] = -1;
        if (d2) d2[o] = std::numeric_limits<float>::quiet_NaN();
        if (w) w[o] = 0.0;
      }
    for (int d = 0; d < n; ++d) {
      const int i = S.caller_index(d);
      for (int c = 0; c < K; ++c) {
        const size_t e = (size_t)d * K + c, o = (size_t)i * K + c;
        const int j = hi[e];
        if (idx) idx[o] = j < 0 ? -1 : T.caller_index(j);
        if (d2) d2[o] = hd[e];
        if (w) w[o] = j < 0 ? 0.0 : (h->corr_weighted ? hw[e] : 1.0);
      }
    }
    return SICP_OK;
  });
}

int sicp_accumulate(sicp_handle h, const double qt[7], double out28[28]) {
  return abi_guard(h, [&]() -> int {
    if (!h || !qt || !out28) return SICP_ERR_INVALID_ARGUMENT;
    if (!h->corr_valid) return SICP_ERR_NOT_READY;
    SICPCHECK(set_device(h));
    return eval28(h, qt, out28);
  });
}

int sicp_solve(sicp_handle h, const double init_qt[7], double out_qt[7], int32_t* lm_iters, int32_t* evals,
               double* final_cost) {
  return abi_guard(h, [&]() -> int {
    if (!h || !init_qt || !out_qt) return SICP_ERR_INVALID_ARGUMENT;
    if (!h->corr_valid) return SICP_ERR_NOT_READY;
    SICPCHECK(set_device(h));
    SolveResult r;
    SICPCHECK(run_solve(h, init_qt, out_qt, &r));
    if (lm_iters) *lm_iters = r.iterations;
    if (evals) *evals = r.evaluations;
    if (final_cost) *final_cost = r.cost;
    return SICP_OK;
  });
}

int sicp_fused_labels(sicp_handle h, const double qt[7], uint32_t* out_labels) {
  return abi_guard(h, [&]() -> int {
    if (!h || !qt || !out_labels) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    if (h->params.mode != SICP_MODE_EM) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(check_ready(h, true));
    Cloud &S = h->cloud(0), &T = h->cloud(1);
    if (T.n < 4) return SICP_ERR_TOO_FEW_POINTS;
    SICPCHECK(prepare_cloud(h, S));
    SICPCHECK(prepare_cloud(h, T));
    // getFusedLabels reuses what align() left behind (em_icp.hpp:230-241)
    if (!features_current(h, S, true)) SICPCHECK(compute_features(h, S, true));
    if (!features_current(h, T, true)) SICPCHECK(compute_features(h, T, true));
    SICPCHECK(labels_search(h, qt));
    SICPCHECK(labels_launch(h, qt, h->stream));
    std::vector<uint32_t> tmp(S.n);
    if (S.n > 0) HIPCHECK(hipMemcpyAsync(tmp.data(), h->tmpl.p, sizeof(uint32_t) * S.n, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (int i : S.drop_i) out_labels[i] = 0;  // no correspondences, no fused label (labels are 1-based: em_icp.hpp:265)
    for (int d = 0; d < S.n; ++d) out_labels[S.caller_index(d)] = tmp[d];
    return SICP_OK;
  });
}

int sicp_se3_device(sicp_handle h, int op, int32_t n, const double* in, double* out) {
  return abi_guard(h, [&]() -> int {
    if (!h || op < SICP_SE3_EXP || op > SICP_LM_SEQUENCE_ONE_LANE || n < 0 || (n > 0 && (!in || !out))) return SICP_ERR_INVALID_ARGUMENT;
    if (n == 0) return SICP_OK;
    SICPCHECK(set_device(h));
    const bool lm_seq = op >= SICP_LM_SEQUENCE;
    const size_t n_in = lm_seq ? (size_t)sicp::kLmSeqIn : op == SICP_SE3_EXP ? 6 : (op == SICP_SE3_PLUS ? 13 : (op == SICP_SE3_MUL ? 14 : 7));
    const size_t n_out = lm_seq ? (size_t)sicp::kLmSeqOut : op == SICP_SE3_LOG ? 6 : 7;
    DevBuf<double> d_in, d_out;
    HIPCHECK(d_in.reserve(n_in * n));
    HIPCHECK(d_out.reserve(n_out * n));
    HIPCHECK(hipMemcpyAsync(d_in.p, in, sizeof(double) * n_in * n, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(sicp::launch_se3_ops(op, n, d_in.p, d_out.p, h->stream));
    HIPCHECK(hipMemcpyAsync(out, d_out.p, sizeof(double) * n_out * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return SICP_OK;
  });
}

int sicp_get_stats(sicp_handle h, sicp_stats* stats) {
  return abi_guard(h, [&]() -> int {
    if (!h || !stats) return SICP_ERR_INVALID_ARGUMENT;
    *stats = h->st;
    return SICP_OK;
  });
}

int sicp_synchronize(sicp_handle h) {
  return abi_guard(h, [&]() -> int {
    if (!h) return SICP_ERR_INVALID_ARGUMENT;
    SICPCHECK(set_device(h));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return SICP_OK;
  });
}

}  // extern "C"
