// sicp_api.cpp -- implementation of the C ABI in include/sicp.h on top of the gfx950 kernels.
//
// Host side of the hot path: cloud upload (SoA), stage drivers, the outer ICP loops of the three
// reference classes (em_icp.hpp:25-200, gicp.hpp:29-175, semantic_icp.hpp:28-166), the 6-DoF LM
// driver (lm.hpp) and SE(3) (se3.hpp).  There is no CPU fallback: every stage runs on the GPU
// and every entry point fails with SICP_ERR_NO_DEVICE / SICP_ERR_HIP if it cannot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "build_tree.h"
#include "kernels.h"
#include "lm.hpp"
#include "se3.hpp"
#include "sicp.h"

namespace {

using sicp::se3::matrix34;

// ---- device memory arena -----------------------------------------------------------------------------
// hipMalloc costs 0.1 ... several ms and hipFree synchronises the device; a stream of registrations creates
// clouds and slot buffers all the time (a fresh cloud is ~26 buffers), on the very thread that feeds the GPU.
// Device buffers therefore come from a process-wide arena per device: slabs (64 MB doubling to 1 GB, or the
// request if larger) carved by a bump pointer into blocks of a few size classes (1/16 steps between powers of
// two: at most ~12 % slack); a released block goes to its class's free list and is handed out again for the
// same class.  Nothing is returned to the driver before sicp_release_pool, which frees the slabs of a device
// that hold no live block.  (Measured before: the align-only leg of an open stream took 0.4 or 1.3 s for the same
// 1024 registrations, depending on how the ~3000 hipMalloc calls inside it happened to go.)
constexpr int kArenaDevices = 64;
struct DevArena {
  struct Slab { char* base = nullptr; size_t size = 0, used = 0; long long live = 0; };
  struct Block { void* p; int slab; };
  struct Dev {
    std::vector<Slab> slabs;
    std::unordered_map<size_t, std::vector<Block>> free_by_class;
  };
  std::mutex m;
  Dev dev[kArenaDevices];
  static size_t size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    size_t p2 = 256;
    while (p2 < bytes) p2 <<= 1;
    const size_t step = std::max<size_t>(p2 >> 4, 256);
    return (bytes + step - 1) / step * step;
  }
  hipError_t alloc(size_t bytes, void** out, int* device, int* slab, size_t* cls_out) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    const size_t cls = size_class(bytes);
    std::lock_guard<std::mutex> lock(m);
    Dev& D = dev[d % kArenaDevices];
    auto it = D.free_by_class.find(cls);
    if (it != D.free_by_class.end() && !it->second.empty()) {
      const Block b = it->second.back();
      it->second.pop_back();
      D.slabs[b.slab].live++;
      *out = b.p; *device = d; *slab = b.slab; *cls_out = cls;
      return hipSuccess;
    }
    int k = -1;
    for (int i = (int)D.slabs.size() - 1; i >= 0 && i >= (int)D.slabs.size() - 4; --i)
      if (D.slabs[i].base && D.slabs[i].size - D.slabs[i].used >= cls) { k = i; break; }
    if (k < 0) {
      size_t want = (size_t)64 << 20;
      for (const Slab& sl : D.slabs) if (sl.base) want = std::min<size_t>(std::max(want, 2 * sl.size), (size_t)1 << 30);
      want = std::max(want, cls);
      Slab sl;
      e = hipMalloc((void**)&sl.base, want);
      if (e != hipSuccess && want > cls) { want = cls; e = hipMalloc((void**)&sl.base, want); }  // (memory is tight: the request alone)
      if (e != hipSuccess) return e;
      sl.size = want;
      k = -1;
      for (size_t i = 0; i < D.slabs.size(); ++i) if (!D.slabs[i].base) { k = (int)i; break; }  // (a slot freed by release)
      if (k < 0) { D.slabs.push_back(sl); k = (int)D.slabs.size() - 1; } else D.slabs[k] = sl;
    }
    Slab& S = D.slabs[k];
    *out = S.base + S.used;
    S.used += cls;
    S.live++;
    *device = d; *slab = k; *cls_out = cls;
    return hipSuccess;
  }
  // A block may be handed out again at once, to any thread and stream: like hipFree, giving one back first waits
  // for the device (launches that still read or write it may be in flight on streams the caller knows nothing of).
  // Releases are rare next to allocations: buffers that grow, handles and clouds (beyond the cloud pool) that go.
  void free(void* p, int device, int slab, size_t cls) {
    {
      int cur = -1;
      const bool ok = hipGetDevice(&cur) == hipSuccess;
      const bool switched = ok && cur != device && hipSetDevice(device) == hipSuccess;
      (void)hipDeviceSynchronize();
      if (switched) (void)hipSetDevice(cur);
    }
    std::lock_guard<std::mutex> lock(m);
    Dev& D = dev[device % kArenaDevices];
    D.free_by_class[cls].push_back(Block{p, slab});
    D.slabs[slab].live--;
  }
  // frees the slabs of `device` that hold no live block (the current device must be `device`)
  void release(int device) {
    std::lock_guard<std::mutex> lock(m);
    Dev& D = dev[device % kArenaDevices];
    for (size_t i = 0; i < D.slabs.size(); ++i) {
      Slab& S = D.slabs[i];
      if (!S.base || S.live != 0) continue;
      for (auto& kv : D.free_by_class) {
        std::vector<Block>& v = kv.second;
        v.erase(std::remove_if(v.begin(), v.end(), [&](const Block& b) { return b.slab == (int)i; }), v.end());
      }
      (void)hipFree(S.base);
      S = Slab();
    }
  }
};
DevArena& dev_arena() {
  static DevArena* a = new DevArena;  // never destroyed: it may outlive the HIP runtime at process exit
  return *a;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int dev_ = -1, slab_ = -1;
  size_t cls_ = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) dev_arena().free(p, dev_, slab_, cls_);
    p = nullptr;
    cap = 0;
  }
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    release();
    // 64 elements of slack beyond the capacity: kernels that read whole vectors may touch up to one
    // vector past the last element (the values are never used)
    size_t want = n + n / 8;
    void* q = nullptr;
    hipError_t e = dev_arena().alloc((want + 64) * sizeof(T), &q, &dev_, &slab_, &cls_);
    if (e != hipSuccess) { p = nullptr; return e; }
    p = static_cast<T*>(q);
    cap = want;
    return hipSuccess;
  }
};

// pinned host memory: uploads and read-backs through it are real asynchronous DMA copies
template <class T>
struct HostBuf {
  T* p = nullptr;
  size_t cap = 0, n = 0;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { if (p) (void)hipHostFree(p); }
  hipError_t resize(size_t count) {
    if (count > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      const size_t want = count + count / 8 + 64;
      hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
      if (e != hipSuccess) { p = nullptr; n = 0; return e; }
      cap = want;
    }
    n = count;
    return hipSuccess;
  }
  hipError_t assign(const T* src, size_t count) {
    hipError_t e = resize(count);
    if (e == hipSuccess && count) std::memcpy(p, src, count * sizeof(T));
    return e;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

struct Cloud {
  int n = 0;         // points on the device (the finite ones: what the search index holds)
  int n_caller = 0;  // points the caller handed over (what every per-point output is sized by)
  // Non-finite points are left out of the device cloud, as pcl::KdTreeFLANN::setInputCloud leaves them
  // out of its index (em_icp.h:50-66): keep[i] = caller index of device-side input point i (empty when
  // nothing was dropped), drop_i / drop_xyz = the dropped points themselves (for the final_cloud output).
  std::vector<int> keep, drop_i;
  std::vector<float> drop_xyz;
  bool is_set = false, has_label = false;
  float bb_lo[3] = {0, 0, 0}, bb_hi[3] = {0, 0, 0};  // bounding box of the staged (finite) points
  bool bb_valid = false;
  HostBuf<float> hx, hy, hz;  // caller order (pinned: the staging buffers of the upload)
  HostBuf<uint32_t> hl;
  uint32_t label_min = 0, label_max = 0;  // of hl (EM labels are validated against 1..C)
  // device layout: one segment (GICP / EM) or one segment per label in first-seen order
  // (SEMANTIC); inside a segment the points are in Morton order
  int layout = -1;        // -1 none, 0 flat, 1 grouped
  HostBuf<int> perm;  // device index -> caller index
  std::vector<uint32_t> seg_label;
  std::vector<int> seg_off;  // n_seg + 1
  DevBuf<float> x, y, z;
  DevBuf<uint32_t> label;
  // search structure (bvh.hpp): packed points (x, y, z, caller index), boxes, seed tables
  struct SegTree {
    sicp::TreeLevels lv;
    int n, pt_begin, node_begin, code_begin;
    float lo[3], scale;
  };
  std::vector<SegTree> trees;
  DevBuf<float4> pts4, box_lo, box_hi;
  DevBuf<unsigned long long> leaf_code;
  DevBuf<int> inv;  // caller index -> device index
  // build scratch: the cloud as the caller gave it, sort buffers
  DevBuf<float> rx, ry, rz;
  DevBuf<uint32_t> rl;
  DevBuf<int> ids, d_perm, vals_in, vals_out;
  HostBuf<int> h_ids;
  DevBuf<unsigned long long> keys_in, keys_out;
  DevBuf<unsigned char> sort_temp;
  DevBuf<sicp::PointRec> rec;  // position + normal of every point (what the weight / accumulate kernels gather)
  DevBuf<char> rec_dense;      // the same as three dense arrays (what the accumulate kernel streams for the source points)
  int rec_dense_n = 0;         // the cloud size they were written for (0: not written)
  DevBuf<uint8_t> hist;
  DevBuf<double> proj;  // [n][proj_stride(C)] label distribution x confusion matrix
  bool proj_valid = false;
  DevBuf<int> nn;
  int nn_stride = 0;  // 0: [n][k]; > 0: [k][nn_stride]
  bool feat_valid = false;
  int feat_k = 0, feat_C = 0, feat_float_products = 0;
  bool feat_hist = false;
  // which align() / align_batch() call computed the features last (a cloud shared by two handles of
  // one batch is only searched once per call), and which confusion matrix the projections belong to
  unsigned long long feat_epoch = 0;
  unsigned long long proj_cm_id = 0;
  // the upload + tree build is left running on the uploading handle's stream: whoever uses the cloud
  // next (any handle, any stream, or the host reading `perm`) waits for this event first
  hipEvent_t ready_ev = nullptr;
  // set by the uploading thread, cleared by whoever waits first (a sequence driver uploads the next
  // batch's scans on a second host thread while the main thread registers clouds that share them)
  std::atomic<bool> pending{false};
  ~Cloud() { if (ready_ev) (void)hipEventDestroy(ready_ev); }
  int n_seg() const { return (int)seg_label.size(); }
  int caller_index(int d) const { return keep.empty() ? perm[d] : keep[perm[d]]; }
};

// Clouds (with all their device and pinned buffers) are recycled through a per-device pool: a scan
// sequence uploads a new cloud per registration, and allocating / freeing ~25 buffers each time would
// serialise the pipeline (hipFree synchronises the device).  The pool is never destroyed (it may
// outlive the HIP runtime at process exit); sicp_release_pool frees what it holds.
constexpr int kPoolDevices = 64;
// parked clouds per device beyond which a released cloud is freed instead (two batches of 256 pairs with
// their own source and target clouds fit; ~11 MB of HBM and ~2 MB of pinned memory per 100K-point cloud)
constexpr size_t kPoolCap = 1024;
struct CloudPool {
  std::mutex m;
  std::vector<Cloud*> free_list[kPoolDevices];
};
CloudPool& cloud_pool() {
  static CloudPool* pool = new CloudPool;
  return *pool;
}

std::shared_ptr<Cloud> acquire_cloud(int device) {
  CloudPool& pool = cloud_pool();
  const int slot = device % kPoolDevices;
  Cloud* c = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool.m);
    if (!pool.free_list[slot].empty()) { c = pool.free_list[slot].back(); pool.free_list[slot].pop_back(); }
  }
  if (!c) c = new Cloud();
  return std::shared_ptr<Cloud>(c, [slot, device](Cloud* dead) {
    // (a stream's cloud may be dropped without any handle having waited for its upload on the host: settle it
    // while the uploading stream still exists -- see settle_cloud)
    if (dead->pending && dead->ready_ev) (void)hipEventSynchronize(dead->ready_ev);
    dead->pending = false;
    dead->n = 0; dead->n_caller = 0; dead->is_set = false; dead->has_label = false; dead->layout = -1;
    dead->keep.clear(); dead->drop_i.clear(); dead->drop_xyz.clear();
    dead->feat_valid = false; dead->proj_valid = false; dead->feat_epoch = 0; dead->proj_cm_id = 0;
    CloudPool& pl = cloud_pool();
    {
      std::lock_guard<std::mutex> lock(pl.m);
      if (pl.free_list[slot].size() < kPoolCap) { pl.free_list[slot].push_back(dead); return; }
    }
    // the pool is full: free this one (hipFree synchronises the device -- only beyond the cap)
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    delete dead;
    if (switched) (void)hipSetDevice(cur);
  });
}

unsigned long long next_epoch() {
  static std::atomic<unsigned long long> counter{0};
  return ++counter;
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// lock-step batch: instead of launching, the per-pair stages append their jobs here; the batch driver
// launches each kind once for all pairs (kernels.h: *_jobs launchers), in dependency order
constexpr int kParts = 4;  // slices of a batch whose stage sequences run on their own streams

struct JobCollector {
  int knn_K[kParts] = {};  // list length of a slice's searches (one launch per slice: one length)
  int slice = 0;  // slice of the batch the pair whose stage is running belongs to (set by the driver)
  std::vector<sicp::KnnArgs> knn[kParts];
  std::vector<sicp::CovArgs> cov[kParts];
  std::vector<sicp::ProjArgs> proj[kParts];
  std::vector<sicp::WeightArgs> weight[kParts];
  std::vector<sicp::CountJob> count[kParts];
};

// the argument buffers of one stream of ticks (run_tick): argument array + header in HBM with pinned
// mirrors, and the instantiated [accumulate, LM step] x lm_batch graph that reads them
struct TickSet {
  DevBuf<sicp::BatchArgs> d_batch;
  DevBuf<sicp::BatchHeader> d_bhdr;
  sicp::BatchHeader* h_bhdr = nullptr;
  sicp::BatchArgs* h_batch = nullptr;
  sicp::LmJoin* h_join = nullptr;  // pinned: the pairs that join with the next tick
  DevBuf<sicp::LmJoin> d_join;
  int cap = 0;
  sicp::BatchGraph graph;
  std::vector<int> tick_act;  // the pairs whose arguments d_batch currently holds
  bool tick_valid = false;
};

struct sicp_context {
  int device = 0;
  JobCollector* collect = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // second cloud's feature kernels run beside the first's
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_join = nullptr;
  sicp_params params;
  std::shared_ptr<Cloud> cl[2];
  Cloud& cloud(int which) { return *cl[which]; }
  const Cloud& cloud(int which) const { return *cl[which]; }
  unsigned long long epoch = 0;  // id of the running align() / align_batch() call
  int C = 0;
  std::vector<double> cm;
  unsigned long long cm_id = 0;  // changes with every sicp_set_confusion
  DevBuf<double> d_cm, d_hval;
  int hval_k = 0;
  // correspondences of the last search
  DevBuf<int> idx;
  DevBuf<float> d2;
  DevBuf<double> w;
  int corr_n = 0, corr_K = 0;
  bool corr_valid = false, corr_weighted = false;
  bool hint_ok = false;  // idx holds this align()'s previous search: usable as the next search's seed hint
  DevBuf<unsigned long long> part;
  DevBuf<double> partials, out28;
  DevBuf<long long> d_count;
  DevBuf<sicp::LmState> d_lm;
  // one batch of the device-resident solve ([accumulate, lm_step] x lm_batch) captured as a graph:
  // a single launch call per batch instead of 2 x lm_batch trips through the runtime's launch path
  sicp::LmState* h_lm = nullptr;  // pinned mirror of the device-resident LM state
  double* h_out28 = nullptr;      // pinned, 28 doubles
  long long* h_count = nullptr;   // pinned
  DevBuf<float> tmpx, tmpy, tmpz;
  DevBuf<uint32_t> tmpl;
  // lock-step batch (sicp_align_batch), owned by the batch's first handle: one BatchArgs and one LM
  // state per pair, pinned mirrors, and the captured [accumulate_batch, lm_step_batch] x lm_batch graph
  TickSet ts[2];  // two sets: the halves of a batch alternate, one's tick runs while the host turns the other around
  DevBuf<sicp::LmState> d_bstates;
  DevBuf<unsigned> d_solo_sync;       // the last pair still iterating: hand-off words of the persistent solve (solve_one_kernel)
  unsigned solo_tag = 0;              // its tags so far (a launch uses solo_tag + 1 ...: the words are never zeroed in between)
  int solo_seq = 0, solo_pair = 0;    // launch counter (the state's pad_ word echoes it at a regular end) and the pair's state slot
  bool solo_was_init = false, solo_failed = false;  // the launch in flight starts a solve / the last one did not run to its end
  int solo_skip = 0, solo_penalty = 0;  // after a persistent launch timed out: solves that stay with the tick graph before the next try (doubling)
  bool count_stats = false;           // the align() in progress reports statistics: every search also counts its live slots
  bool counted_in_search = false;     // ... and the search kernel of the current correspondences did so itself
  DevBuf<double> d_bout28;
  sicp::LmState* h_bstates = nullptr;
  double* h_bout28 = nullptr;
  int h_batch_cap = 0;  // capacity of the per-pair state mirrors (h_bstates, h_bout28)
  hipStream_t side_stream = nullptr;  // batch leader: searches of the pairs between two inner solves
  hipEvent_t side_done = nullptr, side_done2 = nullptr, main_done = nullptr;
  hipStream_t feat_stream = nullptr;          // batch leader: the start-up pipelines (features + first search) of a large batch
  std::vector<hipEvent_t> chunk_ev;           // one per start-up chunk
  hipStream_t part_stream[kParts] = {};
  hipEvent_t part_fork = nullptr, part_done[kParts] = {};
  // a slot of a registration stream: an upload that is still in flight (queued by the submitting thread on
  // the stream's upload stream) is waited for ON THE DEVICE, by the stream the slot's kernels run on
  bool wait_on_device = false;
  std::string last_error;
  sicp_stats st;
};

// ---- a registration stream (sicp_stream_*): the continuous batching of sicp_align_batch without the closed
// batch.  Clouds are uploaded by the submitting thread on the stream's own upload stream; a worker thread owns
// `cap` handles (slots) and runs the tick loop: admit queued registrations into free slots, one turn, retire.
struct StreamCloudRef;
struct sicp_stream_ctx {
  int device = 0, cap = 0;
  sicp_params params;
  int C = 0;
  std::vector<double> cm;
  std::vector<sicp_context*> slots;      // slots[0] leads: tick sets, LM states, side stream
  std::vector<hipStream_t> own1, own2;   // the slots' own streams (restored before the handles are destroyed)
  sicp_context* uploader = nullptr;      // runs the uploads + search-tree builds (caller's thread, own stream)
  std::mutex up_m;                       // one upload at a time
  // ---- shared between the caller's threads and the worker, under `m`
  std::mutex m;
  std::condition_variable cv_work, cv_done, cv_space;
  struct Submission {
    long long ticket;
    std::shared_ptr<Cloud> src, tgt;
    double init[7];
  };
  std::deque<Submission> queue;
  std::deque<sicp_stream_result> done;
  std::unordered_map<long long, std::shared_ptr<Cloud>> clouds;
  long long next_cloud = 1, next_ticket = 1;
  long long submitted = 0, completed = 0, busy_evals = 0, slot_evals = 0;
  int draining = 0;  // callers blocked in sicp_stream_poll(wait >= 2): nothing new will be submitted by them meanwhile
  int in_flight = 0;
  bool stop = false;
  int error = 0;
  std::string error_msg;
  // ---- worker only
  std::vector<long long> slot_ticket;
  std::vector<double> slot_t0;
  std::thread worker;
};

namespace {

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      h->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return SICP_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define SICPCHECK(expr)          \
  do {                           \
    int _s = (expr);             \
    if (_s != SICP_OK) return _s; \
  } while (0)

struct KernelTimer {
  // brackets a group of launches with events when params.profile is on
  sicp_context* h;
  bool on;
  KernelTimer(sicp_context* ctx, int bit) : h(ctx), on((ctx->params.profile & bit) != 0) {
    if (on) (void)hipEventRecord(h->ev0, h->stream);
  }
  // returns elapsed ms (synchronises the stream up to here); 0 when profiling is off
  double stop() {
    if (!on) return 0.0;
    float ms = 0.f;
    if (hipEventRecord(h->ev1, h->stream) != hipSuccess) return 0.0;
    if (hipEventSynchronize(h->ev1) != hipSuccess) return 0.0;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return 0.0;
    return (double)ms;
  }
};

int set_device(sicp_context* h) {
  HIPCHECK(hipSetDevice(h->device));
  return SICP_OK;
}

// ---- cloud layout -----------------------------------------------------------------------------
// A cloud's upload is recorded in an event on the uploading handle's stream, and the cloud may outlive
// that handle (it is shared, or goes back to the pool).  Waiting for an event whose stream has been
// destroyed is not safe with this runtime (it intermittently answers "event last recorded in a
// capturing stream"), so a handle settles every cloud it lets go of while its streams still exist.
void settle_cloud(Cloud& c) {
  if (c.pending && c.ready_ev) (void)hipEventSynchronize(c.ready_ev);
  c.pending = false;
}

int cloud_wait(sicp_context* h, Cloud& c) {
  if (c.pending && h->wait_on_device) {  // (the flag stays up: whoever needs the host copy of `perm` still waits on the host)
    HIPCHECK(hipStreamWaitEvent(h->stream, c.ready_ev, 0));
    return SICP_OK;
  }
  if (c.pending) {
    HIPCHECK(hipEventSynchronize(c.ready_ev));
    c.pending = false;
  }
  return SICP_OK;
}

// host side of an upload: the caller's arrays -> the cloud's pinned staging buffers.  The caller's layout is a base
// pointer per coordinate and one byte stride (SoA: three arrays, stride 4; a pcl::PointXYZL array: one base + 0 / 4 /
// 8, stride 32), labels likewise.  ONE pass over the cloud: finite test, copy, bounding box, label range (a scan
// sequence stages a cloud per registration on the thread that submits them: five passes were 0.3 ms per 100K points).
struct StridedCloud {
  const char *x, *y, *z, *label;  // label may be null
  long long stride, label_stride;
};
int stage_cloud(sicp_context* h, Cloud& c, int32_t n, const StridedCloud& in) {
  if ((long long)n > ((long long)sicp::kLeaf << (2 * (sicp::kMaxLevels - 1)))) {
    h->last_error = "cloud too large for the search tree (16 * 4^11 = 67 M points per cloud)";
    return SICP_ERR_INVALID_ARGUMENT;
  }
  SICPCHECK(cloud_wait(h, c));  // a previous upload may still be reading the staging buffers
  // Non-finite points (the NaNs of an organized RGB-D cloud) never enter the device cloud:
  // pcl::KdTreeFLANN::setInputCloud (em_icp.h:50-66) leaves them out of the search index, so the
  // reference can neither find them as neighbours nor -- a NaN query keeps no candidate -- match them.
  // Everything below works on the finite points; outputs are mapped back to the caller's indices.
  c.n_caller = n;
  c.keep.clear(); c.drop_i.clear(); c.drop_xyz.clear();
  c.has_label = in.label != nullptr;
  HIPCHECK(c.hx.resize(n)); HIPCHECK(c.hy.resize(n)); HIPCHECK(c.hz.resize(n)); HIPCHECK(c.hl.resize(in.label ? n : 0));
  const float inf = std::numeric_limits<float>::infinity();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  uint32_t lmin = 0xffffffffu, lmax = 0;
  auto ld = [](const char* base, long long stride, int i) { float v; std::memcpy(&v, base + (long long)i * stride, sizeof v); return v; };
  int m = 0;  // finite points so far: they are stored compacted as they come
  float* const hx = c.hx.data(); float* const hy = c.hy.data(); float* const hz = c.hz.data();
  uint32_t* const hl = c.hl.data();
  for (int i = 0; i < n; ++i) {
    const float px = ld(in.x, in.stride, i), py = ld(in.y, in.stride, i), pz = ld(in.z, in.stride, i);
    // (x - x is 0 for a finite x and NaN otherwise: one test for the three coordinates)
    const float t = (px - px) + (py - py) + (pz - pz);
    if (t == 0.f) {
      hx[m] = px; hy[m] = py; hz[m] = pz;
      lo[0] = px < lo[0] ? px : lo[0]; hi[0] = px > hi[0] ? px : hi[0];
      lo[1] = py < lo[1] ? py : lo[1]; hi[1] = py > hi[1] ? py : hi[1];
      lo[2] = pz < lo[2] ? pz : lo[2]; hi[2] = pz > hi[2] ? pz : hi[2];
      if (in.label) {
        uint32_t lb; std::memcpy(&lb, in.label + (long long)i * in.label_stride, sizeof lb);
        hl[m] = lb;
        lmin = lb < lmin ? lb : lmin; lmax = lb > lmax ? lb : lmax;
      }
      if (m != i) c.keep.push_back(i);  // (only once a point has been dropped; completed below)
      ++m;
    } else {
      if (c.keep.empty() && c.drop_i.empty()) {  // the first dropped point: the kept ones so far map to themselves
        c.keep.reserve(n);
        for (int k = 0; k < m; ++k) c.keep.push_back(k);
      }
      c.drop_i.push_back(i);
      c.drop_xyz.push_back(px); c.drop_xyz.push_back(py); c.drop_xyz.push_back(pz);
    }
  }
  if (m != n) {  // (sizes follow the finite points; keep[] has one entry per kept point)
    HIPCHECK(c.hx.resize(m)); HIPCHECK(c.hy.resize(m)); HIPCHECK(c.hz.resize(m)); HIPCHECK(c.hl.resize(in.label ? m : 0));
  }
  c.n = m;
  c.label_min = lmin; c.label_max = lmax;
  for (int d = 0; d < 3; ++d) { c.bb_lo[d] = lo[d]; c.bb_hi[d] = hi[d]; }
  c.bb_valid = true;
  c.is_set = true;
  c.layout = -1;
  c.feat_valid = false;
  c.proj_valid = false;
  return SICP_OK;
}
int stage_cloud(sicp_context* h, Cloud& c, int32_t n, const float* x, const float* y, const float* z, const uint32_t* label) {
  const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
  return stage_cloud(h, c, n, in);
}

int prepare_cloud(sicp_context* h, Cloud& c) {
  const int want = h->params.mode == SICP_MODE_SEMANTIC ? 1 : 0;
  if (!c.is_set) return SICP_ERR_NOT_READY;
  if (want == 1 && !c.has_label) return SICP_ERR_NOT_READY;
  SICPCHECK(cloud_wait(h, c));  // an upload still in flight (possibly queued by another handle or host thread)
  if (c.layout == want) return SICP_OK;
  const int n = c.n;
  // ---- host: segment membership and per-segment bounding boxes (one pass over the cloud)
  c.seg_label.clear();
  std::vector<int> which(want ? n : 0), counts;
  if (want == 0) {
    c.seg_label.push_back(0);
    counts.push_back(n);
  } else {
    // pcl_2_semantic.h:24-39: one sub-cloud per label, labels in order of first appearance
    for (int i = 0; i < n; ++i) {
      int sidx = -1;
      for (size_t k = 0; k < c.seg_label.size(); ++k)
        if (c.seg_label[k] == c.hl[i]) { sidx = (int)k; break; }
      if (sidx < 0) { sidx = (int)c.seg_label.size(); c.seg_label.push_back(c.hl[i]); counts.push_back(0); }
      which[i] = sidx;
      counts[sidx]++;
    }
  }
  const int n_seg = (int)c.seg_label.size();
  const float inf = std::numeric_limits<float>::infinity();
  std::vector<float> lo(3 * n_seg, inf), hi(3 * n_seg, -inf);
  if (!want && c.bb_valid)  // one segment: its box came with the staging pass
    for (int d = 0; d < 3; ++d) { lo[d] = c.bb_lo[d]; hi[d] = c.bb_hi[d]; }
  for (int i = 0; i < ((!want && c.bb_valid) ? 0 : n); ++i) {
    const int sg = want ? which[i] : 0;
    const float p[3] = {c.hx[i], c.hy[i], c.hz[i]};
    for (int d = 0; d < 3; ++d) {
      if (p[d] < lo[3 * sg + d]) lo[3 * sg + d] = p[d];
      if (p[d] > hi[3 * sg + d]) hi[3 * sg + d] = p[d];
    }
  }
  c.seg_off.assign(n_seg + 1, 0);
  c.trees.assign(n_seg, Cloud::SegTree());
  std::vector<sicp::BuildSegment> segs(n_seg);
  int pt_total = 0, node_total = 0, code_total = 0, max_cnt = 1;
  for (int sg = 0; sg < n_seg; ++sg) {
    sicp::BuildSegment& g = segs[sg];
    g.off = c.seg_off[sg]; g.cnt = counts[sg];
    c.seg_off[sg + 1] = g.off + g.cnt;
    g.lv = sicp::make_levels(g.cnt);
    g.padded = g.lv.cnt[0] * sicp::kLeaf;  // every leaf of the complete tree owns 16 point slots (sentinels beyond the real points)
    g.pt_begin = pt_total; g.node_begin = node_total; g.code_begin = code_total;
    pt_total += g.padded; node_total += sicp::total_nodes(g.lv); code_total += g.lv.cnt[0];
    float ext = 0.f;
    for (int d = 0; d < 3; ++d) { g.lo[d] = g.cnt > 0 ? lo[3 * sg + d] : 0.f; if (g.cnt > 0) ext = std::max(ext, hi[3 * sg + d] - lo[3 * sg + d]); }
    if (!(ext > 0.f) || !std::isfinite(ext)) ext = 1.f;
    g.scale = 2097151.f / ext;
    max_cnt = std::max(max_cnt, g.cnt);
    Cloud::SegTree& st = c.trees[sg];
    st.lv = g.lv; st.n = g.cnt; st.pt_begin = g.pt_begin; st.node_begin = g.node_begin; st.code_begin = g.code_begin;
    st.lo[0] = g.lo[0]; st.lo[1] = g.lo[1]; st.lo[2] = g.lo[2]; st.scale = g.scale;
  }
  HostBuf<int>& ids = c.h_ids;
  if (want) {  // caller indices grouped by segment, cloud order inside a segment
    HIPCHECK(ids.resize(n));
    std::vector<int> fill(c.seg_off.begin(), c.seg_off.end() - 1);
    for (int i = 0; i < n; ++i) ids[fill[which[i]]++] = i;
  }
  // ---- device: upload the caller-order cloud, build curve order + boxes (build_tree.hip)
  const size_t m = (size_t)(n > 0 ? n : 1);
  HIPCHECK(c.rx.reserve(m)); HIPCHECK(c.ry.reserve(m)); HIPCHECK(c.rz.reserve(m)); HIPCHECK(c.rl.reserve(m));
  HIPCHECK(c.ids.reserve(m)); HIPCHECK(c.d_perm.reserve(m));
  HIPCHECK(c.keys_in.reserve((size_t)max_cnt)); HIPCHECK(c.keys_out.reserve((size_t)max_cnt));
  HIPCHECK(c.vals_in.reserve((size_t)max_cnt)); HIPCHECK(c.vals_out.reserve((size_t)max_cnt));
  const size_t temp_bytes = sicp::build_sort_temp_bytes(max_cnt);
  HIPCHECK(c.sort_temp.reserve(temp_bytes + 256));
  HIPCHECK(c.x.reserve(m)); HIPCHECK(c.y.reserve(m)); HIPCHECK(c.z.reserve(m));
  HIPCHECK(c.label.reserve(m)); HIPCHECK(c.inv.reserve(m));
  HIPCHECK(c.pts4.reserve((size_t)pt_total + 1)); HIPCHECK(c.box_lo.reserve((size_t)node_total + 1));
  HIPCHECK(c.box_hi.reserve((size_t)node_total + 1)); HIPCHECK(c.leaf_code.reserve((size_t)code_total + 1));
  auto up = [&](void* dst, const void* src, size_t bytes) {
    return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
  };
  HIPCHECK(up(c.rx.p, c.hx.data(), sizeof(float) * n));
  HIPCHECK(up(c.ry.p, c.hy.data(), sizeof(float) * n));
  HIPCHECK(up(c.rz.p, c.hz.data(), sizeof(float) * n));
  if (c.has_label) HIPCHECK(up(c.rl.p, c.hl.data(), sizeof(uint32_t) * n));
  if (want) HIPCHECK(up(c.ids.p, ids.data(), sizeof(int) * n));
  sicp::BuildBuffers b;
  b.rx = c.rx.p; b.ry = c.ry.p; b.rz = c.rz.p; b.rl = c.has_label ? c.rl.p : nullptr; b.ids = want ? c.ids.p : nullptr;
  b.keys_in = c.keys_in.p; b.keys_out = c.keys_out.p; b.vals_in = c.vals_in.p; b.vals_out = c.vals_out.p;
  b.sort_temp = c.sort_temp.p; b.sort_temp_bytes = temp_bytes;
  b.x = c.x.p; b.y = c.y.p; b.z = c.z.p; b.label = c.label.p; b.perm = c.d_perm.p; b.inv = c.inv.p;
  b.pts4 = c.pts4.p; b.box_lo = c.box_lo.p; b.box_hi = c.box_hi.p; b.leaf_code = c.leaf_code.p;
  HIPCHECK(sicp::build_tree_device(b, segs.data(), n_seg, h->stream));
  HIPCHECK(c.perm.resize(n));  // device -> caller order, for returning results in the caller's order
  if (n > 0) HIPCHECK(hipMemcpyAsync(c.perm.data(), c.d_perm.p, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
  // no synchronisation here: every staging buffer is pinned memory owned by the cloud, and the next
  // user of the cloud waits for ready_ev (cloud_wait).  A sequence driver can therefore queue the
  // uploads of a whole batch of scans back to back, beside the registrations of the previous batch.
  if (!c.ready_ev) HIPCHECK(hipEventCreateWithFlags(&c.ready_ev, hipEventDisableTiming));
  HIPCHECK(hipEventRecord(c.ready_ev, h->stream));
  c.pending = true;
  c.layout = want;
  c.feat_valid = false;
  h->corr_valid = false;
  h->hint_ok = false;
  return SICP_OK;
}

// queries: points [q_begin, q_begin+q_count) of cloud Q (device order), optionally transformed
// by M34; targets: segment `tseg` of cloud T.  Writes device indices of T (or -1) and distances.
int run_nn(sicp_context* h, int K, const Cloud& Qc, int q_begin, int q_count, const double* M34, const Cloud& Tc,
           int tseg, bool self, float gate_sq, int* out_i, float* out_d, int timer_bit, hipStream_t stream, int out_stride = 0) {
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
    // seed hint: what the previous search of the same queries found (same clouds, same K, this align)
    a.seed_hint = (!self && h->hint_ok && out_i == h->idx.p && h->corr_K == K && h->corr_n == Qc.n) ? h->idx.p : nullptr;
    a.hint_K = K;
    a.t_begin = Tc.seg_off.empty() ? 0 : Tc.seg_off[tseg];
    static const bool want_dbg = std::getenv("SICP_KNN_STATS") != nullptr;  // developer aid, off by default
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

int compute_features(sicp_context* h, Cloud& c, bool with_hist, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  const sicp_params& P = h->params;
  const int k = P.k_cov, n = c.n;
  const size_t m = (size_t)(n > 0 ? n : 1);
  HIPCHECK(c.rec.reserve(m));
  // an empty cloud still has one (all-zero) record: the accumulate kernel evaluates dead slots on record 0
  // and weights them by exactly zero, which needs finite values there
  if (n == 0) HIPCHECK(hipMemsetAsync(c.rec.p, 0, sizeof(sicp::PointRec), stream));
  HIPCHECK(c.nn.reserve(m * k));
  if (with_hist) HIPCHECK(c.hist.reserve(m * P.num_classes));
  // the packet search writes the lists rank-major ([k][n]): coalesced stores there and coalesced
  // loads in the covariance kernel; the other engines keep [n][k]
  static const bool no_lane_per_query = std::getenv("SICP_KNN_LANE_PER_QUERY") == nullptr && std::getenv("SICP_KNN_STATS") == nullptr;
  const int nn_stride = (P.nn_method == 1 && no_lane_per_query) ? (int)m : 0;
  for (int s = 0; s < c.n_seg(); ++s) {
    const int o = c.seg_off[s], cnt = c.seg_off[s + 1] - o;
    SICPCHECK(run_nn(h, k, c, o, cnt, nullptr, c, s, true, std::numeric_limits<float>::infinity(), c.nn.p, nullptr,
                     SICP_PROFILE_COV, stream, nn_stride));
  }
  c.nn_stride = nn_stride;
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
      SICPCHECK(run_nn(h, K, S, so, sn, M, T, ts, false, (float)P.gate_sq, h->idx.p, h->d2.p, SICP_PROFILE_NN, h->stream));
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
  if (weights) SICPCHECK(run_weights(h, qt));
  return SICP_OK;
}

// the EM weights of the current correspondences (em_icp.hpp:62-107): what run_correspondences(..., true) ends with
int run_weights(sicp_context* h, const double* qt) {
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  const int K = h->corr_K;
  if (P.mode == SICP_MODE_EM) {
    KernelTimer kt(h, SICP_PROFILE_WEIGHT);
    const double t0 = now_ms();
    sicp::WeightArgs a;
    a.n_s = S.n; a.K = K; a.C = P.num_classes;
    a.idx = h->idx.p;
    a.srec = S.rec.p; a.trec = T.rec.p;
    SICPCHECK(ensure_proj(h, S));
    SICPCHECK(ensure_proj(h, T));
    a.s_proj = S.proj.p; a.t_proj = T.proj.p;
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
int batch_reserve(sicp_context* h, int n);
int run_tick(sicp_context* h, hipStream_t M, sicp_handle* hs, int n, const std::vector<int>& act, const std::vector<int>& joining,
             const double (*start)[7], int len, int solo_evals);

// One evaluation sweep at pose qt: the batched kernel on a batch of one (every path -- a pair alone, a
// lock-step batch, the host-loop solve, this hook -- runs the SAME accumulate kernel, so they agree bit
// for bit), then the fixed-order sum of the chunk partials.
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
  *h->ts[0].h_bhdr = sicp::BatchHeader{1, {0, 0, 0}};
  HIPCHECK(hipMemcpyAsync(h->ts[0].d_bhdr.p, h->ts[0].h_bhdr, sizeof(sicp::BatchHeader), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->ts[0].d_batch.p, h->ts[0].h_batch, sizeof(sicp::BatchArgs), hipMemcpyHostToDevice, h->stream));
  {
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

struct SolveResult {
  int status = 0, iterations = 0, evaluations = 0;
  double cost = 0;
};

// may the next launch of this leader be a persistent one?  (after a timed-out launch a number of them are not)
bool solo_allowed(sicp_context* h) {
  if (h->solo_skip > 0) { --h->solo_skip; return false; }
  return true;
}

// the inner ceres::Solve (em_icp.hpp:162-177) on the current correspondences
int run_solve(sicp_context* h, const double* init_qt, double* out_qt, SolveResult* res) {
  const sicp_params& P = h->params;
  if (!P.lm_on_device) {
    // host loop: one kernel pair + one 224-byte read-back + one synchronisation per evaluation
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

// ---- pieces of align() shared by the single-pair and the lock-step batch drivers ---------------
struct OuterState {
  double cur[7], est[7];
  int outer = 0, count = 0;
  bool converged = false;
};

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
  if (em) {  // label distributions through the confusion matrix: same phase as the features they read
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
  HIPCHECK(h->d_bout28.reserve((size_t)28 * n));
  if (h->h_batch_cap < n) {
    if (h->h_bstates) (void)hipHostFree(h->h_bstates);
    if (h->h_bout28) (void)hipHostFree(h->h_bout28);
    h->h_bstates = nullptr; h->h_bout28 = nullptr; h->h_batch_cap = 0;
    HIPCHECK(hipHostMalloc((void**)&h->h_bstates, sizeof(sicp::LmState) * n, hipHostMallocDefault));
    HIPCHECK(hipHostMalloc((void**)&h->h_bout28, sizeof(double) * 28 * n, hipHostMallocDefault));
    h->h_batch_cap = n;
  }
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
int flush_jobs(sicp_context* h, JobCollector& jc, hipStream_t base = nullptr) {
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

// while a lock-step batch runs, all its handles work on the leader's stream and collect their jobs
struct BatchGuard {
  sicp_handle* hs; int n;
  std::vector<hipStream_t> s1, s2;
  BatchGuard(sicp_handle* handles, int count, JobCollector* jc, hipStream_t stream) : hs(handles), n(count), s1(count), s2(count) {
    for (int p = 0; p < n; ++p) {
      s1[p] = hs[p]->stream; s2[p] = hs[p]->stream2;
      if (jc) { hs[p]->collect = jc; hs[p]->stream = stream; hs[p]->stream2 = stream; }
    }
  }
  // from here on the pairs' own launches (memsets, searches outside the job lists) go to `stream`
  void retarget(hipStream_t stream) {
    for (int p = 0; p < n; ++p) { hs[p]->stream = stream; hs[p]->stream2 = stream; }
  }
  ~BatchGuard() {
    for (int p = 0; p < n; ++p) { hs[p]->collect = nullptr; hs[p]->stream = s1[p]; hs[p]->stream2 = s2[p]; }
  }
};

// One TICK of a batch: `len` LM evaluations of every pair in `act` (pair indices), in one graph launch:
// the accumulate kernel evaluates all of them at their current LM poses, lm_step_batch_kernel
// advances every pair's trust-region machine (csrc/lm.hpp, the same code and the same bits as for a
// pair alone).  `joining` pairs start their inner solve with this tick (their LM state is initialised
// and uploaded first).  tick_launch only queues work on stream M (ending with the read-back of the
// states of pairs [lo, hi) into h_bstates); the caller synchronises M when it wants the result.
int tick_launch(sicp_context* h, TickSet& S, hipStream_t M, sicp_handle* hs, int lo, int hi, const std::vector<int>& act,
                const std::vector<int>& joining, const double (*start)[7], int len, int solo_evals = 0) {
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
    h->solo_pair = p;
    h->solo_was_init = A.init != 0;
    S.tick_valid = false;  // (the argument array in HBM was not refreshed)
    HIPCHECK(sicp::launch_solve_one(g->corr_K, h->params.use_sqloss, A, nb, M));
    HIPCHECK(hipMemcpyAsync(h->h_bstates + p, h->d_bstates.p + p, sizeof(sicp::LmState), hipMemcpyDeviceToHost, M));
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
    B.nb = nb;
  }
  if (!same_set) {
    *S.h_bhdr = sicp::BatchHeader{(int)act.size(), {0, 0, 0}};
    HIPCHECK(hipMemcpyAsync(S.d_bhdr.p, S.h_bhdr, sizeof(sicp::BatchHeader), hipMemcpyHostToDevice, M));
    HIPCHECK(hipMemcpyAsync(S.d_batch.p, S.h_batch, sizeof(sicp::BatchArgs) * act.size(), hipMemcpyHostToDevice, M));
    S.tick_act = act;
    S.tick_valid = true;
  }
  // [accumulate, lm_step_batch] x len as an explicit graph with fixed grids: the kernels read the
  // number of pairs from the header and their status from the LM states, so the graph is instantiated once per tick set
  // (buffer addresses) and never touched when pairs come and go or batches differ in size.
  {
    int built = 0;
    HIPCHECK(sicp::batch_graph_prepare(S.graph, hs[act[0]]->corr_K, h->params.use_sqloss, S.d_bhdr.p, S.d_batch.p, std::min(S.cap, kMaxActivePairs),
                                       len, &built));
    h->st.graph_builds += built;
  }
  HIPCHECK(hipGraphLaunch(S.graph.exec, M));
  HIPCHECK(hipMemcpyAsync(h->h_bstates + lo, h->d_bstates.p + lo, sizeof(sicp::LmState) * (hi - lo), hipMemcpyDeviceToHost, M));
  return SICP_OK;
}

// Wait for the work queued on M so far.  (Polling with hipStreamQuery before blocking, to shorten the
// wake-up of the host thread, measured no different: 3.76 vs 3.78 ms for one pair alone.)
int tick_wait(sicp_context* h, hipStream_t M) {
  HIPCHECK(hipStreamSynchronize(M));
  return SICP_OK;
}

// after a persistent launch has been waited for: did it run to its regular end?
int solo_check(sicp_context* h) {
#if defined(SICP_SOLO_TIMING)  // developer aid: cycles per phase of the master and of worker 0, per evaluation of the launch that just ended
  {
    unsigned w[14];
    if (hipMemcpy(w, h->d_solo_sync.p + sicp::kSoloSyncWords, sizeof w, hipMemcpyDeviceToHost) == hipSuccess) {
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
  static const bool log = std::getenv("SICP_SOLO_LOG") != nullptr;  // developer aid
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
  SICPCHECK(tick_wait(h, M));
  if (solo_evals > 0) SICPCHECK(solo_check(h));
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


// ---- continuous batching: what sicp_align_batch (a closed set of pairs) and sicp_stream_* (pairs that come
// and go) share.  Every pair runs its own sequence
//     search (transform + kNN + weights) -> inner solve -> convergence test -> search -> ...
// and the run advances in TICKS of `len` LM evaluations: one graph launch evaluates every pair that is
// inside an inner solve, while the searches of the pairs that have just finished one run on a second
// stream beside it; those pairs rejoin at the next tick.  No pair waits for another pair's solve or outer
// loop -- only for the end of the current tick.
// PAIR_FIRST: the pair's start-up pipeline (features, first search, weights) is queued on the start-up stream; it joins
// the ticks when the event of its chunk has completed
enum { PAIR_FREE = -1, PAIR_NEED_SEARCH = 0, PAIR_JOINING, PAIR_SOLVING, PAIR_DONE, PAIR_FIRST };

// pairs [lo, hi) that advance together: one tick stream, one argument set
struct TickGroup {
  int lo = 0, hi = 0;
  hipStream_t M = nullptr;
  TickSet* S = nullptr;
  hipEvent_t side_done = nullptr;
  bool pending = false, side_recorded = false;
  int round = 0;
  std::vector<int> act, joining, finished;
};

struct BatchRun {
  sicp_context* L = nullptr;    // leader: owns the tick sets, the LM states and the side stream
  sicp_context** hs = nullptr;  // slot -> handle
  int n = 0;                    // slots
  sicp_params P;                // what every pair of the run agrees on (same_solver)
  int len = 8;                  // LM evaluations per tick
  bool one_launch = true, want_stats = false;
  bool solo = false;            // the last pair still iterating may run its solve as persistent launches (lm_on_device != 2)
  hipStream_t side = nullptr;   // searches / features of the pairs between two inner solves
  struct Start { double q[7]; };
  std::vector<OuterState> o;
  std::vector<int> phase, search_round;
  std::vector<Start> starts;
  double dbg_wait_ms = 0, dbg_search_ms = 0, dbg_launch_ms = 0; long long dbg_ticks = 0, dbg_act = 0;  // developer aid (SICP_STREAM_LOG)
  bool solo_now = false;        // the tick in flight is a persistent solve
  std::vector<int> evals_seen;  // evaluations of the pair's running solve already counted in the statistics
  std::vector<int> first_chunk;        // PAIR_FIRST: the start-up chunk the pair belongs to
  std::vector<hipEvent_t> chunk_ev;    // recorded behind each chunk's start-up pipeline
  // pairs whose start-up pipeline has completed join the ticks; with `block` the host waits for the first chunk
  // that is still running (nothing else is left to do)
  int promote_started(const TickGroup& G, bool block) {
    int waiting = 0, promoted = 0;
    for (int pass = 0; pass < 2; ++pass) {
      waiting = promoted = 0;
      int first_unfinished = -1;
      for (int p = G.lo; p < G.hi; ++p) {
        if (phase[p] != PAIR_FIRST) continue;
        const hipError_t q = hipEventQuery(chunk_ev[first_chunk[p]]);
        if (q == hipSuccess) { phase[p] = PAIR_JOINING; search_round[p] = 0; ++promoted; }
        else if (q == hipErrorNotReady) { ++waiting; if (first_unfinished < 0) first_unfinished = first_chunk[p]; }
        else return -1;
      }
      if (promoted || !block || first_unfinished < 0) break;
      if (hipEventSynchronize(chunk_ev[first_unfinished]) != hipSuccess) return -1;
    }
    return waiting;
  }
  void resize(int slots) {
    n = slots;
    o.assign(slots, OuterState());
    phase.assign(slots, PAIR_FREE);
    search_round.assign(slots, 0);
    starts.assign(slots, Start());
    first_chunk.assign(slots, 0);
    evals_seen.assign(slots, 0);
  }
  // pair p starts its align() at init_qt (its handle's align_begin has run)
  void start_pair(int p, const double* init_qt) {
    o[p] = OuterState();
    std::memcpy(o[p].cur, init_qt, sizeof o[p].cur);
    phase[p] = PAIR_NEED_SEARCH;
    search_round[p] = 0;
  }
  int live(const TickGroup& G) const {
    int k = 0;
    for (int p = G.lo; p < G.hi; ++p) k += phase[p] == PAIR_NEED_SEARCH || phase[p] == PAIR_JOINING || phase[p] == PAIR_SOLVING || phase[p] == PAIR_FIRST;
    return k;
  }
  int turn(TickGroup& G, JobCollector& jc);
};

// What the host does between two ticks of group G: finish the previous tick (if any), queue the searches
// of the pairs that are between two solves, and launch the next tick.
int BatchRun::turn(TickGroup& G, JobCollector& jc) {
  sicp_context* h = L;
  if (G.pending) {
    const double tw0 = now_ms();
    SICPCHECK(tick_wait(h, G.M));
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
  // this round's searches run beside the tick
  if (any_search && one_launch) {
    SICPCHECK(flush_jobs(h, jc, side));
    HIPCHECK(hipEventRecord(G.side_done, side));
    G.side_recorded = true;
  }
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
  int rc = tick_launch(h, *G.S, G.M, hs, G.lo, G.hi, G.act, G.joining, reinterpret_cast<const double(*)[7]>(starts.data()), len,
                       solo_now ? (n == 1 ? sicp::kSoloMaxEvals : 64) : 0);
  dbg_launch_ms += now_ms() - dbg_t_launch0;
  if (rc != SICP_OK) return rc;
  G.pending = true;
  return SICP_OK;
}

}  // namespace

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
    default: return "unknown status";
  }
}

const char* sicp_last_error(sicp_handle h) { return h ? h->last_error.c_str() : ""; }

int sicp_device_count(int* count) {
  if (!count) return SICP_ERR_INVALID_ARGUMENT;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return SICP_ERR_NO_DEVICE; }
  *count = n;
  return SICP_OK;
}

int sicp_default_params(int mode, sicp_params* p) {
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
}

int sicp_create(int device_id, sicp_handle* out) {
  if (!out) return SICP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SICP_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return SICP_ERR_INVALID_ARGUMENT;
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
    sicp_destroy(h);
    return SICP_ERR_NO_DEVICE;
  }
  *out = h;
  return SICP_OK;
}

int sicp_destroy(sicp_handle h) {
  if (!h) return SICP_OK;
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
    sicp::batch_graph_destroy(S.graph);
  }
  if (h->h_bstates) (void)hipHostFree(h->h_bstates);
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
  delete h;
  return SICP_OK;
}

int sicp_release_pool(int device_id) {
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
  for (Cloud* c : dead) delete c;
  (void)hipDeviceSynchronize();   // nothing may still be running out of a block that goes back to the driver
  dev_arena().release(device_id);  // the slabs no live buffer is carved from
  return SICP_OK;
}

int sicp_set_params(sicp_handle h, const sicp_params* p) {
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
}

int sicp_get_params(sicp_handle h, sicp_params* p) {
  if (!h || !p) return SICP_ERR_INVALID_ARGUMENT;
  *p = h->params;
  return SICP_OK;
}

// The buffers a cloud's features will need, taken from the arena when the cloud is SET rather than at its first
// align(): a new arena slab is a hipMalloc of up to 1 GB, which the driver clears before handing it out (~30 ms per
// GB) -- inside a stream that is the worker's turn, i.e. every registration in flight waits (measured: the resident
// leg of the open-stream bench took 1.28 instead of 0.45 s when its 1025 clouds' 27 GB of feature buffers were first
// touched inside the timed region).
static int reserve_features(sicp_context* h, Cloud& c) {
  const sicp_params& P = h->params;
  const size_t m = (size_t)(c.n > 0 ? c.n : 1);
  HIPCHECK(c.rec.reserve(m));
  HIPCHECK(c.nn.reserve(m * (size_t)(P.k_cov > 0 ? P.k_cov : 1)));
  HIPCHECK(c.rec_dense.reserve(sicp::dense_rec_bytes(c.n)));
  if (P.mode == SICP_MODE_EM && P.num_classes > 0) {
    HIPCHECK(c.hist.reserve(m * (size_t)P.num_classes));
    HIPCHECK(c.proj.reserve(m * (size_t)sicp::proj_stride(P.num_classes)));
  }
  return SICP_OK;
}

static int set_cloud_common(sicp_handle h, int which, int32_t n, const StridedCloud& in) {
  SICPCHECK(set_device(h));
  if (h->cl[which].use_count() > 1) {  // shared with another handle: leave theirs alone
    settle_cloud(*h->cl[which]);
    h->cl[which] = acquire_cloud(h->device);
  }
  Cloud& c = h->cloud(which);
  SICPCHECK(stage_cloud(h, c, n, in));
  c.is_set = true;
  c.layout = -1;
  c.feat_valid = false;
  h->corr_valid = false;
  h->hint_ok = false;
  // upload now for the current mode, so that align() starts with the cloud resident in HBM;
  // a later mode change re-lays it out lazily
  if (h->params.mode != SICP_MODE_SEMANTIC || c.has_label) {
    SICPCHECK(prepare_cloud(h, c));
    SICPCHECK(reserve_features(h, c));
  }
  return SICP_OK;
}

int sicp_set_cloud(sicp_handle h, int which, int32_t n, const float* x, const float* y, const float* z,
                   const uint32_t* label) {
  if (!h || (which != SICP_SOURCE && which != SICP_TARGET) || n < 0) return SICP_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!x || !y || !z)) return SICP_ERR_INVALID_ARGUMENT;
  const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
  return set_cloud_common(h, which, n, in);
}

int sicp_set_cloud_strided(sicp_handle h, int which, int32_t n, const void* xyz, int64_t stride_bytes, const void* label,
                           int64_t label_stride_bytes) {
  if (!h || (which != SICP_SOURCE && which != SICP_TARGET) || n < 0) return SICP_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!xyz || stride_bytes < 12 || (label && label_stride_bytes < 4))) return SICP_ERR_INVALID_ARGUMENT;
  const char* b = (const char*)xyz;
  const StridedCloud in = {b, b + 4, b + 8, (const char*)label, stride_bytes, label_stride_bytes};
  return set_cloud_common(h, which, n, in);
}

int sicp_set_cloud_device(sicp_handle h, int which, int32_t n, const float* xd, const float* yd, const float* zd,
                          const uint32_t* ld) {
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
}

int sicp_share_cloud(sicp_handle h, int which, sicp_handle from, int from_which) {
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
}

int sicp_cloud_size(sicp_handle h, int which, int32_t* n_points, int32_t* n_indexed) {
  if (!h || (which != SICP_SOURCE && which != SICP_TARGET)) return SICP_ERR_INVALID_ARGUMENT;
  const Cloud& c = h->cloud(which);
  if (!c.is_set) return SICP_ERR_NOT_READY;
  if (n_points) *n_points = c.n_caller;
  if (n_indexed) *n_indexed = c.n;
  return SICP_OK;
}

int sicp_set_confusion(sicp_handle h, int32_t C, const double* cm) {
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
}

int sicp_align(sicp_handle h, const double init_qt[7], double out_qt[7], int32_t* outer_iters, sicp_stats* stats) {
  if (!h || !init_qt || !out_qt) return SICP_ERR_INVALID_ARGUMENT;
  SICPCHECK(set_device(h));
  SICPCHECK(check_ready(h, false));
  const sicp_params& P = h->params;
  // device-resident solve: a lock-step batch of one (the same kernels, hence the same bits, as any batch)
  if (P.lm_on_device) return sicp_align_batch(&h, 1, init_qt, out_qt, outer_iters, stats);
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

int sicp_align_batch(sicp_handle* hs, int32_t n, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats) {
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
      HIPCHECK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
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

int sicp_accumulate_batch(sicp_handle* hs, int32_t n, const double* qt, double* out28, int32_t repeat, double* kernel_ms) {
  if (!hs || n < 1 || !qt || !out28) return SICP_ERR_INVALID_ARGUMENT;
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
  *h->ts[0].h_bhdr = sicp::BatchHeader{n, {0, 0, 0}};
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
}

// =================================================================================================
// registration streams
// =================================================================================================
}  // extern "C"

namespace {

void stream_fail(sicp_stream_ctx* S, int rc, const std::string& msg) {
  std::lock_guard<std::mutex> lock(S->m);
  if (S->error == SICP_OK) { S->error = rc; S->error_msg = msg; }
  S->cv_done.notify_all();
  S->cv_space.notify_all();
}

// The worker: admit queued registrations into free slots, one turn of the continuous batching
// (BatchRun::turn: finish the tick in flight, queue the searches of the pairs between two solves, launch
// the next tick), retire the pairs that have converged.  One iteration per tick.
void stream_worker(sicp_stream_ctx* S) {
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
  std::vector<sicp_stream_result> out;
  std::vector<std::array<double, 11>> dbg_log;
  double dbg_admit_ms = 0, dbg_flush_ms = 0, dbg_turn_ms = 0;
  for (;;) {
    // ---- admit
    fresh.clear(); fresh_slot.clear();
    {
      std::unique_lock<std::mutex> lock(S->m);
      S->cv_work.wait(lock, [&] { return S->stop || !S->queue.empty() || S->in_flight > 0; });
      if (S->stop) {
        for (size_t i = 0; i < dbg_log.size(); i += std::max<size_t>(1, dbg_log.size() / 40))
          std::fprintf(stderr, "[stream] t %.1f ms completed %.0f ticks %.0f waited %.1f ms pairs-per-tick %.1f solo-allowed %.0f | host ms: admit %.1f flush %.1f turn %.1f (of which waited; searches %.1f, tick launch %.1f)\n",
                       dbg_log[i][0], dbg_log[i][1], dbg_log[i][2], dbg_log[i][3], dbg_log[i][4], dbg_log[i][5], dbg_log[i][6], dbg_log[i][7], dbg_log[i][8], dbg_log[i][9], dbg_log[i][10]);
        return;
      }
      while (!S->queue.empty() && !free_slots.empty()) {
        fresh.push_back(std::move(S->queue.front()));
        S->queue.pop_front();
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
      jc.slice = batch_slice(p, S->cap, S->params.knn);
      int rc = check_ready(h, false);
      if (rc == SICP_OK) rc = align_begin(h, false);
      if (rc != SICP_OK) {  // this registration cannot run (too few points, bad labels ...): report it, free the slot
        sicp_stream_result r;
        std::memset(&r, 0, sizeof r);
        r.ticket = fresh[k].ticket; r.status = rc;
        std::memcpy(r.qt, fresh[k].init, sizeof r.qt);
        out.push_back(r);
        free_slots.push_back(p);
        if (rc == SICP_ERR_HIP) { stream_fail(S, rc, h->last_error); return; }
        continue;
      }
      run.start_pair(p, fresh[k].init);
    }
    // the new pairs' features (self-searches, covariances, projections): one launch per kind, on the side
    // stream, beside the tick in flight and ahead of the pairs' first searches
    const double t_flush0 = now_ms();
    dbg_admit_ms += t_flush0 - t_admit0;
    if (!fresh.empty()) {
      const int rc = flush_jobs(L, jc, run.side);
      if (rc != SICP_OK) { stream_fail(S, rc, L->last_error); return; }
    }
    const double t_turn0 = now_ms();
    dbg_flush_ms += t_turn0 - t_flush0;
    // ---- one turn
    {
      const int rc = run.turn(G, jc);
      if (rc != SICP_OK) { stream_fail(S, rc, L->last_error); return; }
    }
    dbg_turn_ms += now_ms() - t_turn0;
    // ---- retire
    long long busy = 0, slots_sat = 0;
    for (int p = 0; p < S->cap; ++p) {
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
    }
    {
      static const bool slog = std::getenv("SICP_STREAM_LOG") != nullptr;  // developer aid: kept in memory, printed when the stream ends
      if (slog && !out.empty())
        dbg_log.push_back({now_ms(), (double)(S->completed + (long long)out.size()), (double)run.dbg_ticks, run.dbg_wait_ms,
                           run.dbg_ticks ? (double)run.dbg_act / run.dbg_ticks : 0.0, (double)run.solo, dbg_admit_ms, dbg_flush_ms, dbg_turn_ms, run.dbg_search_ms, run.dbg_launch_ms});
    }
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

}  // namespace

extern "C" {

int sicp_stream_create(int device_id, const sicp_params* params, int32_t max_in_flight, sicp_stream* out) {
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
      if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&h->side_done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->main_done, hipEventDisableTiming) != hipSuccess)
        return cleanup(SICP_ERR_HIP);
    }
    // every slot's own launches (memsets of the semantic search, cloud waits) go to the side stream
    for (sicp_context* g : S->slots) { g->stream = h->side_stream; g->stream2 = h->side_stream; g->wait_on_device = true; }
  }
  S->slot_ticket.assign(S->cap, 0);
  S->slot_t0.assign(S->cap, 0.0);
  S->worker = std::thread(stream_worker, S.get());
  *out = S.release();
  return SICP_OK;
}

int sicp_stream_destroy(sicp_stream S) {
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
  S->clouds.clear();
  S->queue.clear();
  if (S->uploader) sicp_destroy(S->uploader);
  delete S;
  return SICP_OK;
}

const char* sicp_stream_last_error(sicp_stream S) { return S ? S->error_msg.c_str() : ""; }

int sicp_stream_set_confusion(sicp_stream S, int32_t C, const double* cm) {
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
  std::lock_guard<std::mutex> lock(S->m);
  const long long id = S->next_cloud++;
  S->clouds.emplace(id, std::move(c));
  *cloud_id = id;
  return SICP_OK;
}

int sicp_stream_add_cloud(sicp_stream S, int32_t n, const float* x, const float* y, const float* z, const uint32_t* label, int64_t* cloud_id) {
  if (!S || !cloud_id || n < 0 || (n > 0 && (!x || !y || !z))) return SICP_ERR_INVALID_ARGUMENT;
  const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
  return stream_add_common(S, n, in, cloud_id);
}

int sicp_stream_add_cloud_strided(sicp_stream S, int32_t n, const void* xyz, int64_t stride_bytes, const void* label, int64_t label_stride_bytes,
                                  int64_t* cloud_id) {
  if (!S || !cloud_id || n < 0) return SICP_ERR_INVALID_ARGUMENT;
  if (n > 0 && (!xyz || stride_bytes < 12 || (label && label_stride_bytes < 4))) return SICP_ERR_INVALID_ARGUMENT;
  const char* b = (const char*)xyz;
  const StridedCloud in = {b, b + 4, b + 8, (const char*)label, stride_bytes, label_stride_bytes};
  return stream_add_common(S, n, in, cloud_id);
}

int sicp_stream_release_cloud(sicp_stream S, int64_t cloud_id) {
  if (!S) return SICP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(S->m);
  return S->clouds.erase(cloud_id) ? SICP_OK : SICP_ERR_INVALID_ARGUMENT;
}

int sicp_stream_submit(sicp_stream S, int64_t source_id, int64_t target_id, const double init_qt[7], int64_t* ticket) {
  if (!S || !init_qt) return SICP_ERR_INVALID_ARGUMENT;
  std::unique_lock<std::mutex> lock(S->m);
  if (S->error != SICP_OK) return S->error;
  auto a = S->clouds.find(source_id), b = S->clouds.find(target_id);
  if (a == S->clouds.end() || b == S->clouds.end()) return SICP_ERR_INVALID_ARGUMENT;
  S->cv_space.wait(lock, [&] { return S->stop || S->error != SICP_OK || (int)S->queue.size() < S->cap; });
  if (S->error != SICP_OK) return S->error;
  if (S->stop) return SICP_ERR_INVALID_ARGUMENT;
  sicp_stream_ctx::Submission q;
  q.ticket = S->next_ticket++;
  q.src = a->second; q.tgt = b->second;
  std::memcpy(q.init, init_qt, sizeof q.init);
  S->queue.push_back(std::move(q));
  ++S->submitted;
  if (ticket) *ticket = S->next_ticket - 1;
  S->cv_work.notify_all();
  return SICP_OK;
}

int sicp_stream_poll(sicp_stream S, int32_t wait, int32_t max_results, sicp_stream_result* results, int32_t* n_results) {
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
}

int sicp_stream_counters(sicp_stream S, int64_t* submitted, int64_t* completed, int64_t* busy_evals, int64_t* slot_evals) {
  if (!S) return SICP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(S->m);
  if (submitted) *submitted = S->submitted;
  if (completed) *completed = S->completed;
  if (busy_evals) *busy_evals = S->busy_evals;
  if (slot_evals) *slot_evals = S->slot_evals;
  return SICP_OK;
}

int sicp_search_batch(sicp_handle* hs, int32_t n, const double* qt, int32_t what, int32_t use_hint, int32_t repeat, double* kernel_ms) {
  if (!hs || n < 1 || what < 0 || what > 2 || (what == 0 && !qt)) return SICP_ERR_INVALID_ARGUMENT;
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
      HIPCHECK(hipStreamSynchronize(h->stream));
    }
  }
  // the handles' own stage drivers collect the jobs exactly as sicp_align_batch makes them
  JobCollector jc;
  {
    BatchGuard guard(hs, n, &jc, h->stream);
    for (int p = 0; p < n; ++p) {
      sicp_context* g = hs[p];
      jc.slice = 0;
      if (what == 0) {
        if (!use_hint) g->hint_ok = false;
        const int rc = run_correspondences(g, qt + 7 * p, g->params.knn, false);
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
}

int sicp_transform_source(sicp_handle h, const double qt[7], float* ox, float* oy, float* oz) {
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
}

int sicp_covariances(sicp_handle h, int which, double* cov9, double* normal3, uint8_t* hist, int32_t* nn_idx) {
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
  std::vector<sicp::PointRec> rec(n);
  if (n > 0) HIPCHECK(hipMemcpyAsync(rec.data(), c.rec.p, sizeof(sicp::PointRec) * n, hipMemcpyDeviceToHost, h->stream));
  std::vector<uint8_t> hh;
  std::vector<int> nn;
  if (hist && n > 0) {
    hh.resize((size_t)n * P.num_classes);
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
    if (normal3) { normal3[3 * (size_t)i] = v[0]; normal3[3 * (size_t)i + 1] = v[1]; normal3[3 * (size_t)i + 2] = v[2]; }
    if (cov9)  // the covariance the kernels use: C = I - (1-eps) n n^T  (== em_icp.hpp:331-338)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) cov9[9 * (size_t)i + 3 * a + b] = (a == b ? 1.0 : 0.0) - ome * v[a] * v[b];
    if (hist) std::memcpy(hist + (size_t)i * P.num_classes, hh.data() + (size_t)d * P.num_classes, P.num_classes);
    if (nn_idx)
      for (int j = 0; j < k; ++j) {
        const int g = c.nn_stride > 0 ? nn[(size_t)j * c.nn_stride + d] : nn[(size_t)d * k + j];
        nn_idx[(size_t)i * k + j] = g < 0 ? -1 : c.caller_index(g);
      }
  }
  return SICP_OK;
}

int sicp_correspondences(sicp_handle h, const double qt[7], int32_t* idx, float* d2, double* w) {
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
}

int sicp_accumulate(sicp_handle h, const double qt[7], double out28[28]) {
  if (!h || !qt || !out28) return SICP_ERR_INVALID_ARGUMENT;
  if (!h->corr_valid) return SICP_ERR_NOT_READY;
  SICPCHECK(set_device(h));
  return eval28(h, qt, out28);
}

int sicp_solve(sicp_handle h, const double init_qt[7], double out_qt[7], int32_t* lm_iters, int32_t* evals,
               double* final_cost) {
  if (!h || !init_qt || !out_qt) return SICP_ERR_INVALID_ARGUMENT;
  if (!h->corr_valid) return SICP_ERR_NOT_READY;
  SICPCHECK(set_device(h));
  SolveResult r;
  SICPCHECK(run_solve(h, init_qt, out_qt, &r));
  if (lm_iters) *lm_iters = r.iterations;
  if (evals) *evals = r.evaluations;
  if (final_cost) *final_cost = r.cost;
  return SICP_OK;
}

int sicp_fused_labels(sicp_handle h, const double qt[7], uint32_t* out_labels) {
  if (!h || !qt || !out_labels) return SICP_ERR_INVALID_ARGUMENT;
  SICPCHECK(set_device(h));
  if (h->params.mode != SICP_MODE_EM) return SICP_ERR_INVALID_ARGUMENT;
  SICPCHECK(check_ready(h, true));
  const sicp_params& P = h->params;
  Cloud &S = h->cloud(0), &T = h->cloud(1);
  if (T.n < 4) return SICP_ERR_TOO_FEW_POINTS;
  SICPCHECK(prepare_cloud(h, S));
  SICPCHECK(prepare_cloud(h, T));
  // getFusedLabels reuses what align() left behind (em_icp.hpp:230-241)
  if (!features_current(h, S, true)) SICPCHECK(compute_features(h, S, true));
  if (!features_current(h, T, true)) SICPCHECK(compute_features(h, T, true));
  SICPCHECK(run_correspondences(h, qt, 4, false));  // K = 4 is a literal here (em_icp.hpp:221)
  h->corr_valid = false;                            // K may differ from params.knn
  h->hint_ok = false;
  sicp::WeightArgs a;
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
  HIPCHECK(sicp::launch_fused_labels(a, h->tmpl.p, h->stream));
  std::vector<uint32_t> tmp(S.n);
  if (S.n > 0) HIPCHECK(hipMemcpyAsync(tmp.data(), h->tmpl.p, sizeof(uint32_t) * S.n, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  for (int i : S.drop_i) out_labels[i] = 0;  // no correspondences, no fused label (labels are 1-based: em_icp.hpp:265)
  for (int d = 0; d < S.n; ++d) out_labels[S.caller_index(d)] = tmp[d];
  return SICP_OK;
}

int sicp_se3_device(sicp_handle h, int op, int32_t n, const double* in, double* out) {
  if (!h || op < SICP_SE3_EXP || op > SICP_SE3_INV || n < 0 || (n > 0 && (!in || !out))) return SICP_ERR_INVALID_ARGUMENT;
  if (n == 0) return SICP_OK;
  SICPCHECK(set_device(h));
  const size_t n_in = op == SICP_SE3_EXP ? 6 : (op == SICP_SE3_PLUS ? 13 : (op == SICP_SE3_MUL ? 14 : 7));
  const size_t n_out = op == SICP_SE3_LOG ? 6 : 7;
  DevBuf<double> d_in, d_out;
  HIPCHECK(d_in.reserve(n_in * n));
  HIPCHECK(d_out.reserve(n_out * n));
  HIPCHECK(hipMemcpyAsync(d_in.p, in, sizeof(double) * n_in * n, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(sicp::launch_se3_ops(op, n, d_in.p, d_out.p, h->stream));
  HIPCHECK(hipMemcpyAsync(out, d_out.p, sizeof(double) * n_out * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return SICP_OK;
}

int sicp_get_stats(sicp_handle h, sicp_stats* stats) {
  if (!h || !stats) return SICP_ERR_INVALID_ARGUMENT;
  *stats = h->st;
  return SICP_OK;
}

int sicp_synchronize(sicp_handle h) {
  if (!h) return SICP_ERR_INVALID_ARGUMENT;
  SICPCHECK(set_device(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return SICP_OK;
}

}  // extern "C"
