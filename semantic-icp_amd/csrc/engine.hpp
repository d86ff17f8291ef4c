// engine.hpp -- internal header of the host side of libsicp.so (never installed; include/sicp.h is the boundary).
//
// The host side is split along its seams:
//   memory.cpp    device arena, cloud pool
//   clouds.cpp    staging, upload + search-tree build of a cloud
//   stages.cpp    stage drivers of one align(): searches, covariances, projections, weights, one evaluation
//   solve.cpp     inner solve: ticks, the persistent launch, continuous batching (BatchRun), sicp_align_batch
//   streams.cpp   registration streams (worker thread + sicp_stream_*)
//   sicp_api.cpp  the remaining C-ABI entry points
// Every extern "C" entry runs inside abi_guard (abi_barrier.hpp): no exception crosses the boundary.
#ifndef SICP_ENGINE_HPP_
#define SICP_ENGINE_HPP_

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
#include <unordered_set>
#include <vector>

#include "abi_barrier.hpp"
#include "build_tree.h"
#include "kernels.h"
#include "lm.hpp"
#include "se3.hpp"
#include "sicp.h"

namespace sicp {
namespace host {

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
    size_t reserved = 0;  // bytes of all slabs (what the arena holds of the device's memory)
    size_t limit = 0;     // sicp_set_memory_limit: no new slab beyond this many bytes (0 = none)
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
      if (D.limit && D.reserved + want > D.limit) want = cls;  // (near the caller's limit: the request alone)
      if (D.limit && D.reserved + want > D.limit) return hipErrorOutOfMemory;
      Slab sl;
      e = hipMalloc((void**)&sl.base, want);
      if (e != hipSuccess && want > cls) { want = cls; e = hipMalloc((void**)&sl.base, want); }  // (memory is tight: the request alone)
      if (e != hipSuccess) return e;
      sl.size = want;
      D.reserved += want;
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
  // An object that gives back MANY blocks at once (a handle: ~40, a cloud beyond the pool's cap: ~26) waits for the
  // device ONCE: FreeScope does the wait, and the frees of this thread inside it skip theirs -- the blocks are the
  // dying object's own, nothing can be launched on them any more.
  static int& scope_device() { static thread_local int d = -1; return d; }
  struct FreeScope {
    int prev;
    explicit FreeScope(int device) : prev(scope_device()) {
      int cur = -1;
      const bool ok = hipGetDevice(&cur) == hipSuccess;
      const bool switched = ok && cur != device && hipSetDevice(device) == hipSuccess;
      (void)hipDeviceSynchronize();
      if (switched) (void)hipSetDevice(cur);
      scope_device() = device;
    }
    ~FreeScope() { scope_device() = prev; }
    FreeScope(const FreeScope&) = delete;
    FreeScope& operator=(const FreeScope&) = delete;
  };
  void free(void* p, int device, int slab, size_t cls) {
    if (scope_device() != device) {
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
      D.reserved -= S.size;
      S = Slab();
    }
  }
};
DevArena& dev_arena();  // memory.cpp; never destroyed: it may outlive the HIP runtime at process exit

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
  // several label segments: their descriptions and sort offsets for the one-launch-per-stage build (build_tree.hip)
  DevBuf<sicp::BuildSegmentDev> d_segs;
  HostBuf<sicp::BuildSegmentDev> h_segs;
  DevBuf<int> d_seg_begin, d_seg_end;
  HostBuf<int> h_seg_begin, h_seg_end;
  DevBuf<sicp::PointRec> rec;  // position + normal of every point (what the weight / accumulate kernels gather)
  DevBuf<char> rec_dense;      // the same as three dense arrays (what the accumulate kernel streams for the source points)
  int rec_dense_n = 0;         // the cloud size they were written for (0: not written)
  // Caller-supplied covariances that are NOT of the form I - (1-eps) n n^T (sicp_set_covariances): the six entries xx xy xz yy
  // yz zz of every point's symmetric matrix, device order.  A handle with such a cloud evaluates through
  // accumulate_general_kernel (the literal gicp_cost_function.h:27-73 with full 3x3 matrices), one pair at a time.
  DevBuf<double> cov6;
  bool cov_general = false;
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
CloudPool& cloud_pool();
std::shared_ptr<Cloud> acquire_cloud(int device);
unsigned long long next_epoch();
double now_ms();
// developer logs (SICP_KNN_STATS, SICP_SOLO_LOG, SICP_STREAM_LOG, -DSICP_SOLO_TIMING) print to stderr ONLY when
// SICP_DEBUG is set in the environment: without it the library never prints (include/sicp.h)
bool debug_enabled();

// lock-step batch: instead of launching, the per-pair stages append their jobs here; the batch driver
// launches each kind once for all pairs (kernels.h: *_jobs launchers), in dependency order
constexpr int kParts = 4;  // slices of a batch whose stage sequences run on their own streams

struct JobCollector {
  // EM weights written by the search's own epilogue (stages.cpp: run_correspondences) instead of a weight kernel behind it:
  // set by a batch of at most 4 pairs, where every launch is on the critical path of an align() (one pair alone: -46 us of
  // 1.98 ms).  Not by larger batches and streams: there the separate weight kernel runs hidden beside the accumulate launches
  // and the longer search does not (measured, profiles/r05/weights_in_search_epilogue.json: 256-pair step 184.4 against 185.3 ms).
  bool fold_weights = false;
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
  sicp::BatchGraph graph[2];   // [1]: the accumulate nodes carry kAccStaticRanges (static_ranges below)
  bool static_ranges = false;  // the next ticks' large accumulate launches keep equal chunk ranges (a stream while scans are being uploaded)
  std::vector<int> tick_act;  // the pairs whose arguments d_batch currently holds
  bool tick_valid = false;
  unsigned epoch_host = 0;    // mirror of d_bhdr->epoch_base: tick_prepare_kernel adds kMaxBatchLen per tick, and so does the host
};

}  // namespace host
}  // namespace sicp

using namespace sicp::host;  // (internal header: only the library's own translation units include it)

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
  DevBuf<double> tmp9;  // sicp_covariances: the 3x3 matrices in the caller's order on their way out
  DevBuf<uint32_t> tmpl;
  HostBuf<uint32_t> h_labels;  // pinned: fused labels of a stream slot on their way back
  // lock-step batch (sicp_align_batch), owned by the batch's first handle: one BatchArgs and one LM
  // state per pair, pinned mirrors, and the captured [accumulate_batch, lm_step_batch] x lm_batch graph
  TickSet ts[2];  // two sets: the halves of a batch alternate, one's tick runs while the host turns the other around
  DevBuf<sicp::LmState> d_bstates;
  DevBuf<sicp::EvalIn> d_ein;         // [2 per pair]: what an evaluation reads of its pair when the accumulate launch steps the machine itself
  DevBuf<unsigned> d_solo_sync;       // the last pair still iterating: hand-off words of the persistent solve (solve_one_kernel)
  unsigned solo_tag = 0;              // its tags so far (a launch uses solo_tag + 1 ...: the words are never zeroed in between)
  int solo_seq = 0, solo_pair = 0;    // launch counter (the state's pad_ word echoes it at a regular end) and the pair's state slot
  bool solo_was_init = false, solo_failed = false;  // the launch in flight starts a solve / the last one did not run to its end
  int solo_skip = 0, solo_penalty = 0;  // after a persistent launch timed out: solves that stay with the tick graph before the next try (doubling)
  bool count_stats = false;           // the align() in progress reports statistics: every search also counts its live slots
  bool counted_in_search = false;     // ... and the search kernel of the current correspondences did so itself
  DevBuf<double> d_bout28;
  sicp::LmState* h_bstates = nullptr;
  int* h_solo_flag = nullptr;  // pinned: the persistent solve's master writes its launch number here at a regular end (solve.cpp: solo_wait)
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
    unsigned flags;  // SICP_SUBMIT_*
    int overtaken = 0;  // admission rounds in which it waited for a live registration to let go of its clouds (worker only)
  };
  std::deque<Submission> queue;
  std::deque<sicp_stream_result> done;
  std::unordered_map<long long, std::vector<uint32_t>> labels;  // ticket -> getFusedLabels of a SICP_SUBMIT_FUSED_LABELS registration, until taken
  std::unordered_map<long long, std::shared_ptr<Cloud>> clouds;
  long long next_cloud = 1, next_ticket = 1;
  long long submitted = 0, completed = 0, busy_evals = 0, slot_evals = 0;
  std::atomic<double> last_add_ms{-1e18};  // when the last scan was added (its upload + tree build run beside the ticks)
  int draining = 0;  // callers blocked in sicp_stream_poll(wait >= 2): nothing new will be submitted by them meanwhile
  int in_flight = 0;
  bool stop = false;
  int error = 0;
  std::string error_msg;   // why the worker stopped (fatal for the stream)
  std::string api_error;   // what an entry point caught at the ABI barrier
  std::string error_copy;  // what sicp_stream_last_error last handed out (stable until its next call)
  // ---- worker only
  std::vector<long long> slot_ticket;
  std::vector<double> slot_t0;
  std::vector<unsigned> slot_flags;
  std::vector<hipEvent_t> slot_ev;  // SICP_SUBMIT_FUSED_LABELS: recorded behind the label kernel + read-back of the slot
  std::thread worker;
};

namespace sicp {
namespace host {

// the barrier of an entry point that has a handle / a stream: the description lands in sicp_last_error /
// sicp_stream_last_error
template <class Body>
inline int abi_guard(sicp_context* h, Body&& body) noexcept {
  return abi_guard_note(static_cast<Body&&>(body), [h](const char* what) {
    if (h && what) h->last_error = std::string("internal: ") + what;
  });
}
template <class Body>
inline int abi_guard(sicp_stream_ctx* S, Body&& body) noexcept {
  return abi_guard_note(static_cast<Body&&>(body), [S](const char* what) {
    if (!S || !what) return;
    std::lock_guard<std::mutex> lock(S->m);
    S->api_error = std::string("internal: ") + what;
  });
}

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      h->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return _e == hipErrorOutOfMemory ? SICP_ERR_OUT_OF_MEMORY : SICP_ERR_HIP;                \
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

int set_device(sicp_context* h);

// ---- clouds.cpp ------------------------------------------------------------------------------------
void settle_cloud(Cloud& c);
int cloud_wait(sicp_context* h, Cloud& c);
// host side of an upload: the caller's arrays -> the cloud's pinned staging buffers.  The caller's layout is a base
// pointer per coordinate and one byte stride (SoA: three arrays, stride 4; a pcl::PointXYZL array: one base + 0 / 4 /
// 8, stride 32), labels likewise.  ONE pass over the cloud: finite test, copy, bounding box, label range (a scan
// sequence stages a cloud per registration on the thread that submits them: five passes were 0.3 ms per 100K points).
struct StridedCloud {
  const char *x, *y, *z, *label;  // label may be null
  long long stride, label_stride;
};
int stage_cloud(sicp_context* h, Cloud& c, int32_t n, const StridedCloud& in);
int prepare_cloud(sicp_context* h, Cloud& c);
int reserve_features(sicp_context* h, Cloud& c);
int set_cloud_common(sicp_handle h, int which, int32_t n, const StridedCloud& in);

// ---- stages.cpp ------------------------------------------------------------------------------------
// (fold: the EM weights written by the search's epilogue -- KnnArgs::w_* -- or nullptr)
struct WeightFold {
  const sicp::PointRec *srec, *trec;
  const double *sproj, *tproj;
  double* w;
  double one_m_eps;
  int C, bool_probability;
};
int run_nn(sicp_context* h, int K, const Cloud& Qc, int q_begin, int q_count, const double* M34, const Cloud& Tc,
           int tseg, bool self, float gate_sq, int* out_i, float* out_d, int timer_bit, hipStream_t stream, int out_stride = 0,
           const WeightFold* fold = nullptr);
int ensure_hval(sicp_context* h, int k);
int ensure_proj(sicp_context* h, Cloud& c);
int compute_features(sicp_context* h, Cloud& c, bool with_hist, hipStream_t stream = nullptr);
bool features_current(const sicp_context* h, const Cloud& c, bool with_hist);
int check_ready(sicp_context* h, bool need_cm);
void fill_pose(const double* qt, sicp::Pose& p);
int segment_of(const Cloud& c, uint32_t label);
int count_active(sicp_context* h);
bool weights_from_histograms(const sicp_params& P, int K);
int run_weights(sicp_context* h, const double* qt);
int run_correspondences(sicp_context* h, const double* qt, int K, bool weights);
void fill_acc(sicp_context* h, sicp::AccArgs& a);
extern const int kMaxActivePairs;  // pairs one launch evaluates (12 bytes of LDS each in the accumulate kernel)
int eval28(sicp_context* h, const double* qt, double* out28);
// ---- pieces of align() shared by the single-pair and the lock-step batch drivers ---------------
struct OuterState {
  double cur[7], est[7];
  int outer = 0, count = 0;
  bool converged = false;
};

int align_begin(sicp_context* h, bool want_stats);
void outer_finish(const sicp_params& P, OuterState& o);
int align_end(sicp_context* h, const OuterState& o, double t_begin, int32_t* outer_iters, sicp_stats* stats);
int batch_slice(int p, int n, int knn);
int flush_jobs(sicp_context* h, JobCollector& jc, hipStream_t base = nullptr);

// ---- solve.cpp -------------------------------------------------------------------------------------
sicp::LmOptions lm_options(const sicp_params& P);
struct SolveResult {
  int status = 0, iterations = 0, evaluations = 0;
  double cost = 0;
};

hipError_t create_side_stream(hipStream_t* st);  // the stream of a batch's / a stream's searches and feature kernels (SICP_SIDE_PRIORITY: A/B aid)
bool solo_allowed(sicp_context* h);
bool general_covariances(const sicp_context* h);  // either cloud carries caller covariances of general form
int align_host_loop(sicp_context* h, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats);
bool lm_step_in_launch();
int run_solve(sicp_context* h, const double* init_qt, double* out_qt, SolveResult* res);
bool same_solver(const sicp_params& a, const sicp_params& b);
int tickset_reserve(sicp_context* h, TickSet& S, int n);
int batch_reserve(sicp_context* h, int n);
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
int tick_launch(sicp_context* h, TickSet& S, hipStream_t M, sicp_handle* hs, int lo, int hi, const std::vector<int>& act,
                const std::vector<int>& joining, const double (*start)[7], int len, int solo_evals = 0);
int tick_wait(sicp_context* h, hipStream_t M);
int solo_check(sicp_context* h);
int run_tick(sicp_context* h, hipStream_t M, sicp_handle* hs, int n, const std::vector<int>& act, const std::vector<int>& joining,
             const double (*start)[7], int len, int solo_evals);
// ---- continuous batching: what sicp_align_batch (a closed set of pairs) and sicp_stream_* (pairs that come
// and go) share.  Every pair runs its own sequence
//     search (transform + kNN + weights) -> inner solve -> convergence test -> search -> ...
// and the run advances in TICKS of `len` LM evaluations: one graph launch evaluates every pair that is
// inside an inner solve, while the searches of the pairs that have just finished one run on a second
// stream beside it; those pairs rejoin at the next tick.  No pair waits for another pair's solve or outer
// loop -- only for the end of the current tick.
// PAIR_FIRST: the pair's start-up pipeline (features, first search, weights) is queued on the start-up stream; it joins
// the ticks when the event of its chunk has completed
// PAIR_LABELS (streams only): converged; its getFusedLabels pass is queued and the slot waits for the labels
enum { PAIR_FREE = -1, PAIR_NEED_SEARCH = 0, PAIR_JOINING, PAIR_SOLVING, PAIR_DONE, PAIR_FIRST, PAIR_LABELS };

// pairs [lo, hi) that advance together: one tick stream, one argument set
struct TickGroup {
  int lo = 0, hi = 0;
  hipStream_t M = nullptr;
  TickSet* S = nullptr;
  hipEvent_t side_done = nullptr;
  bool pending = false, side_recorded = false;
  bool force_wait = false;  // the next tick must wait for what the side stream has been given so far (a stream's ordered feature rewrite)
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
  bool acc_static = false;      // equal chunk ranges in the accumulate launches of the next tick (TickSet::static_ranges)
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

// getFusedLabels (em_icp.hpp:202-268) in two halves, so that a stream slot need not wait for it: queue the K = 4 search
// (collected into the handle's job list when it has one: flush before `labels_launch`) ...
int labels_search(sicp_context* h, const double* qt);
// ... and the label kernel behind it on `st`, device order -> h->tmpl
int labels_launch(sicp_context* h, const double* qt, hipStream_t st);
int align_batch(sicp_handle* hs, int32_t n, const double* init_qt, double* out_qt, int32_t* outer_iters, sicp_stats* stats);

}  // namespace host
}  // namespace sicp
#endif
