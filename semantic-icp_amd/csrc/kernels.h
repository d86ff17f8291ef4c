// kernels.h -- argument blocks and launch wrappers of the gfx950 kernels (knn_kernels.hip, feature_kernels.hip, solve_kernels.hip).
#ifndef SICP_KERNELS_H_
#define SICP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh.hpp"
#include "lm.hpp"

#ifndef SICP_HD
#define SICP_HD
#endif

namespace sicp {

struct Pose {
  double R[9];  // row-major rotation (Eigen Quaternion::toRotationMatrix of the pose)
  double t[3];
};

struct Mat4f {
  float m[16];
};

// One point as the weight / accumulate kernels gather it: position (the float32 the search ran on) and
// the unit normal of its k-neighbourhood in double (its covariance is I - (1-eps) n n^T), 48 bytes =
// three loads instead of one load from each of six arrays.  Written by cov_kernel.
// Field order: the batched accumulate kernel moves a record into LDS with three LDS-DMA loads of
// 16 + 16 + 4 bytes (the 36 bytes that carry data).
struct alignas(16) PointRec {
  double nx, ny;
  double nz;
  float x, y;
  float z;
  uint32_t pad_[3];
};
static_assert(sizeof(PointRec) == 48, "PointRec is read with three dwordx4 / dwordx2 loads");

// Build-time shape of the accumulate kernels (tuning: see DESIGN.md section 3):
//   SICP_SG4     slots per group when K is a multiple of 4 (4 = the K = 4 slots of one source point per
//                step, 2 = half of them: fewer live registers per lane)
//   SICP_ACC_OCC workgroups (of 4 waves) per CU the batched kernel is compiled for = waves per SIMD
#ifndef SICP_SG4
#define SICP_SG4 4
#endif
#ifndef SICP_ACC_OCC
#define SICP_ACC_OCC 2
#endif
constexpr int acc_slots_per_group(int K) { return K % 4 == 0 ? SICP_SG4 : 2; }

// Decomposition of a pair's correspondence slots into chunks (solve_kernels.hip): groups of
// `slots_per_group` slots; a chunk = what one 256-lane workgroup sums = 2048 m slots (m = 1 up to 1024
// chunks), every lane taking `steps` = 8 m / slots_per_group groups, 256 apart; one row of 28 partial
// sums per chunk.  Host and device compute it from the slot count alone.
struct AccGeometry {
  int n_groups, steps, chunk_groups, n_chunks;
};
SICP_HD inline AccGeometry acc_geometry(int total_slots, int slots_per_group) {
  AccGeometry g;
  const int per_lane = 8 / slots_per_group;  // groups per lane and chunk at m = 1
  g.n_groups = (total_slots + slots_per_group - 1) / slots_per_group;
  int m = (g.n_groups + 256 * per_lane * 1024 - 1) / (256 * per_lane * 1024);
  if (m < 1) m = 1;
  g.steps = per_lane * m;
  g.chunk_groups = 256 * g.steps;
  g.n_chunks = (g.n_groups + g.chunk_groups - 1) / g.chunk_groups;
  if (g.n_chunks < 1) g.n_chunks = 1;
  return g;
}

struct LossArgs {
  double cauchy_a;
  int use_sqloss;
};

// brute force: queries [q_begin, q_begin+q_count) (SoA, device order) against the packed points
// pts4[t_begin .. t_begin+t_count) = (x, y, z, caller index bits)
struct NNArgs {
  const float *qx, *qy, *qz;
  int q_begin, q_count;
  int do_xform;   // 1: query = float(M * p) (pcl::transformPointCloud), 0: query = p
  double M[12];   // rows 0..2 of the 4x4 pose matrix
  const float4* pts4;
  int t_begin, t_count;
  int chunk_len;              // targets per grid.y slice
  unsigned long long* part;   // [n_chunks][q_count][K] keys
};

struct MergeArgs {
  int q_begin, q_count, n_chunks;
  const unsigned long long* part;
  const int* inv;   // caller index -> device index of the target cloud
  float gate_sq;    // +inf for the covariance self-query
  int* out_i;       // [n][k_out] device indices of the target cloud, -1 = none / gated out
  float* out_d;     // [n][k_out] or nullptr
  int k_out;        // neighbours written per query (<= the list length K the kernels run with)
};

// one tree = one cloud segment (bvh.hpp)
struct TreeArgs {
  const float4* pts4;     // packed points of the whole cloud, every segment padded to kLeaf
  const float4* box_lo;   // boxes of all segments
  const float4* box_hi;
  const unsigned long long* leaf_code;  // first Morton code of every leaf, all segments
  TreeLevels lv;
  int n;                  // points in this segment
  int pt_begin;           // first packed point of the segment
  int node_begin;         // first box of the segment
  int code_begin;
  float lo[3];
  float scale;
};

struct KnnArgs {
  const float *qx, *qy, *qz;
  int q_begin, q_count;
  int do_xform;
  double M[12];
  TreeArgs tree;
  int self;        // queries are the tree's own points (covariance neighbourhoods)
  float gate_sq;
  const int* inv;
  int* out_i;
  float* out_d;
  int* dbg;        // nullable: [q_count][2] = (boxes tested, leaves scanned), debugging only
  const int* seed_hint;  // nullable, packet kernel: [query][hint_K] device indices found by the previous search of
                         // the same queries (the seed of the walk; any valid target index is a legal hint)
  int hint_K, t_begin;   // t_begin = device index of the target segment's first point
  int out_stride;  // 0: out_i / out_d are [query][k_out]; > 0: [k_out][out_stride] (packet kernel only: coalesced
                   // for the covariance kernel, which reads one neighbour rank of 64 points at a time)
  int k_out;       // neighbours written per query (<= the list length K the kernel runs with)
  unsigned long long* live_cnt;  // nullable, packet kernel: kLiveCounters partial counters; every wave adds the number of
                                 // neighbours it wrote that passed the gate (statistics: sicp_stats.total_active)
  // EM weights in the search's epilogue (packet kernel, K = 4, at most 16 classes; w_out == nullptr: not wanted).  The lane
  // that writes a neighbour's index already holds everything em_weight_rows4_kernel would re-read -- the index, the query's
  // place -- so it gathers the two records and projection rows and writes the slot's weight beside the index: the same
  // operations in the same order as that kernel (em_icp.hpp:84-89,108), hence the same bits.  The pose is M (rows [R | t]).
  const PointRec *w_srec, *w_trec;
  const double *w_sproj, *w_tproj;  // [n][proj_stride(w_C)]
  double* w_out;                    // [query][k_out]
  double w_one_m_eps;
  int w_C, w_bool_probability;
};
constexpr int kLiveCounters = 1024;  // (spread: ~6 of a 100K-query search's 6250 waves per counter)

struct CovArgs {
  int n, k, C;
  const float *x, *y, *z;
  const uint32_t* label;  // nullable
  const int* nn;          // neighbour lists, device indices: [n][k], or [k][nn_stride] when nn_stride > 0
  int nn_stride;
  int float_products;
  PointRec* rec;          // out: position + normal of every point
  uint8_t* hist;          // [n][hist_stride(C)] or nullptr
  char* rec_dense;        // out, nullable: the same 36 data bytes per point as three dense arrays (dense_rec_*)
  int rec_dense_n;        // points the dense arrays are laid out for (their pitch)
};

// The records again, dense: [n] x 16 B (nx ny) | [n] x 16 B (nz x y) | [n] x 4 B (z).  What the accumulate kernel STREAMS
// -- the source points of a pass, in order -- is read from here: 36 bytes per point from HBM instead of the 48 of a
// record (the 48-byte record is what a gather wants: one point, one place).
SICP_HD inline size_t dense_rec_bytes(int n) { return (size_t)(n > 0 ? n : 1) * 36; }

// rows of the label histograms (uint8 neighbour counts) are padded to 16 bytes: one aligned dwordx4 load fetches the row of up
// to 16 classes (the EM weight kernel gathers a target's 16-byte row instead of its 96-byte projection row)
SICP_HD inline int hist_stride(int C) { return (C + 15) & ~15; }

// rows of the projection arrays are padded to an even number of doubles: 16-byte aligned, read with
// dwordx4 loads by the weight kernels
SICP_HD inline int proj_stride(int C) { return (C + 1) & ~1; }

struct ProjArgs {
  int n, C;
  const uint8_t* hist;  // [n][hist_stride(C)] neighbour counts
  const double* cm;     // C*C row-major
  const double* hval;   // hval[c] = c additions of 1/k (em_icp.hpp:279,301)
  double* proj;         // [n][proj_stride(C)]
};

struct WeightArgs {
  int n_s, K, C;
  const int* idx;
  const PointRec *srec, *trec;
  const double *s_proj, *t_proj;  // [n][proj_stride(C)] label distributions projected through CM (proj_kernel); unused with histograms
  // K = 4, C <= 16: the weights straight from the label histograms (rows of hist_stride(C) = 16 bytes), the projections
  // formed in the kernel -- the same sums in the same order as proj_kernel's, so the same bits -- from cm / hval
  const uint8_t *s_hist, *t_hist;  // nullable: then s_proj / t_proj are read
  const double *cm, *hval;         // C*C row-major; hval[c] = c additions of 1/k
  Pose pose;
  double one_m_eps;
  int bool_probability;
  double* w;
};

// What one evaluation of a device-resident solve reads about its pair -- pose and status -- when the LM step runs INSIDE the
// accumulate launch (the last workgroup to finish a pair's chunks steps its machine: accumulate_staged_kernel).  Two entries
// per pair, picked by the parity of the evaluation's EPOCH (BatchHeader::epoch_base + the launch's index in its tick): launch
// e reads entry e & 1 and its LM step writes entry (e + 1) & 1, so a workgroup that is dispatched late -- beside a flood of
// search workgroups they are -- still sees what every other workgroup of the launch saw.  An entry counts only when its
// epoch is the launch's: a pair that finished at e is not running at e + 2.
struct alignas(16) EvalIn {
  double pose[7];
  int status;       // LM_*
  unsigned epoch;
};
static_assert(sizeof(EvalIn) == 64, "one cache line half");

struct AccArgs {
  int n_s, K;
  const int* idx;
  const double* w;  // nullable (weight 1)
  const PointRec *srec, *trec;
  const char* srec_dense;  // nullable: the source records as dense arrays (dense_rec_*), pitch n_s
  Pose pose;            // used when lm == nullptr
  const LmState* lm;    // device-resident solve: evaluate at lm->pose, skip when it has finished
  LmState* lm_step;     // batched solve: the state lm_step_batch_kernel advances (== lm)
  EvalIn* ein;          // nullable: [2]; with it the accumulate launch steps the machine itself (no lm_step_batch_kernel)
  double one_m_eps;
  LossArgs loss;
  double* partials;  // [28][accumulate_blocks]
};

// One evaluation of ONE pair whose covariances are (partly) the caller's own, of general form: full symmetric 3x3 matrices
// (six doubles per point, device order; nullptr: that cloud's are I - (1-eps) n n^T from its records).  Same columns of
// partials[28][n_chunks] as the product kernel writes, summed by the same finalize_batch_kernel.
struct GenAccArgs {
  AccArgs a;
  const double *scov6, *tcov6;
  int n_chunks, pad_;
};
hipError_t launch_accumulate_general(const GenAccArgs& g, hipStream_t st);

// job arrays passed by value to one launch (lock-step batch); sized to stay inside the 4 KB of
// kernel arguments
constexpr int kMaxKnnJobs = 8;
constexpr int kMaxSmallJobs = 16;
struct KnnJobs { KnnArgs job[kMaxKnnJobs]; };
struct CovJobs { CovArgs job[kMaxSmallJobs]; };
struct ProjJobs { ProjArgs job[kMaxSmallJobs]; };
struct WeightJobs { WeightArgs job[kMaxKnnJobs]; };
struct CountJob { const int* idx; int n; unsigned long long* out; };
struct CountJobs { CountJob job[kMaxSmallJobs]; };
static_assert(sizeof(KnnJobs) <= 4000 && sizeof(WeightJobs) <= 4000 && sizeof(CovJobs) <= 4000, "kernel argument segment");

// one pair of a lock-step batch (sicp_align_batch); an array of these lives in HBM
struct BatchArgs {
  AccArgs a;
  int nb;    // columns of this pair's partials ([28][nb]): its chunks
  int pad_;
};
// what the batched kernels need to know about the current launch; lives in HBM next to the BatchArgs
// array, so the instantiated graph never changes
struct BatchHeader {
  int n_pairs;   // entries of the BatchArgs array (pairs whose solve has ended are skipped on the device)
  unsigned epoch_base;  // epoch of the current tick's first evaluation (advanced on the device by tick_prepare_kernel)
  int pad_[2];
};

// [accumulate, lm_step_batch] x len of a lock-step batch as an instantiated graph with explicit
// kernel nodes and fixed grids (solve_kernels.hip)
constexpr int kMaxBatchLen = 32;
struct BatchGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int len = 0, K = 0, sqloss = 0, capacity = 0, fold = 0, static_ranges = 0;
  const BatchArgs* batch = nullptr;
  const BatchHeader* hdr = nullptr;
};
// *built is set to 1 when the graph had to be (re)instantiated
// fold != 0: [tick_prepare, accumulate x len] (the accumulate launches step the LM machines: AccArgs::ein is set);
// static_ranges: every accumulate node carries kAccStaticRanges
hipError_t batch_graph_prepare(BatchGraph& g, int K, int use_sqloss, const BatchHeader* hdr, const BatchArgs* batch, int capacity, int len,
                               int* built, int fold = 0, int static_ranges = 0);
void batch_graph_destroy(BatchGraph& g);

// neighbour-list length the search kernels run with for a request of k neighbours (the k nearest
// are the first k of any longer exact list): 1, 4, 20 or 32; 0 = unsupported (k < 1 or k > 32)
int nn_list_len(int k);
bool nn_k_supported(int K);
int nn_queries_per_thread(int K);
hipError_t launch_nn_partial(int K, const NNArgs& a, int n_chunks, hipStream_t st);
hipError_t launch_nn_merge(int K, const MergeArgs& m, hipStream_t st);
hipError_t launch_bvh_knn(int K, const KnnArgs& a, hipStream_t st);
hipError_t launch_bvh_knn_quad(int K, const KnnArgs& a, hipStream_t st);
hipError_t launch_bvh_knn_packet(int K, const KnnArgs& a, hipStream_t st);
hipError_t launch_cov(const CovArgs& a, hipStream_t st);
hipError_t launch_proj(const ProjArgs& a, hipStream_t st);
// sicp_covariances' fast path: the n 3x3 matrices (row-major) in the CALLER's point order, formed on the device from the records'
// normals (C = I - (1 - eps) n n^T) or from the caller's own general matrices (cov6 != nullptr); perm = device -> caller index
hipError_t launch_cov9_caller_order(int n, const PointRec* rec, const double* cov6, const int* perm, double one_m_eps, double* out9, hipStream_t st);
// caller-supplied normals (sicp_set_covariances): the point records and their dense copy, as cov_kernel writes them
hipError_t launch_set_normals(int n, const float* x, const float* y, const float* z, const double* normal3, PointRec* rec, char* rec_dense,
                              int rec_dense_n, hipStream_t st);
hipError_t launch_em_weight(const WeightArgs& a, hipStream_t st);
hipError_t launch_fused_labels(const WeightArgs& a, uint32_t* out, hipStream_t st);
hipError_t launch_bvh_knn_packet_jobs(int K, const KnnArgs* jobs, int n, hipStream_t st);
hipError_t launch_cov_jobs(const CovArgs* jobs, int n, hipStream_t st);
hipError_t launch_proj_jobs(const ProjArgs* jobs, int n, hipStream_t st);
hipError_t launch_em_weight_jobs(const WeightArgs* jobs, int n, hipStream_t st);
hipError_t launch_count_active_jobs(const CountJob* jobs, int n, hipStream_t st);
int accumulate_blocks(int total, int K);  // chunks of a pair with `total` slots, K correspondences per source point
// every pair of the batch in one launch: hdr / batch in HBM, capacity = slots of the batch buffers
// node = index of the launch inside its tick (its epoch is hdr->epoch_base + node; only read by pairs with AccArgs::ein),
// optionally | kAccStaticRanges: equal contiguous chunk ranges even for a large launch (see accumulate_staged_kernel)
constexpr int kAccStaticRanges = 0x40000000;
hipError_t launch_accumulate_batch(int K, int use_sqloss, const BatchHeader* hdr, const BatchArgs* batch, int capacity, hipStream_t st, int node = 0);
hipError_t launch_lm_step_batch(const BatchHeader* hdr, const BatchArgs* batch, int capacity, hipStream_t st);
// first kernel of a tick whose accumulate launches step the LM machines themselves: advances hdr->epoch_base and
// publishes every pair's pose / status (from its LmState) as the first evaluation's EvalIn
hipError_t launch_tick_prepare(BatchHeader* hdr, const BatchArgs* batch, hipStream_t st);
// one pair alone: the whole inner solve of batch[0] in one persistent launch (one workgroup per chunk); its partials
// buffer holds TWO sets of columns, sync = max_evals + 1 words (word 0 is raised when a device-wide wait timed out)
bool solve_one_fits(int total_slots, int K);
constexpr int kSoloMaxEvals = 1024;   // evaluations one persistent launch may run (a solve that needs more is relaunched)
constexpr int kSoloSyncWords = 288;   // its hand-off words (zeroed once, when allocated); 16 words of developer timers follow

hipError_t launch_finalize_batch(const BatchArgs* batch, int n, double* out28, hipStream_t st);
// pairs that start an inner solve with the next tick: their LM states are initialised ON the device from
// one small upload (states[j.pair] = lm_init(j.opt, j.start)) instead of one 800-byte copy per pair
struct LmJoin {
  int pair, pad_;
  double start[7];
  LmOptions opt;
};
hipError_t launch_lm_init(const LmJoin* joins, int n, LmState* states, hipStream_t st);
// One persistent launch of the last pair still iterating (solve_kernels.hip: solve_one_kernel); everything it
// needs travels in the kernel arguments: no upload, no state initialisation kernel, no memset ahead of it.
struct SoloArgs {
  AccArgs a;           // the pair; a.lm_step = its trust-region state in HBM
  unsigned* sync;      // kSoloSyncWords hand-off words (+ 16 developer timers)
  int max_evals;       // evaluations this launch may run
  int wait_ticks;      // 100 MHz ticks before a wait gives up
  unsigned tag_base;   // the launch's hand-off tags are tag_base + 1 ... tag_base + max_evals: older words never match
  int init;            // 1: the inner solve starts with this launch (state := lm_init(opt, start)), 0: it continues
  int seq;             // written to the state's pad_ word at a regular end: the host's proof the launch ran to it
  int pad_;
  // nullable: pinned, device-visible host memory.  At a regular end the master also writes the state THERE (plain stores, a
  // system-scope fence) and then `seq` into *host_flag: the host polls that word instead of waiting for a read-back copy
  // behind the kernel (copy kernel + completion signal + wake-up: ~30 us between an inner solve and the next search of a
  // pair alone, four or five times per align()).
  LmCore* host_state;
  int* host_flag;
  double start[7];
  LmOptions opt;
};
hipError_t launch_solve_one(int K, int use_sqloss, const SoloArgs& args, int n_chunks, hipStream_t st);
int solo_wait_ticks();
hipError_t launch_count_active(const int* idx, int n, unsigned long long* out, hipStream_t st);
// test hook: csrc/se3.hpp on the device, one lane per item (op = SICP_SE3_*; in/out strides per op)
// op 5 / 6: the trust-region machine fed with a given sequence of evaluations, as the kernels run it (a whole wave) / as the host
// runs it (one lane): in = n x kLmSeqIn (start pose | kLmSeqEvals x 28 sums), out = n x kLmSeqOut (the final state)
constexpr int kLmSeqEvals = 24, kLmSeqIn = 7 + 28 * kLmSeqEvals, kLmSeqOut = 37;
hipError_t launch_se3_ops(int op, int n, const double* in, double* out, hipStream_t st);
hipError_t launch_transform_float(int n, const float* x, const float* y, const float* z, const Mat4f& M,
                                  float* ox, float* oy, float* oz, hipStream_t st);

}  // namespace sicp
#endif
