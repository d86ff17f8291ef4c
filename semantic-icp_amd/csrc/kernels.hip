// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the semantic-ICP hot path.
//
// What each kernel replaces in the reference (paths relative to /root/reference):
//   nn_partial / nn_merge : pcl::transformPointCloud + KdTreeFLANN::nearestKSearch + the
//                           dist^2 < 250 gate            em_icp.hpp:46-65, gicp.hpp:54-70,
//                                                        semantic_icp.hpp:53-69
//   cov_kernel            : ComputeCovariances body      em_icp.hpp:298-340
//   em_weight_kernel      : label posterior * Probability em_icp.hpp:77-89,108
//   accumulate / finalize : GICPCostFunction::Evaluate + LocalParameterizationSE3 + losses +
//                           Ceres' Corrector, summed to 28 doubles
//                                                        gicp_cost_function.h:27-73
//
// Design notes (MI355X): clouds live in HBM in Hilbert-curve order (SoA float32 + a packed float4
// x,y,z,caller-index copy for the search kernels).  Two exact kNN engines produce identical
// results: an LDS-tiled brute force (target tiles broadcast from LDS, 2-D grid of query blocks x
// target chunks, deterministic merge) and a stackless walk of a 4-ary box tree over the curve
// order.  Top-K lists are 64-bit (distance, caller index) keys in statically indexed VGPRs.
// No floating-point atomics anywhere, so every result is run-to-run reproducible.  Nothing here
// is GEMM shaped: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define SICP_HD __host__ __device__
#include "kernels.h"

namespace sicp {

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
// NOTE: this file is compiled with -ffp-contract=off (build.py).  HIP's __fmul_rn/__fadd_rn are
// plain operators that hipcc's default -ffp-contract=fast would fuse into FMAs; the float32
// distance, the double transform and the float32 moment products must round exactly like the
// reference's x86 code (separate multiply and add), or neighbour order flips on near-ties.

// pcl::transformPointCloud<PointT,double>: (((m0*x + m1*y) + m2*z) + m3) in double, no
// contraction, then one rounding to float.
__device__ __forceinline__ float xform_row(const double* m, double x, double y, double z) {
#pragma clang fp contract(off)
  double a = __dmul_rn(m[0], x);
  a = __dadd_rn(a, __dmul_rn(m[1], y));
  a = __dadd_rn(a, __dmul_rn(m[2], z));
  a = __dadd_rn(a, m[3]);
  return __double2float_rn(a);
}

// FLANN L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz, every product and sum rounded to float
// (no FMA contraction, so that neighbour order matches the CPU kd-tree bit for bit).
__device__ __forceinline__ float l2_simple(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  float r = __fmul_rn(dx, dx);
  r = __fadd_rn(r, __fmul_rn(dy, dy));
  r = __fadd_rn(r, __fmul_rn(dz, dz));
  return r;
}

// ---- (distance, caller index) keys -----------------------------------------------------------
// A neighbour is the 64-bit key  float_bits(d2) << 32 | caller_index.  d2 >= +0, so the float
// bit pattern orders like the value and one unsigned compare is the lexicographic order
// "smaller distance first, lower caller index on exact ties" (the tie rule this build defines;
// FLANN leaves tie order unspecified) --
// independent of the order in which candidates are visited (device order is curve order).
typedef unsigned long long u64;
// Empty list entry: above every real key (a float d2 pattern is at most 0x7f800000 = +inf), caller
// index -1, NaN as a float distance -- and, read as a double, the largest FINITE value (see below).
static constexpr u64 KEY_EMPTY = 0x7fefffffffffffffull;

__device__ __forceinline__ u64 make_key(float d, unsigned orig) { return ((u64)__float_as_uint(d) << 32) | orig; }
__device__ __forceinline__ float key_dist(u64 k) { return __uint_as_float((unsigned)(k >> 32)); }

// Keys are sorted with TWO instructions per list element.  Every key is a non-negative, non-NaN bit
// pattern when read as an IEEE double (sign bit clear, exponent field < 0x7ff because the high word
// is at most 0x7fefffff), and for such patterns the double order IS the unsigned order, so
// v_min_f64 / v_max_f64 order keys exactly; double denormals (tiny d2, e.g. the query itself at
// d2 = 0) are preserved because FP64 denormals are never flushed on gfx9.  The u64 formulation
// costs two 64-bit compares and four v_cndmask per element (8 issue slots with the VCC hazards).
//
// Insert into an ascending key list held in registers.  Precondition: key < bk[K-1].
// The new key travels down from the top as a carry c: slot j+1 receives max(bk[j], c) and c becomes
// min(bk[j], c).  Every list element is written in its own register (no temporaries, so no copies
// where the divergent "insert / do not insert" paths join), and bk[K-1] is simply dropped.
// One asm block per insertion (the compiler pads every inline-asm statement with hazard nops, and
// the fmin / fmax builtins would add a canonicalising v_max_f64 per element).
template <int K>
__device__ __forceinline__ void key_insert(u64 (&bk)[K], u64 key);

#define SICP_KI_STEP(hi, lo) "v_max_f64 %" #hi ", %" #lo ", %0\n\tv_min_f64 %0, %" #lo ", %0\n\t"
template <>
__device__ __forceinline__ void key_insert<1>(u64 (&bk)[1], u64 key) { bk[0] = key; }
template <>
__device__ __forceinline__ void key_insert<4>(u64 (&bk)[4], u64 key) {
  double c = __longlong_as_double((long long)key);
  double* b = reinterpret_cast<double*>(bk);
  asm(SICP_KI_STEP(4, 3) SICP_KI_STEP(3, 2) SICP_KI_STEP(2, 1) "v_mov_b64 %1, %0"
      : "+v"(c), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
template <>
__device__ __forceinline__ void key_insert<20>(u64 (&bk)[20], u64 key) {
  double c = __longlong_as_double((long long)key);
  double* b = reinterpret_cast<double*>(bk);
  asm(SICP_KI_STEP(20, 19) SICP_KI_STEP(19, 18) SICP_KI_STEP(18, 17) SICP_KI_STEP(17, 16) SICP_KI_STEP(16, 15) SICP_KI_STEP(15, 14)
      SICP_KI_STEP(14, 13) SICP_KI_STEP(13, 12) SICP_KI_STEP(12, 11) SICP_KI_STEP(11, 10) SICP_KI_STEP(10, 9) SICP_KI_STEP(9, 8)
      SICP_KI_STEP(8, 7) SICP_KI_STEP(7, 6) SICP_KI_STEP(6, 5) SICP_KI_STEP(5, 4) SICP_KI_STEP(4, 3) SICP_KI_STEP(3, 2)
      SICP_KI_STEP(2, 1) "v_mov_b64 %1, %0"
      : "+v"(c), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(b[8]), "+v"(b[9]),
        "+v"(b[10]), "+v"(b[11]), "+v"(b[12]), "+v"(b[13]), "+v"(b[14]), "+v"(b[15]), "+v"(b[16]), "+v"(b[17]), "+v"(b[18]),
        "+v"(b[19]));
}
#undef SICP_KI_STEP

// compare-exchange of two keys (v_min_f64 / v_max_f64, see above)
__device__ __forceinline__ void key_cswap(u64& a, u64& b) {
  const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
  double lo, hi;
  asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y));
  a = (u64)__double_as_longlong(lo);
  b = (u64)__double_as_longlong(hi);
}

// Batcher's odd-even merge sort of 16 keys (63 compare-exchanges, 2 instructions each)
template <int K>
__device__ __forceinline__ void key_sort16(u64 (&bk)[K]) {
  static_assert(K >= 16, "sorts the first 16 entries");
#define SICP_CSWAP(i, j) key_cswap(bk[i], bk[j]);
  SICP_CSWAP(0, 1) SICP_CSWAP(2, 3) SICP_CSWAP(0, 2) SICP_CSWAP(1, 3) SICP_CSWAP(1, 2) SICP_CSWAP(4, 5)
  SICP_CSWAP(6, 7) SICP_CSWAP(4, 6) SICP_CSWAP(5, 7) SICP_CSWAP(5, 6) SICP_CSWAP(0, 4) SICP_CSWAP(2, 6)
  SICP_CSWAP(2, 4) SICP_CSWAP(1, 5) SICP_CSWAP(3, 7) SICP_CSWAP(3, 5) SICP_CSWAP(1, 2) SICP_CSWAP(3, 4)
  SICP_CSWAP(5, 6) SICP_CSWAP(8, 9) SICP_CSWAP(10, 11) SICP_CSWAP(8, 10) SICP_CSWAP(9, 11) SICP_CSWAP(9, 10)
  SICP_CSWAP(12, 13) SICP_CSWAP(14, 15) SICP_CSWAP(12, 14) SICP_CSWAP(13, 15) SICP_CSWAP(13, 14) SICP_CSWAP(8, 12)
  SICP_CSWAP(10, 14) SICP_CSWAP(10, 12) SICP_CSWAP(9, 13) SICP_CSWAP(11, 15) SICP_CSWAP(11, 13) SICP_CSWAP(9, 10)
  SICP_CSWAP(11, 12) SICP_CSWAP(13, 14) SICP_CSWAP(0, 8) SICP_CSWAP(4, 12) SICP_CSWAP(4, 8) SICP_CSWAP(2, 10)
  SICP_CSWAP(6, 14) SICP_CSWAP(6, 10) SICP_CSWAP(2, 4) SICP_CSWAP(6, 8) SICP_CSWAP(10, 12) SICP_CSWAP(1, 9)
  SICP_CSWAP(5, 13) SICP_CSWAP(5, 9) SICP_CSWAP(3, 11) SICP_CSWAP(7, 15) SICP_CSWAP(7, 11) SICP_CSWAP(3, 5)
  SICP_CSWAP(7, 9) SICP_CSWAP(11, 13) SICP_CSWAP(1, 2) SICP_CSWAP(3, 4) SICP_CSWAP(5, 6) SICP_CSWAP(7, 8)
  SICP_CSWAP(9, 10) SICP_CSWAP(11, 12) SICP_CSWAP(13, 14)
#undef SICP_CSWAP
}

// a fresh list; the asm keeps the compiler from treating the K equal constants as one value (it
// would share one register among them and copy at every control-flow join of the first scans)
template <int K>
__device__ __forceinline__ void key_list_init(u64 (&bk)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bk[k] = KEY_EMPTY;
    asm volatile("" : "+v"(bk[k]));
  }
}

// candidate test shared by both search kernels; wd caches key_dist(bk[K-1]) (NaN while the list
// is not full, and `!(d > NaN)` is true)
template <int K>
__device__ __forceinline__ void consider(u64 (&bk)[K], float& wd, float d, unsigned orig) {
  if (!(d > wd)) {
    const u64 key = make_key(d, orig);
    if (key < bk[K - 1]) {
      key_insert<K>(bk, key);
      wd = key_dist(bk[K - 1]);
    }
  }
}

__device__ __forceinline__ void load_query(const float* qx, const float* qy, const float* qz, int g, int do_xform,
                                           const double* M, float& px, float& py, float& pz) {
  const float x = qx[g], y = qy[g], z = qz[g];
  if (do_xform) {
    const double dx = x, dy = y, dz = z;
    px = xform_row(M + 0, dx, dy, dz);
    py = xform_row(M + 4, dx, dy, dz);
    pz = xform_row(M + 8, dx, dy, dz);
  } else {
    px = x; py = y; pz = z;
  }
}

// results -> device indices (caller index -> device index through inv[]), gate, distances
template <int K>
__device__ __forceinline__ void emit(const u64 (&bk)[K], const int* inv, float gate_sq, int* out_i, float* out_d, size_t o) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const unsigned orig = (unsigned)bk[k];
    const float d = bk[k] == KEY_EMPTY ? INFINITY : key_dist(bk[k]);
    const bool keep = orig != 0xffffffffu && d < gate_sq;  // strict <, float compare (em_icp.hpp:65)
    out_i[o + k] = keep ? inv[orig] : -1;
    if (out_d) out_d[o + k] = d;
  }
}

// ------------------------------------------------------------------------------------------
// brute-force kNN, partial pass: grid = (query blocks, target chunks)
// ------------------------------------------------------------------------------------------
template <int K, int Q, int BS, int TILE>
__global__ __launch_bounds__(BS) void nn_partial_kernel(NNArgs a) {
  __shared__ float4 tile[TILE];
  const int chunk = blockIdx.y;
  const int c_lo = chunk * a.chunk_len;
  const int c_hi = min(c_lo + a.chunk_len, a.t_count);

  float px[Q], py[Q], pz[Q], wd[Q];
  u64 bk[Q][K];
#pragma unroll
  for (int j = 0; j < Q; ++j) {
    int q = blockIdx.x * (BS * Q) + j * BS + threadIdx.x;
    q = min(q, a.q_count - 1);
    load_query(a.qx, a.qy, a.qz, a.q_begin + q, a.do_xform, a.M, px[j], py[j], pz[j]);
#pragma unroll
    for (int k = 0; k < K; ++k) bk[j][k] = KEY_EMPTY;
    wd[j] = key_dist(KEY_EMPTY);
  }

  for (int t0 = c_lo; t0 < c_hi; t0 += TILE) {
    const int n = min(TILE, c_hi - t0);
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += BS) tile[p] = a.pts4[a.t_begin + t0 + p];
    __syncthreads();
#pragma unroll 4
    for (int p = 0; p < n; ++p) {
      const float4 t = tile[p];  // wave-uniform address: one broadcast LDS read
#pragma unroll
      for (int j = 0; j < Q; ++j) consider<K>(bk[j], wd[j], l2_simple(px[j], py[j], pz[j], t.x, t.y, t.z), __float_as_uint(t.w));
    }
  }

#pragma unroll
  for (int j = 0; j < Q; ++j) {
    const int q = blockIdx.x * (BS * Q) + j * BS + threadIdx.x;
    if (q < a.q_count) {
      const size_t o = ((size_t)chunk * a.q_count + q) * K;
#pragma unroll
      for (int k = 0; k < K; ++k) a.part[o + k] = bk[j][k];
    }
  }
}

// merge the per-chunk key lists, apply the distance gate, emit device indices
template <int K>
__global__ __launch_bounds__(256) void nn_merge_kernel(MergeArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.q_count) return;
  u64 bk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) bk[k] = KEY_EMPTY;
  for (int c = 0; c < a.n_chunks; ++c) {
    const size_t o = ((size_t)c * a.q_count + q) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const u64 key = a.part[o + k];
      if (key < bk[K - 1]) key_insert<K>(bk, key);
    }
  }
  emit<K>(bk, a.inv, a.gate_sq, a.out_i, a.out_d, (size_t)(a.q_begin + q) * K);
}

// ------------------------------------------------------------------------------------------
// exact kNN through the Hilbert-ordered 4-ary box tree (bvh.hpp): one query per lane, seed leaf
// for a first bound, then a stackless fixed-order depth-first walk pruned by the float32 box
// distance.  Result sets are order independent (keys), so the output equals brute force bit for
// bit.
// ------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void scan_leaf(const float4* __restrict__ pts, float px, float py, float pz, u64 (&bk)[K], float& wd) {
  float4 t[kLeaf];
#pragma unroll
  for (int p = 0; p < kLeaf; ++p) t[p] = pts[p];  // padded with (+inf, +inf, +inf, -1): no bounds test
#pragma unroll
  for (int p = 0; p < kLeaf; ++p) consider<K>(bk, wd, l2_simple(px, py, pz, t[p].x, t[p].y, t[p].z), __float_as_uint(t[p].w));
}

__device__ __forceinline__ float box_lb(const float4 lo, const float4 hi, float px, float py, float pz) {
  const float ex = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.f);
  const float ey = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.f);
  const float ez = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.f);
  return (ex * ex + ey * ey) + ez * ez;  // lower bound of l2_simple over the box (monotone rounding)
}

// 4-bit mask of the children of node (level, parent) whose box can still hold a neighbour
__device__ __forceinline__ unsigned child_mask(const float4* __restrict__ blo, const float4* __restrict__ bhi, int child_off,
                                               int child_cnt, int parent, float px, float py, float pz, float wd) {
  const int c0 = parent * kFan;
  unsigned m = 0;
  float4 lo[kFan], hi[kFan];
#pragma unroll
  for (int c = 0; c < kFan; ++c) {  // 8 independent loads; indices clamped, validity applied below
    const int node = child_off + min(c0 + c, child_cnt - 1);
    lo[c] = blo[node];
    hi[c] = bhi[node];
  }
#pragma unroll
  for (int c = 0; c < kFan; ++c) {
    const float lb = box_lb(lo[c], hi[c], px, py, pz);
    // lb == wd may still hide an equal distance with a lower caller index: keep it
    if (c0 + c < child_cnt && !(lb > wd)) m |= 1u << c;
  }
  return m;
}

template <int K>
__global__ __launch_bounds__(64) void bvh_knn_kernel(KnnArgs a) {
  __shared__ int s_off[kMaxLevels], s_cnt[kMaxLevels];
  if (threadIdx.x < kMaxLevels) { s_off[threadIdx.x] = a.tree.lv.off[threadIdx.x]; s_cnt[threadIdx.x] = a.tree.lv.cnt[threadIdx.x]; }
  __syncthreads();
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.q_count) return;
  float px, py, pz;
  load_query(a.qx, a.qy, a.qz, a.q_begin + q, a.do_xform, a.M, px, py, pz);
  u64 bk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) bk[k] = KEY_EMPTY;
  float wd = key_dist(KEY_EMPTY);
  const float4* __restrict__ pts = a.tree.pts4 + a.tree.pt_begin;
  const float4* __restrict__ blo = a.tree.box_lo + a.tree.node_begin;
  const float4* __restrict__ bhi = a.tree.box_hi + a.tree.node_begin;
  const int top = a.tree.lv.n_levels - 1;
  const int n_leaf = s_cnt[0];
  int dbg_nodes = 0, dbg_leaves = 0;

  // --- seed: a level-1 node (<= 4 leaves, 64 points) next to the query gives the first bound:
  // the query's own node for the covariance self-query, the node of its curve index otherwise.
  int seed = 0;  // index at level min(1, top)
  if (top >= 1) {
    if (a.self) {
      seed = (q / kLeaf) / kFan;
    } else {
      // locate the query's curve index among the leaves' first indices (last leaf with code <= qc)
      const u64 qc = curve_code(px, py, pz, a.tree.lo[0], a.tree.lo[1], a.tree.lo[2], a.tree.scale);
      const u64* __restrict__ codes = a.tree.leaf_code + a.tree.code_begin;
      int lo_i = 0, hi_i = n_leaf - 1;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i + 1) >> 1;
        if (codes[mid] <= qc) lo_i = mid; else hi_i = mid - 1;
      }
      seed = lo_i / kFan;
    }
    const int l0 = seed * kFan, l1 = min(l0 + kFan, n_leaf);
    for (int l = l0; l < l1; ++l) { scan_leaf<K>(pts + (size_t)l * kLeaf, px, py, pz, bk, wd); ++dbg_leaves; }
  } else {
    scan_leaf<K>(pts, px, py, pz, bk, wd);
  }

  // --- depth-first walk of everything else.  State: the level L whose nodes are being iterated,
  // the index `base` of the first sibling of the current group at L, and one 4-bit mask per level
  // of the siblings still to visit.  "while-while": each lane walks boxes until it holds a leaf,
  // then the wave scans leaves together (the scan is the expensive, divergence-sensitive part).
  if (top >= 2) {
    unsigned masks = 0;  // 4 bits per level, levels 0..top-1 (top <= 13 needs two words)
    unsigned masks_hi = 0;
    auto get = [&](int L) -> unsigned { return L < 8 ? (masks >> (4 * L)) & 15u : (masks_hi >> (4 * (L - 8))) & 15u; };
    auto put = [&](int L, unsigned m) {
      if (L < 8) masks = (masks & ~(15u << (4 * L))) | (m << (4 * L));
      else masks_hi = (masks_hi & ~(15u << (4 * (L - 8)))) | (m << (4 * (L - 8)));
    };
    int L = top - 1, base = 0;
    put(L, child_mask(blo, bhi, s_off[L], s_cnt[L], 0, px, py, pz, wd));
    ++dbg_nodes;
    bool done = false;
    while (!done) {
      int leaf = -1;
      while (leaf < 0 && !done) {
        const unsigned m = get(L);
        if (m == 0) {  // this sibling group is exhausted: back to the parent's group
          if (L == top - 1) { done = true; break; }
          ++L;
          base = (base / kFan) & ~(kFan - 1);
          continue;
        }
        const int c = __ffs(m) - 1;
        put(L, m & (m - 1));
        const int node = base + c;
        if (L == 1 && node == seed) continue;  // already scanned as the seed group
        if (L == 0) {
          leaf = node;
        } else {
          put(L - 1, child_mask(blo, bhi, s_off[L - 1], s_cnt[L - 1], node, px, py, pz, wd));
          ++dbg_nodes;
          --L;
          base = node * kFan;
        }
      }
      if (leaf >= 0) {
        // the bound may have tightened since the mask was computed: re-test before paying for the scan
        const int bn = s_off[0] + leaf;
        if (!(box_lb(blo[bn], bhi[bn], px, py, pz) > wd)) { scan_leaf<K>(pts + (size_t)leaf * kLeaf, px, py, pz, bk, wd); ++dbg_leaves; }
      }
    }
  } else if (top == 1) {
    // two levels: the seed group was one level-1 node == the root; nothing else exists
  }
  emit<K>(bk, a.inv, a.gate_sq, a.out_i, a.out_d, (size_t)(a.q_begin + q) * K);
  if (a.dbg) { a.dbg[2 * q] = dbg_nodes; a.dbg[2 * q + 1] = dbg_leaves; }
}

// Quad-per-query variant of the tree search (the default).  With one query per lane a 100 K-point
// search is only ~1.5 waves per SIMD and a wave lives as long as its slowest lane (measured: mean
// wave 172 us, kernel 1040 us at K = 20).  Here the four lanes of a DPP quad share one query: each
// lane tests one of the four children of a node, scans 4 of a leaf's 16 points into its own partial
// top-K list, and the pruning bound is the quad-minimum of the four lists' K-th distances (each is
// a valid upper bound of the true K-th distance).  4x more waves that are 4x shorter: the tail
// shrinks and there are enough waves to hide memory latency.  The four lists are merged through
// LDS at the end; keys make the result independent of visiting order, so it is bit-identical to
// the other engines.
__device__ __forceinline__ float quad_min(float v) {
  float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
  v = fminf(v, o);
  o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  return fminf(v, o);
}
__device__ __forceinline__ float quad_max(float v) {
  float o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
  v = fmaxf(v, o);
  o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
  return fmaxf(v, o);
}

// Pruning bound of a quad: an upper bound of the query's true K-th smallest distance from the four
// partial lists (each the K best of a quarter of the candidates seen so far).
//   (1) any lane's own K-th distance: its K candidates are among all candidates;
//   (2) max over lanes of the lane's ceil(K/4)-th distance: 4 * ceil(K/4) >= K candidates lie
//       within it.  The quarters are a quasi-random split, so (2) is close to the true K-th.
// +inf stands for "not enough entries yet".
template <int K>
__device__ __forceinline__ float quad_bound(const u64 (&bk)[K]) {
  constexpr int M = (K + 3) / 4;
  const float own_k = bk[K - 1] == KEY_EMPTY ? INFINITY : key_dist(bk[K - 1]);
  const float own_m = bk[M - 1] == KEY_EMPTY ? INFINITY : key_dist(bk[M - 1]);
  return fminf(quad_min(own_k), quad_max(own_m));
}

template <int K>
__device__ __forceinline__ void scan_leaf_quad(const float4* __restrict__ pts, float px, float py, float pz, u64 (&bk)[K], float& wd) {
  float4 t[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) t[p] = pts[4 * p];  // lane s of the quad takes points s, s+4, s+8, s+12: a mixed quarter
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float d = l2_simple(px, py, pz, t[p].x, t[p].y, t[p].z);
    if (!(d > wd)) {  // beyond the quad bound it cannot be among the K nearest
      const u64 key = make_key(d, __float_as_uint(t[p].w));
      if (key < bk[K - 1]) key_insert<K>(bk, key);
    }
  }
  wd = quad_bound<K>(bk);
}

__device__ __forceinline__ unsigned child_mask_quad(const float4* __restrict__ blo, const float4* __restrict__ bhi, int child_off,
                                                    int child_cnt, int parent, int sub, int lane, float px, float py, float pz,
                                                    float wd) {
  const int c = parent * kFan + sub;
  const int node = child_off + min(c, child_cnt - 1);
  const float lb = box_lb(blo[node], bhi[node], px, py, pz);
  const bool ok = c < child_cnt && !(lb > wd);  // lb == wd may hide an equal distance with a lower index
  return (unsigned)(__ballot(ok) >> (lane & ~3)) & 15u;  // the quad's lanes are always in the same control path
}

template <int K>
__global__ __launch_bounds__(64) void bvh_knn_quad_kernel(KnnArgs a) {
  __shared__ int s_off[kMaxLevels], s_cnt[kMaxLevels];
  __shared__ u64 s_merge[16][4][K];
  if (threadIdx.x < kMaxLevels) { s_off[threadIdx.x] = a.tree.lv.off[threadIdx.x]; s_cnt[threadIdx.x] = a.tree.lv.cnt[threadIdx.x]; }
  __syncthreads();
  const int lane = threadIdx.x, sub = lane & 3, slot = lane >> 2;
  const int q_raw = blockIdx.x * 16 + slot;
  const int q = min(q_raw, a.q_count - 1);  // a padding quad repeats the last query and is not emitted
  float px, py, pz;
  load_query(a.qx, a.qy, a.qz, a.q_begin + q, a.do_xform, a.M, px, py, pz);
  u64 bk[K];
  key_list_init<K>(bk);
  float wd = INFINITY;
  const float4* __restrict__ pts = a.tree.pts4 + a.tree.pt_begin + sub;
  const float4* __restrict__ blo = a.tree.box_lo + a.tree.node_begin;
  const float4* __restrict__ bhi = a.tree.box_hi + a.tree.node_begin;
  const int top = a.tree.lv.n_levels - 1;
  const int n_leaf = s_cnt[0];

  // --- seed group: the level-1 node next to the query
  int seed = 0;
  if (top >= 1) {
    if (a.self) {
      seed = (q / kLeaf) / kFan;
    } else {
      const u64 qc = curve_code(px, py, pz, a.tree.lo[0], a.tree.lo[1], a.tree.lo[2], a.tree.scale);
      const u64* __restrict__ codes = a.tree.leaf_code + a.tree.code_begin;
      int lo_i = 0, hi_i = n_leaf - 1;
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i + 1) >> 1;
        if (codes[mid] <= qc) lo_i = mid; else hi_i = mid - 1;
      }
      seed = lo_i / kFan;
    }
    const int l0 = seed * kFan, l1 = min(l0 + kFan, n_leaf);
    for (int l = l0; l < l1; ++l) scan_leaf_quad<K>(pts + (size_t)l * kLeaf, px, py, pz, bk, wd);
  } else {
    scan_leaf_quad<K>(pts, px, py, pz, bk, wd);
  }

  // --- depth-first walk (quad-uniform state), while-while so that the wave scans leaves together
  if (top >= 2) {
    unsigned masks = 0, masks_hi = 0;
    auto get = [&](int lv) -> unsigned { return lv < 8 ? (masks >> (4 * lv)) & 15u : (masks_hi >> (4 * (lv - 8))) & 15u; };
    auto put = [&](int lv, unsigned m) {
      if (lv < 8) masks = (masks & ~(15u << (4 * lv))) | (m << (4 * lv));
      else masks_hi = (masks_hi & ~(15u << (4 * (lv - 8)))) | (m << (4 * (lv - 8)));
    };
    int L = top - 1, base = 0;
    put(L, child_mask_quad(blo, bhi, s_off[L], s_cnt[L], 0, sub, lane, px, py, pz, wd));
    bool done = false;
    while (!done) {
      int leaf = -1;
      while (leaf < 0 && !done) {
        const unsigned m = get(L);
        if (m == 0) {
          if (L == top - 1) { done = true; break; }
          ++L;
          base = (base / kFan) & ~(kFan - 1);
          continue;
        }
        const int c = __ffs(m) - 1;
        put(L, m & (m - 1));
        const int node = base + c;
        if (L == 1 && node == seed) continue;
        if (L == 0) {
          leaf = node;
        } else {
          put(L - 1, child_mask_quad(blo, bhi, s_off[L - 1], s_cnt[L - 1], node, sub, lane, px, py, pz, wd));
          --L;
          base = node * kFan;
        }
      }
      if (leaf >= 0) {
        const int bn = s_off[0] + leaf;
        if (!(box_lb(blo[bn], bhi[bn], px, py, pz) > wd)) scan_leaf_quad<K>(pts + (size_t)leaf * kLeaf, px, py, pz, bk, wd);
      }
    }
  }

  // --- merge the quad's four ascending lists (LDS), lane 0 of the quad emits
#pragma unroll
  for (int k = 0; k < K; ++k) s_merge[slot][sub][k] = bk[k];
  __syncthreads();
  if (sub == 0 && q_raw < a.q_count) {
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    const size_t o = (size_t)(a.q_begin + q) * K;
    for (int k = 0; k < K; ++k) {
      const u64 h0 = p0 < K ? s_merge[slot][0][p0] : KEY_EMPTY, h1 = p1 < K ? s_merge[slot][1][p1] : KEY_EMPTY;
      const u64 h2 = p2 < K ? s_merge[slot][2][p2] : KEY_EMPTY, h3 = p3 < K ? s_merge[slot][3][p3] : KEY_EMPTY;
      const u64 m01 = h0 <= h1 ? h0 : h1, m23 = h2 <= h3 ? h2 : h3;
      const u64 best = m01 <= m23 ? m01 : m23;
      // keys are unique (caller index), except KEY_EMPTY: advance exactly one list
      if (best == h0 && p0 < K) ++p0; else if (best == h1 && p1 < K) ++p1; else if (best == h2 && p2 < K) ++p2; else ++p3;
      const unsigned orig = (unsigned)best;
      const float d = best == KEY_EMPTY ? INFINITY : key_dist(best);
      const bool keep = orig != 0xffffffffu && d < a.gate_sq;  // strict <, float compare (em_icp.hpp:65)
      a.out_i[o + k] = keep ? a.inv[orig] : -1;
      if (a.out_d) a.out_d[o + k] = d;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Packet search (default): the 16 queries of a wave walk the tree TOGETHER.
//
// Queries are consecutive points of a curve-ordered cloud, so the 16 of a wave are neighbours in
// space (for the covariance search they are exactly one leaf) and need almost the same nodes.
// The walk is therefore shared: one wave-uniform depth-first traversal (level, sibling masks and
// node indices live in scalar registers, no divergence), a node is entered when ANY of the 16
// queries still needs it (each query prunes with its own bound).  Lane (query q, sub c) tests child
// c against query q, one ballot folds the 64 answers into the 4-bit sibling mask.  At a leaf every
// quad scans the same 16 points (4 per lane, addresses shared by all quads: one cache line per
// load).  Compared with the per-quad walk above this removes the per-lane stack bookkeeping and the
// "wait for the slowest quad" rounds (profile: 6100 -> VALU instructions per wave).  Lists, keys
// and the final 4-way merge are those of the quad kernel, so the result is bit-identical.
//
// Workgroup b runs on XCD b % 8 (observed dispatch order; a speed assumption only): the block
// index is remapped so that each XCD gets one contiguous run of the curve, i.e. one compact
// region of space, and its private L2 only has to hold that region of the target.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return xcd < r ? xcd * (q + 1) + loc : r * (q + 1) + (xcd - r) * q + loc;
}

__device__ __forceinline__ unsigned child_mask_packet(const float4* __restrict__ blo, const float4* __restrict__ bhi, int child_off,
                                                      int child_cnt, int parent, int sub, float px, float py, float pz, float wd) {
  const int c = parent * kFan + sub;
  const int node = child_off + min(c, child_cnt - 1);
  const float lb = box_lb(blo[node], bhi[node], px, py, pz);
  const bool ok = c < child_cnt && !(lb > wd);  // lb == wd may hide an equal distance with a lower index
  u64 b = __ballot(ok);                         // bit 4*q + c
  b |= b >> 32; b |= b >> 16; b |= b >> 8; b |= b >> 4;
  return (unsigned)b & 15u;
}

// WPB waves (packets) per workgroup: one-wave workgroups are launched too slowly to fill the chip
// (6250 of them at 100K queries: ~1.5 waves per SIMD resident on average)
template <int K, int WPB>
__device__ __forceinline__ void knn_packet_body(const KnnArgs& a, int wg, int n_wg) {
  __shared__ u64 s_merge_all[WPB][16][4][K];
  u64 (&s_merge)[16][4][K] = s_merge_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, sub = lane & 3, slot = lane >> 2;
  const int bid = xcd_contiguous_block(wg, n_wg) * WPB + (int)(threadIdx.x >> 6);
  if (bid * 16 >= a.q_count) return;  // a surplus wave of the last workgroup (no barrier below is block-wide)
  const int q_raw = bid * 16 + slot;
  const int q = min(q_raw, a.q_count - 1);  // a padding quad repeats the last query and is not emitted
  float px, py, pz;
  load_query(a.qx, a.qy, a.qz, a.q_begin + q, a.do_xform, a.M, px, py, pz);
  u64 bk[K];
  key_list_init<K>(bk);
  float wd = INFINITY;
  const float4* __restrict__ pts = a.tree.pts4 + a.tree.pt_begin + sub;
  const float4* __restrict__ blo = a.tree.box_lo + a.tree.node_begin;
  const float4* __restrict__ bhi = a.tree.box_hi + a.tree.node_begin;
  const int top = a.tree.lv.n_levels - 1;
  const int n_leaf = a.tree.lv.cnt[0];

  // --- seed group: the level-1 node at the packet's position on the curve
  int seed = 0;
  if (top >= 1) {
    if (a.self) {
      seed = bid / kFan;  // the 16 queries ARE leaf `bid`
    } else {
      // (1) the previous search of these queries, when there is one: the leaf of a previous nearest
      // neighbour (outer iterations move the pose little) -- one load instead of a curve transform
      // The hint is only trusted while it is still close: the query's distance to its previous
      // nearest neighbour must not have grown beyond twice what it was (after the first solve of an
      // align() the pose jumps, and a stale hint is a worse seed than the curve position).
      int hint_leaf = -1;
      if (a.seed_hint) {
        const size_t hq = (size_t)(a.q_begin + q) * a.hint_K;
        const int prev = a.seed_hint[hq];
        bool ok = prev >= a.t_begin && prev < a.t_begin + a.tree.n;
        if (ok && a.out_d) {
          const float4 hp = a.tree.pts4[a.tree.pt_begin + (prev - a.t_begin)];
          ok = l2_simple(px, py, pz, hp.x, hp.y, hp.z) <= 4.0f * a.out_d[hq] + 1e-12f;
        }
        const u64 m = __ballot(ok);
        if (m) {
          const u64 upper = m >> 32;  // prefer a query from the middle of the packet
          const int src = upper ? 32 + __ffsll((unsigned long long)upper) - 1 : __ffsll((unsigned long long)m) - 1;
          hint_leaf = (__builtin_amdgcn_readlane(prev, src) - a.t_begin) / kLeaf;
        }
      }
      if (hint_leaf >= 0) {
        seed = hint_leaf / kFan;
      } else {
        // (2) binary search of the middle query's (10 bits per axis) curve index in the leaves' first
        // indices, on the scalar unit (uniform addresses: ~13 dependent scalar loads).  A 64-ary
        // ballot search is three rounds instead of thirteen but every round is 64 scattered vector
        // loads per wave: in a batch launch those were +60 % memory transactions and +67 % time.
        const u64 qc_lane = curve_code_coarse<10>(px, py, pz, a.tree.lo[0], a.tree.lo[1], a.tree.lo[2], a.tree.scale);
        const u64 qc = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(qc_lane >> 32), 32) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)qc_lane, 32);
        const u64* __restrict__ codes = a.tree.leaf_code + a.tree.code_begin;
        int lo_i = 0, hi_i = n_leaf - 1;  // last leaf whose first index is <= qc, else 0
        while (lo_i < hi_i) {
          const int mid = (lo_i + hi_i + 1) >> 1;
          if (codes[mid] <= qc) lo_i = mid; else hi_i = mid - 1;
        }
        seed = lo_i / kFan;
      }
    }
    seed = __builtin_amdgcn_readfirstlane(seed);
  }
  {
    const int l0 = seed * kFan, l1 = min(l0 + kFan, n_leaf);  // a one-leaf tree: leaf 0
    if (K >= 16 && l1 - l0 == kFan) {
      // K = 20: the 16 seed candidates of a lane all enter its (empty) list.  Sixteen insertions are
      // 16 x 39 instructions; writing them into the first 16 slots and sorting those with a
      // 63-comparator network is 126 (same list: the keys are unique up to identical padding keys).
      if constexpr (K >= 16) {
#pragma unroll
        for (int l = 0; l < kFan; ++l) {
          const float4* __restrict__ lp = pts + (size_t)(l0 + l) * kLeaf;
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float4 t = lp[4 * p];
            bk[4 * l + p] = make_key(l2_simple(px, py, pz, t.x, t.y, t.z), __float_as_uint(t.w));
          }
        }
        key_sort16<K>(bk);
        wd = quad_bound<K>(bk);
      }
    } else {
#pragma unroll 1
      for (int l = l0; l < l1; ++l) scan_leaf_quad<K>(pts + (size_t)l * kLeaf, px, py, pz, bk, wd);
    }
  }

  // --- shared depth-first walk: all of this state is wave-uniform.  Two nested loops: the inner
  // one only moves through the tree (scalar state, box tests) until it stands on a leaf some
  // query needs, the outer one scans that leaf -- so the K-entry lists are carried by exactly one
  // loop with one back edge (with `continue`s in a single loop the compiler kept up to three
  // copies of the lists alive and moved them at every edge).
  int n_box = 0, n_scan = 0;  // statistics (a.dbg), dead code otherwise
  if (top >= 2) {
    u64 masks = 0;  // 4 sibling bits per level
    int L = top - 1, base = 0;
    masks = (u64)child_mask_packet(blo, bhi, a.tree.lv.off[L], a.tree.lv.cnt[L], 0, sub, px, py, pz, wd) << (4 * L);
    for (;;) {
      int leaf = -1;
      for (;;) {
        const unsigned m = (unsigned)(masks >> (4 * L)) & 15u;
        if (m == 0) {
          if (L == top - 1) break;
          ++L;
          base = (base / kFan) & ~(kFan - 1);
          continue;
        }
        const int c = __ffs(m) - 1;
        masks &= ~(1ull << (4 * L + c));
        const int node = base + c;
        if (L == 1 && node == seed) continue;
        if (L == 0) {
          const int bn = a.tree.lv.off[0] + node;
          const float lb = box_lb(blo[bn], bhi[bn], px, py, pz);  // the bounds may have tightened since the parent's test
          if (__ballot(!(lb > wd)) != 0) { leaf = node; break; }
        } else {
          ++n_box;
          const unsigned cm = child_mask_packet(blo, bhi, a.tree.lv.off[L - 1], a.tree.lv.cnt[L - 1], node, sub, px, py, pz, wd);
          --L;
          masks = (masks & ~(15ull << (4 * L))) | ((u64)cm << (4 * L));
          base = node * kFan;
        }
      }
      if (leaf < 0) break;
      scan_leaf_quad<K>(pts + (size_t)leaf * kLeaf, px, py, pz, bk, wd);
      ++n_scan;
    }
  }
  if (a.dbg && sub == 0 && q_raw < a.q_count) { a.dbg[2 * q] = n_box; a.dbg[2 * q + 1] = n_scan; }

  // --- merge the quad's four ascending lists (LDS), lane 0 of the quad emits
#pragma unroll
  for (int k = 0; k < K; ++k) s_merge[slot][sub][k] = bk[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes, read below by its lanes 0 mod 4
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (sub == 0 && q_raw < a.q_count) {
    int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    // [query][K], or [K][out_stride] (consecutive queries -> consecutive addresses)
    const size_t o = a.out_stride > 0 ? (size_t)(a.q_begin + q) : (size_t)(a.q_begin + q) * K;
    const size_t ks = a.out_stride > 0 ? (size_t)a.out_stride : 1;
    for (int k = 0; k < K; ++k) {
      const u64 h0 = p0 < K ? s_merge[slot][0][p0] : KEY_EMPTY, h1 = p1 < K ? s_merge[slot][1][p1] : KEY_EMPTY;
      const u64 h2 = p2 < K ? s_merge[slot][2][p2] : KEY_EMPTY, h3 = p3 < K ? s_merge[slot][3][p3] : KEY_EMPTY;
      const u64 m01 = h0 <= h1 ? h0 : h1, m23 = h2 <= h3 ? h2 : h3;
      const u64 best = m01 <= m23 ? m01 : m23;
      if (best == h0 && p0 < K) ++p0; else if (best == h1 && p1 < K) ++p1; else if (best == h2 && p2 < K) ++p2; else ++p3;
      const unsigned orig = (unsigned)best;
      const float d = best == KEY_EMPTY ? INFINITY : key_dist(best);
      const bool keep = orig != 0xffffffffu && d < a.gate_sq;  // strict <, float compare (em_icp.hpp:65)
      a.out_i[o + k * ks] = keep ? a.inv[orig] : -1;
      if (a.out_d) a.out_d[o + k * ks] = d;
    }
  }
}

template <int K, int WPB>
__global__ __launch_bounds__(64 * WPB) void bvh_knn_packet_kernel(KnnArgs a) {
  knn_packet_body<K, WPB>(a, (int)blockIdx.x, (int)gridDim.x);
}

// Several searches in one launch (lock-step batch: all pairs' searches of a phase): blockIdx.y picks
// the job from an array passed BY VALUE -- kernel arguments keep their pointers typed as HBM and
// are read with scalar loads.  The jobs' long tails overlap inside the one launch.
template <int K, int WPB>
__global__ __launch_bounds__(64 * WPB) void bvh_knn_packet_jobs_kernel(KnnJobs jobs) {
  const KnnArgs& a = jobs.job[blockIdx.y];
  const int n_wg = ((a.q_count + 15) / 16 + WPB - 1) / WPB;
  if ((int)blockIdx.x >= n_wg) return;
  knn_packet_body<K, WPB>(a, (int)blockIdx.x, n_wg);
}

// ------------------------------------------------------------------------------------------
// covariance / normal / label histogram from the k-neighbour lists   (em_icp.hpp:298-340)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_rotate(double (&A)[3][3], double (&V)[3][3], int p, int q) {
  const double apq = A[p][q];
  if (apq == 0.0) return;
  const double tau = (A[q][q] - A[p][p]) / (2.0 * apq);
  const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double akp = A[k][p], akq = A[k][q];
    A[k][p] = c * akp - s * akq;
    A[k][q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double apk = A[p][k], aqk = A[q][k];
    A[p][k] = c * apk - s * aqk;
    A[q][k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double vkp = V[k][p], vkq = V[k][q];
    V[k][p] = c * vkp - s * vkq;
    V[k][q] = s * vkp + c * vkq;
  }
}

__device__ __forceinline__ void cov_body(const CovArgs& a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  double mean0 = 0, mean1 = 0, mean2 = 0;
  double c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
  const int* nn = a.nn_stride > 0 ? a.nn + i : a.nn + (size_t)i * a.k;
  const size_t js = a.nn_stride > 0 ? (size_t)a.nn_stride : 1;
  for (int j = 0; j < a.k; ++j) {
    const int g = nn[j * js];
    if (g < 0) continue;
    const float x = a.x[g], y = a.y[g], z = a.z[g];
    mean0 += (double)x; mean1 += (double)y; mean2 += (double)z;
    if (a.float_products) {
      // quirk Q2: pt.y*pt.x is a float32 product (em_icp.hpp:307-314)
      c00 += (double)__fmul_rn(x, x);
      c10 += (double)__fmul_rn(y, x);
      c11 += (double)__fmul_rn(y, y);
      c20 += (double)__fmul_rn(z, x);
      c21 += (double)__fmul_rn(z, y);
      c22 += (double)__fmul_rn(z, z);
    } else {
      const double dx = x, dy = y, dz = z;
      c00 += dx * dx; c10 += dy * dx; c11 += dy * dy;
      c20 += dz * dx; c21 += dz * dy; c22 += dz * dz;
    }
  }
  // quirk Q3: divide by k whatever the neighbour count (em_icp.hpp:317,320)
  const double kk = (double)a.k;
  mean0 /= kk; mean1 /= kk; mean2 /= kk;
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  A[0][0] = __dsub_rn(c00 / kk, __dmul_rn(mean0, mean0));
  A[1][0] = A[0][1] = __dsub_rn(c10 / kk, __dmul_rn(mean1, mean0));
  A[1][1] = __dsub_rn(c11 / kk, __dmul_rn(mean1, mean1));
  A[2][0] = A[0][2] = __dsub_rn(c20 / kk, __dmul_rn(mean2, mean0));
  A[2][1] = A[1][2] = __dsub_rn(c21 / kk, __dmul_rn(mean2, mean1));
  A[2][2] = __dsub_rn(c22 / kk, __dmul_rn(mean2, mean2));
  // stand-in for Eigen::JacobiSVD(ComputeFullU) on a symmetric matrix: cyclic Jacobi;
  // singular values = |eigenvalues|, the "normal" is the column of smallest |eigenvalue|
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    jacobi_rotate(A, V, 0, 1);
    jacobi_rotate(A, V, 0, 2);
    jacobi_rotate(A, V, 1, 2);
  }
  const double e0 = fabs(A[0][0]), e1 = fabs(A[1][1]), e2 = fabs(A[2][2]);
  // last column after a stable descending sort by |eigenvalue| (ties keep the later column)
  int col = 0;
  double em = e0;
  if (e1 <= em) { em = e1; col = 1; }
  if (e2 <= em) { em = e2; col = 2; }
  double nx = col == 0 ? V[0][0] : (col == 1 ? V[0][1] : V[0][2]);
  double ny = col == 0 ? V[1][0] : (col == 1 ? V[1][1] : V[1][2]);
  double nz = col == 0 ? V[2][0] : (col == 1 ? V[2][1] : V[2][2]);
  a.nx[i] = nx; a.ny[i] = ny; a.nz[i] = nz;
  if (a.hist) {
    // label histogram as neighbour counts (em_icp.hpp:301: dist(label-1) += 1/k)
    uint8_t* h = a.hist + (size_t)i * a.C;  // this lane owns the row
    for (int c = 0; c < a.C; ++c) h[c] = 0;
    for (int j = 0; j < a.k; ++j) {
      const int g = nn[j * js];
      if (g < 0) continue;
      const uint32_t l = a.label[g];
      if (l >= 1u && l <= (uint32_t)a.C) h[l - 1] = (uint8_t)(h[l - 1] + 1);
    }
  }
}

// per-point projections of the label distribution through the confusion matrix:
//   proj[i][s] = dist_i^T * CM[:, s]   (the two factors of em_icp.hpp:86-87), dist = counts * 1/k
// accumulated over r in ascending order exactly like the reference's dot product.  Computed once per
// align() per cloud, so the per-correspondence weight is a C-term product-sum of two such rows.
__global__ __launch_bounds__(256) void cov_kernel(CovArgs a) { cov_body(a); }
__global__ __launch_bounds__(256) void cov_jobs_kernel(CovJobs jobs) { cov_body(jobs.job[blockIdx.y]); }

__device__ __forceinline__ void proj_body(const ProjArgs& a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n * a.C) return;
  const int i = e / a.C, s = e - i * a.C;
  const uint8_t* h = a.hist + (size_t)i * a.C;
  double temp = 0.0;
  for (int r = 0; r < a.C; ++r) temp += a.hval[h[r]] * a.cm[r * a.C + s];
  a.proj[e] = temp;
}

// ------------------------------------------------------------------------------------------
// per-correspondence math (closed form of gicp_cost_function.h:31-70
// chained with Sophus' Dx_this_mul_exp_x_at_0, for C = I - (1-eps) n n^T)
// ------------------------------------------------------------------------------------------
struct Corr {
  double r;        // res^T A^-1 res  (the Ceres residual: squared Mahalanobis distance)
  double J[6];     // d r / d delta, T*exp(delta), delta = [upsilon; omega]
  double detA;
};

template <bool WANT_J>
__device__ __forceinline__ void corr_eval(const Pose& P, double one_m_eps, double psx, double psy,
                                          double psz, double nsx, double nsy, double nsz,
                                          double ptx, double pty, double ptz, double ntx,
                                          double nty, double ntz, Corr& o) {
  // pure float64 algebra with tolerance-level parity (1e-9): fused multiply-adds are welcome here,
  // unlike in the float32 distance / transform code whose rounding must match the reference's
#pragma clang fp contract(fast)
  const double* R = P.R;
  const double mx = R[0] * nsx + R[1] * nsy + R[2] * nsz;
  const double my = R[3] * nsx + R[4] * nsy + R[5] * nsz;
  const double mz = R[6] * nsx + R[7] * nsy + R[8] * nsz;
  // A = C_t + R C_s R^T = 2I - (1-eps)(n_t n_t^T + m m^T)
  const double a00 = 2.0 - one_m_eps * (ntx * ntx + mx * mx);
  const double a01 = -one_m_eps * (ntx * nty + mx * my);
  const double a02 = -one_m_eps * (ntx * ntz + mx * mz);
  const double a11 = 2.0 - one_m_eps * (nty * nty + my * my);
  const double a12 = -one_m_eps * (nty * ntz + my * mz);
  const double a22 = 2.0 - one_m_eps * (ntz * ntz + mz * mz);
  const double rx = ptx - (R[0] * psx + R[1] * psy + R[2] * psz + P.t[0]);
  const double ry = pty - (R[3] * psx + R[4] * psy + R[5] * psz + P.t[1]);
  const double rz = ptz - (R[6] * psx + R[7] * psy + R[8] * psz + P.t[2]);
  // Eigen Matrix3d::inverse(): cofactors / determinant
  const double k00 = a11 * a22 - a12 * a12;
  const double k01 = a02 * a12 - a01 * a22;
  const double k02 = a01 * a12 - a02 * a11;
  const double k11 = a00 * a22 - a02 * a02;
  const double k12 = a01 * a02 - a00 * a12;
  const double k22 = a00 * a11 - a01 * a01;
  const double det = a00 * k00 + a01 * k01 + a02 * k02;
  const double inv = 1.0 / det;
  const double ax = inv * (k00 * rx + k01 * ry + k02 * rz);
  const double ay = inv * (k01 * rx + k11 * ry + k12 * rz);
  const double az = inv * (k02 * rx + k12 * ry + k22 * rz);
  o.r = rx * ax + ry * ay + rz * az;
  o.detA = det;
  if (WANT_J) {
    const double bx = R[0] * ax + R[3] * ay + R[6] * az;  // b = R^T a
    const double by = R[1] * ax + R[4] * ay + R[7] * az;
    const double bz = R[2] * ax + R[5] * ay + R[8] * az;
    const double nb = one_m_eps * (nsx * bx + nsy * by + nsz * bz);
    const double cx = psx + bx - nb * nsx;  // c = p_s + C_s b
    const double cy = psy + by - nb * nsy;
    const double cz = psz + bz - nb * nsz;
    o.J[0] = -2.0 * bx; o.J[1] = -2.0 * by; o.J[2] = -2.0 * bz;
    o.J[3] = 2.0 * (by * cz - bz * cy);
    o.J[4] = 2.0 * (bz * cx - bx * cz);
    o.J[5] = 2.0 * (bx * cy - by * cx);
  }
}

// 1/d for a normal, finite d: v_rcp_f64 (2^-26) + two Newton steps (~1 ulp; 5 instructions, the
// correctly rounded division sequence is 14)
__device__ __forceinline__ double rcp_newton(double d) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(d);
  r = r + r * (1.0 - d * r);
  r = r + r * (1.0 - d * r);
  return r;
}

// The accumulate kernels' form of corr_eval<true>: everything that only depends on the SOURCE point
// (shared by the K = 4 or 20 slots of one source point) is computed once per group of four slots.
struct SrcTerms {
  double qx, qy, qz;                      // R p_s + t
  double mx, my, mz;                      // m = R n_s
  double d00, d11, d22, n01, n02, n12;    // 2I - (1-eps) m m^T: diagonal, and the (negative) off-diagonal entries
};

__device__ __forceinline__ void src_terms(const Pose& P, double one_m_eps, double psx, double psy, double psz, double nsx,
                                          double nsy, double nsz, SrcTerms& s) {
#pragma clang fp contract(fast)
  const double* R = P.R;
  s.qx = R[0] * psx + R[1] * psy + R[2] * psz + P.t[0];
  s.qy = R[3] * psx + R[4] * psy + R[5] * psz + P.t[1];
  s.qz = R[6] * psx + R[7] * psy + R[8] * psz + P.t[2];
  s.mx = R[0] * nsx + R[1] * nsy + R[2] * nsz;
  s.my = R[3] * nsx + R[4] * nsy + R[5] * nsz;
  s.mz = R[6] * nsx + R[7] * nsy + R[8] * nsz;
  const double ex = one_m_eps * s.mx, ey = one_m_eps * s.my, ez = one_m_eps * s.mz;
  s.d00 = 2.0 - ex * s.mx; s.n01 = -(ex * s.my); s.n02 = -(ex * s.mz);
  s.d11 = 2.0 - ey * s.my; s.n12 = -(ey * s.mz); s.d22 = 2.0 - ez * s.mz;
}

__device__ __forceinline__ void corr_eval_src(const Pose& P, double one_m_eps, const SrcTerms& s, double psx, double psy,
                                              double psz, double nsx, double nsy, double nsz, double ptx, double pty,
                                              double ptz, double ntx, double nty, double ntz, Corr& o) {
#pragma clang fp contract(fast)
  const double* R = P.R;
  // A = C_t + R C_s R^T = 2I - (1-eps) n_t n_t^T - (1-eps) m m^T
  const double fx = one_m_eps * ntx, fy = one_m_eps * nty, fz = one_m_eps * ntz;
  const double a00 = s.d00 - fx * ntx;
  const double a01 = s.n01 - fx * nty;
  const double a02 = s.n02 - fx * ntz;
  const double a11 = s.d11 - fy * nty;
  const double a12 = s.n12 - fy * ntz;
  const double a22 = s.d22 - fz * ntz;
  const double rx = ptx - s.qx, ry = pty - s.qy, rz = ptz - s.qz;
  // Eigen Matrix3d::inverse(): cofactors / determinant
  const double k00 = a11 * a22 - a12 * a12;
  const double k01 = a02 * a12 - a01 * a22;
  const double k02 = a01 * a12 - a02 * a11;
  const double k11 = a00 * a22 - a02 * a02;
  const double k12 = a01 * a02 - a00 * a12;
  const double k22 = a00 * a11 - a01 * a01;
  const double det = a00 * k00 + a01 * k01 + a02 * k02;
  const double inv = rcp_newton(det);  // det in [~eps^2, 8]
  const double ax = inv * (k00 * rx + k01 * ry + k02 * rz);
  const double ay = inv * (k01 * rx + k11 * ry + k12 * rz);
  const double az = inv * (k02 * rx + k12 * ry + k22 * rz);
  o.r = rx * ax + ry * ay + rz * az;
  o.detA = det;
  const double bx = R[0] * ax + R[3] * ay + R[6] * az;  // b = R^T a
  const double by = R[1] * ax + R[4] * ay + R[7] * az;
  const double bz = R[2] * ax + R[5] * ay + R[8] * az;
  const double nb = one_m_eps * (nsx * bx + nsy * by + nsz * bz);
  const double cx = psx + bx - nb * nsx;  // c = p_s + C_s b
  const double cy = psy + by - nb * nsy;
  const double cz = psz + bz - nb * nsz;
  // HALF the Jacobian: J = 2 [-b; b x c].  The caller folds the powers of two into its weight
  // (scaling by 2 and 4 is exact, so the sums keep their bits) and saves six multiplications.
  o.J[0] = -bx; o.J[1] = -by; o.J[2] = -bz;
  o.J[3] = by * cz - bz * cy;
  o.J[4] = bz * cx - bx * cz;
  o.J[5] = bx * cy - by * cx;
}

// log(x) for finite x >= 1 -- the only arguments the losses produce (1 + s/a^2 and 1 + sqrt(s)/a^2).
// The classic argument-reduction + odd-polynomial scheme of fdlibm's e_log.c (x = 2^k m,
// f = m - 1, s = f / (2 + f), log(1+f) = 2s + s R(s^2) ...), < 1 ulp, with the division replaced by
// v_rcp_f64 + two Newton steps and none of the library routine's special cases: ~35 instructions
// instead of ~80.  The logarithm was 40 % of the accumulate kernel's instructions.
__device__ __forceinline__ double log_ge1(double x) {
#pragma clang fp contract(fast)
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  k = low ? k - 1 : k;
  const double f = m - 1.0, d = 2.0 + f, dk = (double)k;
  const double r = rcp_newton(d);
  const double sq = f * r, z = sq * sq, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// rho0 / rho1 of the reference's loss stacks at s = r^2 (em_icp.hpp:109-117, gicp.hpp:98-104,
// semantic_icp.hpp:96; Ceres CauchyLoss/ScaledLoss/ComposedLoss, sqloss.h); b = a^2, c = 1/b.  rho2 < 0
// for all of them, so Ceres' Corrector scales residual and Jacobian by sqrt(rho1).  g0 = sqrt(v) and
// g1 = 1 / (2 g0) both come from one reciprocal square root (a square root and a division less per
// correspondence; ~1 ulp), the logarithm is log_ge1.
__device__ __forceinline__ void loss_eval_acc(const LossArgs& L, double b, double c, double s, double w, double& rho0, double& rho1) {
#pragma clang fp contract(fast)
  if (L.use_sqloss) {
    const double v = s + 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
    const double y = rsqrt(v);
    const double g0 = v * y, g1 = 0.5 * y;
    const double sum = 1.0 + g0 * c, invs = rcp_newton(sum);
    rho0 = w * (b * log_ge1(sum));
    rho1 = (w * fmax(2.2250738585072014e-308, invs)) * g1;
  } else {
    const double sum = 1.0 + s * c, invs = rcp_newton(sum);
    rho0 = b * log_ge1(sum);
    rho1 = fmax(2.2250738585072014e-308, invs);
  }
}

// ------------------------------------------------------------------------------------------
// EM weight: label posterior from the confusion matrix x the (bool) geometric gate
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_kernel(ProjArgs a) { proj_body(a); }
__global__ __launch_bounds__(256) void proj_jobs_kernel(ProjJobs jobs) { proj_body(jobs.job[blockIdx.y]); }

__device__ __forceinline__ void em_weight_body(const WeightArgs& a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_s * a.K) return;
  const int i = e / a.K;
  const int j = a.idx[e];
  if (j < 0) { a.w[e] = 0.0; return; }
  // em_icp.hpp:84-89 with the two dot products of each term taken from the per-point projections
  const double* __restrict__ ps = a.s_proj + (size_t)i * a.C;
  const double* __restrict__ pt = a.t_proj + (size_t)j * a.C;
  double prob = 0.0;
  for (int s = 0; s < a.C; ++s) {
    double temp = pt[s];
    temp *= ps[s];
    prob += temp;
  }
  // em_icp.hpp:108 -> gicp_cost_function.h:75-87
  Corr c;
  corr_eval<false>(a.pose, a.one_m_eps, a.sx[i], a.sy[i], a.sz[i], a.snx[i], a.sny[i], a.snz[i],
                   a.tx[j], a.ty[j], a.tz[j], a.tnx[j], a.tny[j], a.tnz[j], c);
  const double two_pi = 6.283185307179586;
  const double probability = pow(two_pi * two_pi * two_pi * c.detA, -0.5) * exp(-0.5 * c.r);
  if (a.bool_probability) {
    prob *= (probability != 0.0) ? 1.0 : 0.0;  // quirk Q1: double -> bool (NaN -> true)
  } else {
    prob *= probability;
  }
  a.w[e] = prob;
}

__global__ __launch_bounds__(256) void em_weight_kernel(WeightArgs a) { em_weight_body(a); }
__global__ __launch_bounds__(256) void em_weight_jobs_kernel(WeightJobs jobs) { em_weight_body(jobs.job[blockIdx.y]); }

// ------------------------------------------------------------------------------------------
// accumulate: 28 doubles = [H upper 21 | g 6 | cost] over all correspondence slots
// ------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave without touching the LDS crossbar: four DPP butterfly steps
// inside each 16-lane row (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four
// row sums are read into SGPRs and added.  Every lane returns the same value; the order of
// additions is fixed, so the result is run-to-run reproducible.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm:[1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm:[2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

#define SICP_GLOBAL __attribute__((address_space(1)))

// the per-lane part of one evaluation: groups of 4 slots, loads first (see accumulate_kernel)
// the pose is the same in every lane: held in scalar registers it costs no VGPRs (24 otherwise)
__device__ __forceinline__ void pose_to_sgprs(Pose& P) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = readlane_f64(P.R[k], 0);
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = readlane_f64(P.t[k], 0);
}

// One group = 4 consecutive slots.  A lane issues the loads of TWO groups (its grid-stride
// neighbours) before it computes the first: the kernel runs at two waves per SIMD with ~70 spare
// VGPRs, and an index -> gather chain from HBM is ~2 us, about the time one group takes to compute.
// PF (prefetch depth inside a group) is kept for the chained kernel experiments; the slots are always
// accumulated in ascending order, so every variant produces the same bits.
template <int K>
struct SlotGroup {
  int j[4];
  double w[4];
  float sx[4], sy[4], sz[4], tx[4], ty[4], tz[4];
  double snx[4], sny[4], snz[4], tnx[4], tny[4], tnz[4];
};

template <int K, int BS, int PF = 4>
__device__ __forceinline__ void accumulate_groups(const AccArgs& a, const Pose& P, int block, int nb, double (&acc)[28]) {
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  // every array is HBM: typed as such, the loads are global_load even when the pointers themselves
  // were fetched from memory (batch form), where the compiler would otherwise emit flat_load
  const SICP_GLOBAL int* idx = (const SICP_GLOBAL int*)a.idx;
  const SICP_GLOBAL double* wgt = (const SICP_GLOBAL double*)a.w;
  const SICP_GLOBAL float *sx = (const SICP_GLOBAL float*)a.sx, *sy = (const SICP_GLOBAL float*)a.sy, *sz = (const SICP_GLOBAL float*)a.sz;
  const SICP_GLOBAL float *tx = (const SICP_GLOBAL float*)a.tx, *ty = (const SICP_GLOBAL float*)a.ty, *tz = (const SICP_GLOBAL float*)a.tz;
  const SICP_GLOBAL double *g_snx = (const SICP_GLOBAL double*)a.snx, *g_sny = (const SICP_GLOBAL double*)a.sny, *g_snz = (const SICP_GLOBAL double*)a.snz;
  const SICP_GLOBAL double *g_tnx = (const SICP_GLOBAL double*)a.tnx, *g_tny = (const SICP_GLOBAL double*)a.tny, *g_tnz = (const SICP_GLOBAL double*)a.tnz;
  const int total = a.n_s * K;
  const int n_groups = (total + 3) >> 2;
  const double loss_b = a.loss.cauchy_a * a.loss.cauchy_a, loss_c = 1.0 / loss_b;

  auto load = [&](int g, SlotGroup<K>& G) {
    const int e0 = g << 2;
    if (e0 + 3 < total) {
      typedef int v4i __attribute__((ext_vector_type(4)));
      const v4i jv = *(const SICP_GLOBAL v4i*)(idx + e0);
      G.j[0] = jv.x; G.j[1] = jv.y; G.j[2] = jv.z; G.j[3] = jv.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) G.j[c] = e0 + c < total ? idx[e0 + c] : -1;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = e0 + c;
      const int i = min(e / K, a.n_s - 1);
      const int jj = max(G.j[c], 0);
      G.w[c] = wgt ? wgt[min(e, total - 1)] : 1.0;
      if (K % 4 != 0 || c == 0) {
        G.sx[c] = sx[i]; G.sy[c] = sy[i]; G.sz[c] = sz[i];
        G.snx[c] = g_snx[i]; G.sny[c] = g_sny[i]; G.snz[c] = g_snz[i];
      } else {  // K a multiple of 4: the four slots of a group share one source point
        G.sx[c] = G.sx[0]; G.sy[c] = G.sy[0]; G.sz[c] = G.sz[0];
        G.snx[c] = G.snx[0]; G.sny[c] = G.sny[0]; G.snz[c] = G.snz[0];
      }
      G.tx[c] = tx[jj]; G.ty[c] = ty[jj]; G.tz[c] = tz[jj];
      G.tnx[c] = g_tnx[jj]; G.tny[c] = g_tny[jj]; G.tnz[c] = g_tnz[jj];
    }
  };
  auto compute = [&](const SlotGroup<K>& G) {
    SrcTerms st;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma clang fp contract(fast)
      if (K % 4 != 0 || c == 0) src_terms(P, a.one_m_eps, G.sx[c], G.sy[c], G.sz[c], G.snx[c], G.sny[c], G.snz[c], st);
      // A gated-out slot (index -1) is evaluated on target 0 and weighted by exactly zero instead of
      // being branched around: x + (+-0 * finite) == x bit for bit, and a divergent skip makes the
      // compiler copy all 28 accumulators at the join (10 % of the kernel's instructions).
      Corr cr;
      corr_eval_src(P, a.one_m_eps, st, G.sx[c], G.sy[c], G.sz[c], G.snx[c], G.sny[c], G.snz[c], G.tx[c], G.ty[c], G.tz[c], G.tnx[c],
                    G.tny[c], G.tnz[c], cr);
      double rho0, rho1;
      loss_eval_acc(a.loss, loss_b, loss_c, cr.r * cr.r, G.w[c], rho0, rho1);
      if (G.j[c] < 0) { rho0 = 0.0; rho1 = 0.0; }
      // cr.J is J/2:  rho1 J J^T = (4 rho1) (J/2)(J/2)^T,  rho1 r J = (4 rho1) (J/2) (r/2)
      const double rho4 = 4.0 * rho1, rh = 0.5 * cr.r;
      int o = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const double jp = rho4 * cr.J[p];
#pragma unroll
        for (int q = p; q < 6; ++q) acc[o++] += jp * cr.J[q];
        acc[21 + p] += jp * rh;
      }
      acc[27] += 0.5 * rho0;
    }
  };

  const int stride = nb * BS;
  for (int g = block * BS + threadIdx.x; g < n_groups; g += (PF == 8 ? 2 : 1) * stride) {
    SlotGroup<K> A;
    load(g, A);
    if (PF == 8) {
      SlotGroup<K> B;
      const bool two = g + stride < n_groups;
      load(two ? g + stride : g, B);  // a lane without a second group re-reads its first, with zero weight:
      if (!two) { B.j[0] = -1; B.j[1] = -1; B.j[2] = -1; B.j[3] = -1; }  // no branch around loads or sums
      compute(A);
      compute(B);
    } else {
      compute(A);
    }
  }
}

// One lane handles groups of 4 consecutive slots: all index / weight / point / normal loads of the
// group are issued before the first residual is computed (4 independent gather chains in flight
// per lane instead of one), then the 28 partial sums are combined across the block through an
// LDS transpose so that each wave only performs 7 cross-lane reductions.
//
// FUSED (device-resident solve): the last block to finish -- decided by an arrival ticket -- also
// sums the per-block partials (four waves, seven rows each, same fixed order as reduce_partials)
// and advances the LM machine (lm.hpp: lm_feed) in its lane 0, so one LM evaluation is ONE kernel
// and one launch boundary instead of two.
template <int K, int BS, bool FUSED>
__global__ __launch_bounds__(BS) void accumulate_kernel(AccArgs a) {
  __shared__ double red[28][BS];
  Pose P;
  if (a.lm) {
    // device-resident solve: the pose to evaluate lives in the LM state; once the solve has
    // finished, the launches still queued behind it do nothing (uniform exit)
    if (a.lm->status != LM_RUNNING) return;
    se3::rotation(a.lm->pose, P.R);
    P.t[0] = a.lm->pose[4]; P.t[1] = a.lm->pose[5]; P.t[2] = a.lm->pose[6];
  } else {
    P = a.pose;
  }
  pose_to_sgprs(P);
  double acc[28];
  accumulate_groups<K, BS>(a, P, (int)blockIdx.x, (int)gridDim.x, acc);
  // block reduction: transpose through LDS, then wave w owns outputs w, w + BS/64, ...
#pragma unroll
  for (int k = 0; k < 28; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  constexpr int NW = BS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < 28; k += NW) {
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NW; ++t) sum += red[k][lane + 64 * t];
    sum = wave_sum(sum);
    if (lane == 0) {  // [28][blocks]: coalesced for the reducer
      double* dst = a.partials + (size_t)k * gridDim.x + blockIdx.x;
      if constexpr (FUSED)  // device-scope store: written through to where every XCD sees it
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__double_as_longlong(sum), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      else
        *dst = sum;
    }
  }
  if constexpr (FUSED) {
    // Arrival ticket WITHOUT fences: a release/acquire fence at device scope writes back and
    // invalidates the XCD's whole L2 (measured: 2x slower, the other blocks lose the cloud).  The
    // partials are device-scope atomic stores and loads (sc1: coherent across XCDs by themselves),
    // so it is enough that a block's stores have completed (vmcnt(0)) before it takes its ticket.
    __shared__ unsigned s_ticket;
    __shared__ double s_out[28];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    const int nb = (int)gridDim.x;
    for (int k = wave; k < 28; k += NW) {
      const unsigned long long* __restrict__ row = reinterpret_cast<const unsigned long long*>(a.partials) + (size_t)k * nb;
      double s = 0.0;
      for (int b0 = lane; b0 < nb; b0 += 64 * 8) {  // 8 trips of loads in flight, summed in trip order
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int b = b0 + 64 * t;
          v[t] = b < nb ? __longlong_as_double((long long)__hip_atomic_load(row + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (b0 + 64 * t < nb) s += v[t];
      }
      s = wave_sum(s);
      if (lane == 0) s_out[k] = s;
    }
    // the 0.8 KB state goes HBM -> LDS (all lanes, one round trip) -> lane 0's registers, and back
    // the same way.  Lane 0 needs it in registers: it runs alone, so every LDS or HBM access
    // inside lm_feed would be an exposed latency (measured: +5 us with the state left in LDS).
    __shared__ double s_state[(sizeof(LmState) + 7) / 8];
    constexpr int NSTATE = (int)((sizeof(LmState) + 7) / 8);
    static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as doubles");
    double* g_state = reinterpret_cast<double*>(a.lm_step);
    for (int i = threadIdx.x; i < NSTATE; i += BS) s_state[i] = g_state[i];
    __syncthreads();
    if (wave == 0) {
      if (lane == 0) {
        double o[28];
#pragma unroll
        for (int k = 0; k < 28; ++k) o[k] = s_out[k];
        LmState st = *reinterpret_cast<const LmState*>(s_state);
        lm_feed(st, o);
        *reinterpret_cast<LmState*>(s_state) = st;
      }
      // same wave: lane 0's LDS writes are ordered before these reads
      for (int i = lane; i < NSTATE; i += 64) g_state[i] = s_state[i];
      if (lane == 0) *a.ticket = 0u;  // the next launch starts counting from zero
    }
  }
}

// Chained device-resident solve: one kernel per LM evaluation and nothing in between.
//
// Launch n reads state[n & 1] and the partials of launch n-1, and EVERY block first finishes that
// previous evaluation itself: it sums the partials (four waves, seven rows each, the fixed order of
// reduce_partials) and runs lm_feed in its lane 0 -- all blocks compute the same bits, so all of
// them know the next pose without a second kernel, a grid barrier or a fence ("combine in the next
// kernel's prologue").  Then the block accumulates its share of the new evaluation into
// partials[(n+1) & 1]; block 0 also publishes the advanced state as state[(n+1) & 1] (double
// buffering: other blocks of this launch may still be reading state[n & 1]).  The loads of the
// block's first slots do not depend on the pose and are in flight while lane 0 steps the solver.
template <int K, int BS>
__global__ __launch_bounds__(BS) void accumulate_chain_kernel(AccArgs a) {
  __shared__ double red[28][BS];
  constexpr int NSTATE = (int)(sizeof(LmState) / 8);
  static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as doubles");
  __shared__ double s_state[NSTATE];
  __shared__ double s_out[28];
  constexpr int NW = BS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = (int)gridDim.x;
  const double* __restrict__ g_in = reinterpret_cast<const double*>(a.lm);
  for (int i = threadIdx.x; i < NSTATE; i += BS) s_state[i] = g_in[i];
  // the previous launch's partials: issued before anybody looks at the state (one round trip
  // for both); harmless when there is nothing pending
  {
    for (int k = wave; k < 28; k += NW) {
      const double* __restrict__ row = a.partials_in + (size_t)k * nb;
      double s = 0.0;
      for (int b0 = lane; b0 < nb; b0 += 64 * 4) {
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = b0 + 64 * t < nb ? row[b0 + 64 * t] : 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (b0 + 64 * t < nb) s += v[t];
      }
      s = wave_sum(s);
      if (lane == 0) s_out[k] = s;
    }
  }
  __syncthreads();
  LmState* S = reinterpret_cast<LmState*>(s_state);
  if (S->status == LM_RUNNING && S->pending) {
    if (threadIdx.x == 0) {
      double o[28];
#pragma unroll
      for (int k = 0; k < 28; ++k) o[k] = s_out[k];
      LmState st = *S;
      lm_feed(st, o);
      st.pending = 0;
      *S = st;
    }
    __syncthreads();
  }
  double* g_out = reinterpret_cast<double*>(a.lm_step);
  if (S->status != LM_RUNNING) {  // finished (now or earlier): hand the state on, nothing to evaluate
    if (blockIdx.x == 0)
      for (int i = threadIdx.x; i < NSTATE; i += BS) g_out[i] = s_state[i];
    return;
  }
  Pose P;
  se3::rotation(S->pose, P.R);
  P.t[0] = S->pose[4]; P.t[1] = S->pose[5]; P.t[2] = S->pose[6];
  pose_to_sgprs(P);
  double acc[28];
  accumulate_groups<K, BS>(a, P, (int)blockIdx.x, nb, acc);
#pragma unroll
  for (int k = 0; k < 28; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int k = wave; k < 28; k += NW) {
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NW; ++t) sum += red[k][lane + 64 * t];
    sum = wave_sum(sum);
    if (lane == 0) a.partials[(size_t)k * nb + blockIdx.x] = sum;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) S->pending = 1;
    __syncthreads();
    for (int i = threadIdx.x; i < NSTATE; i += BS) g_out[i] = s_state[i];
  }
}

// Lock-step batch of independent pairs (sicp_align_batch): blockIdx.y selects the pair, whose
// arguments live in HBM (one BatchArgs per pair, read through scalar loads).  One launch evaluates
// the current LM pose of EVERY pair of the batch: P times fewer launches, launch boundaries and L2
// invalidations than P pairs solving side by side on their own streams, and P x 15 MB of
// algorithmic traffic behind one ~10 us launch floor.  Per pair the arithmetic, the block
// decomposition and therefore the bits are those of accumulate_kernel.
template <int K, int BS, int PF>
__global__ __launch_bounds__(BS) void accumulate_batch_kernel(const BatchArgs* __restrict__ batch) {
  constexpr int RED_ROWS = 14;
  __shared__ double red[RED_ROWS][BS];
  const BatchArgs& B = batch[blockIdx.y];
  const int nb = B.nb, block = (int)blockIdx.x;
  if (block >= nb) return;
  const AccArgs& a = B.a;
  Pose P;
  if (a.lm) {
    if (a.lm->status != LM_RUNNING) return;
    se3::rotation(a.lm->pose, P.R);
    P.t[0] = a.lm->pose[4]; P.t[1] = a.lm->pose[5]; P.t[2] = a.lm->pose[6];
  } else {
    P = a.pose;
  }
  pose_to_sgprs(P);
  double acc[28];
  accumulate_groups<K, BS, PF>(a, P, block, nb, acc);
  // the same transpose reduction as accumulate_kernel, RED_ROWS rows at a time (same additions in
  // the same order, so the same bits): 28 rows at once are 56 KB of LDS, i.e. two workgroups per CU
  constexpr int NW = BS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SICP_GLOBAL double* partials = (SICP_GLOBAL double*)a.partials;
#pragma unroll
  for (int p0 = 0; p0 < 28; p0 += RED_ROWS) {
    if (p0) __syncthreads();
#pragma unroll
    for (int k = 0; k < RED_ROWS; ++k) red[k][threadIdx.x] = acc[p0 + k];
    __syncthreads();
    for (int kk = wave; kk < RED_ROWS; kk += NW) {
      double sum = 0.0;
#pragma unroll
      for (int t = 0; t < NW; ++t) sum += red[kk][lane + 64 * t];
      sum = wave_sum(sum);
      if (lane == 0) partials[(size_t)(p0 + kk) * nb + block] = sum;
    }
  }
}

// fixed-order sum of the block partials (layout [28][n_blocks]) by one wave: every lane owns rows
// lane, lane+64, ... ; the 28 loads of one trip are independent and coalesced
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int n_blocks, int lane, double (&o)[28]) {
  double s[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) s[k] = 0.0;
  for (int b = lane; b < n_blocks; b += 64) {
    // issue all 28 loads of the trip before the first add: left to itself hipcc recycles one
    // address register and keeps only ~3 loads in flight, which serialises ~200 L2 round trips
    double v[28];
    const double* __restrict__ p = partials + b;
#pragma unroll
    for (int k = 0; k < 28; ++k) v[k] = __builtin_nontemporal_load(p + (size_t)k * n_blocks);
    // one empty asm that "uses" all 28 values: every load has to be issued (and waited for once)
    // before the adds start
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                      "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]));
    asm volatile("" : "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]), "+v"(v[20]), "+v"(v[21]),
                      "+v"(v[22]), "+v"(v[23]), "+v"(v[24]), "+v"(v[25]), "+v"(v[26]), "+v"(v[27]));
#pragma unroll
    for (int k = 0; k < 28; ++k) s[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 28; ++k) o[k] = wave_sum(s[k]);
}

// host-loop solve: sum the per-block partials in a fixed order
__global__ __launch_bounds__(64) void finalize_kernel(const double* partials, int n_blocks, double* out28) {
  double o[28];
  reduce_partials(partials, n_blocks, threadIdx.x, o);
  if (threadIdx.x < 28) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 28; ++k) v = (int)threadIdx.x == k ? o[k] : v;
    out28[threadIdx.x] = v;
  }
}

// device-resident solve: reduce the block partials and advance the LM machine by one evaluation
// (lm.hpp: the same lm_feed the host loop runs).  One wave: 512 VGPRs are available to it, so the
// whole 6x6 trust-region step stays in registers; lane 0 does the serial part.
__global__ __launch_bounds__(64) void lm_step_kernel(LmState* lm, const double* partials, int n_blocks) {
  if (lm->status != LM_RUNNING) return;
  const int lane = threadIdx.x;
  double o[28];
  reduce_partials(partials, n_blocks, lane, o);
  if (lane == 0) {
    LmCore st = *lm;  // the options stay in memory: uniform, read with scalar loads
    lm_feed(st, lm->opt, o);
    *static_cast<LmCore*>(lm) = st;
  }
}

// batch forms: one block (one wave) per pair
__global__ __launch_bounds__(64) void lm_step_batch_kernel(const BatchArgs* __restrict__ batch) {
  const BatchArgs& B = batch[blockIdx.x];
  LmState* lm = B.a.lm_step;
  if (lm->status != LM_RUNNING) return;
  const int lane = threadIdx.x;
  double o[28];
  reduce_partials(B.a.partials, B.nb, lane, o);
  if (lane == 0) {
    LmCore st = *lm;
    lm_feed(st, lm->opt, o);
    *static_cast<LmCore*>(lm) = st;
  }
}

__global__ __launch_bounds__(64) void finalize_batch_kernel(const BatchArgs* __restrict__ batch, double* out28) {
  const BatchArgs& B = batch[blockIdx.x];
  double o[28];
  reduce_partials(B.a.partials, B.nb, threadIdx.x, o);
  if (threadIdx.x < 28) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 28; ++k) v = (int)threadIdx.x == k ? o[k] : v;
    out28[28 * blockIdx.x + threadIdx.x] = v;
  }
}

// final_cloud = float(matrix) * source, the float overload of pcl::transformPointCloud
// (em_icp.hpp:192-198): float matrix, float arithmetic, row by row
__global__ void transform_float_kernel(int n, const float* x, const float* y, const float* z, Mat4f M,
                                       float* ox, float* oy, float* oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = x[i], py = y[i], pz = z[i];
  const float* m = M.m;
  ox[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], px), __fmul_rn(m[1], py)), __fmul_rn(m[2], pz)), m[3]);
  oy[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[4], px), __fmul_rn(m[5], py)), __fmul_rn(m[6], pz)), m[7]);
  oz[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[8], px), __fmul_rn(m[9], py)), __fmul_rn(m[10], pz)), m[11]);
}

// fused label = arg max_s sum_c prob_c * (t_c . CM[:,s]) (s_i . CM[:,s])   (em_icp.hpp:224-266)
__global__ __launch_bounds__(256) void fused_label_kernel(WeightArgs a, uint32_t* out_labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_s) return;
  // the geometric factor of each of the K correspondences does not depend on s
  double gprob[4];
  int jj[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = c < a.K ? a.idx[(size_t)i * a.K + c] : -1;
    jj[c] = j;
    gprob[c] = 0.0;
    if (j >= 0) {
      Corr cr;
      corr_eval<false>(a.pose, a.one_m_eps, a.sx[i], a.sy[i], a.sz[i], a.snx[i], a.sny[i], a.snz[i],
                       a.tx[j], a.ty[j], a.tz[j], a.tnx[j], a.tny[j], a.tnz[j], cr);
      const double two_pi = 6.283185307179586;
      const double probability = pow(two_pi * two_pi * two_pi * cr.detA, -0.5) * exp(-0.5 * cr.r);
      gprob[c] = a.bool_probability ? ((probability != 0.0) ? 1.0 : 0.0) : probability;
    }
  }
  const double* __restrict__ ps = a.s_proj + (size_t)i * a.C;
  double max_prob = 0.0;
  int max_s = 0;
  for (int s = 0; s < a.C; ++s) {
    double sprob = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (jj[c] < 0) continue;
      double temp = a.t_proj[(size_t)jj[c] * a.C + s];
      temp *= ps[s];
      sprob += temp * gprob[c];  // em_icp.hpp:249-253
    }
    if (sprob > max_prob) { max_s = s; max_prob = sprob; }  // first max wins (em_icp.hpp:259)
  }
  out_labels[i] = (uint32_t)(max_s + 1);
}

// statistics: number of live correspondence slots (integer atomics: order independent)
__global__ __launch_bounds__(256) void count_active_kernel(const int* idx, int n, unsigned long long* out) {
  __shared__ unsigned cnt[4];
  unsigned c = 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) c += idx[e] >= 0 ? 1u : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static constexpr int NN_BS = 256;
static constexpr int NN_TILE = 1024;

int nn_queries_per_thread(int K) { return K == 1 ? 4 : (K <= 4 ? 2 : 1); }

template <int K, int Q>
static hipError_t launch_nn_partial(const NNArgs& a, int n_chunks, hipStream_t st) {
  dim3 grid((a.q_count + NN_BS * Q - 1) / (NN_BS * Q), n_chunks);
  hipLaunchKernelGGL((nn_partial_kernel<K, Q, NN_BS, NN_TILE>), grid, dim3(NN_BS), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_nn_partial(int K, const NNArgs& a, int n_chunks, hipStream_t st) {
  switch (K) {
    case 1: return launch_nn_partial<1, 4>(a, n_chunks, st);
    case 4: return launch_nn_partial<4, 2>(a, n_chunks, st);
    case 20: return launch_nn_partial<20, 1>(a, n_chunks, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_nn_merge(int K, const MergeArgs& m, hipStream_t st) {
  dim3 grid((m.q_count + 255) / 256);
  switch (K) {
    case 1: hipLaunchKernelGGL((nn_merge_kernel<1>), grid, dim3(256), 0, st, m); break;
    case 4: hipLaunchKernelGGL((nn_merge_kernel<4>), grid, dim3(256), 0, st, m); break;
    case 20: hipLaunchKernelGGL((nn_merge_kernel<20>), grid, dim3(256), 0, st, m); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_bvh_knn_quad(int K, const KnnArgs& a, hipStream_t st) {
  if (a.q_count <= 0) return hipSuccess;
  dim3 grid((a.q_count + 15) / 16);
  switch (K) {
    case 1: hipLaunchKernelGGL((bvh_knn_quad_kernel<1>), grid, dim3(64), 0, st, a); break;
    case 4: hipLaunchKernelGGL((bvh_knn_quad_kernel<4>), grid, dim3(64), 0, st, a); break;
    case 20: hipLaunchKernelGGL((bvh_knn_quad_kernel<20>), grid, dim3(64), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_bvh_knn_packet(int K, const KnnArgs& a, hipStream_t st) {
  if (a.q_count <= 0) return hipSuccess;
  const int packets = (a.q_count + 15) / 16;
  static const int wpb_small = [] { const char* e = getenv("SICP_KNN_WPB"); return e ? atoi(e) : 4; }();  // tuning aid
  static const int wpb_big = [] { const char* e = getenv("SICP_KNN_WPB20"); return e ? atoi(e) : 2; }();
  auto grid = [&](int wpb) { return dim3((packets + wpb - 1) / wpb); };
#define SICP_PK(KK, W) hipLaunchKernelGGL((bvh_knn_packet_kernel<KK, W>), grid(W), dim3(64 * W), 0, st, a)
  switch (K) {
    case 1: if (wpb_small == 1) SICP_PK(1, 1); else if (wpb_small == 2) SICP_PK(1, 2); else SICP_PK(1, 4); break;
    case 4: if (wpb_small == 1) SICP_PK(4, 1); else if (wpb_small == 2) SICP_PK(4, 2); else SICP_PK(4, 4); break;
    case 20: if (wpb_big == 1) SICP_PK(20, 1); else if (wpb_big == 2) SICP_PK(20, 2); else SICP_PK(20, 4); break;
    default: return hipErrorInvalidValue;
  }
#undef SICP_PK
  return hipGetLastError();
}

hipError_t launch_bvh_knn(int K, const KnnArgs& a, hipStream_t st) {
  if (a.q_count <= 0) return hipSuccess;
  dim3 grid((a.q_count + 63) / 64);
  switch (K) {
    case 1: hipLaunchKernelGGL((bvh_knn_kernel<1>), grid, dim3(64), 0, st, a); break;
    case 4: hipLaunchKernelGGL((bvh_knn_kernel<4>), grid, dim3(64), 0, st, a); break;
    case 20: hipLaunchKernelGGL((bvh_knn_kernel<20>), grid, dim3(64), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool nn_k_supported(int K) { return K == 1 || K == 4 || K == 20; }

hipError_t launch_cov(const CovArgs& a, hipStream_t st) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cov_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_proj(const ProjArgs& a, hipStream_t st) {
  const int total = a.n * a.C;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(proj_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_em_weight(const WeightArgs& a, hipStream_t st) {
  const int total = a.n_s * a.K;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(em_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

// job-array launches (lock-step batch): every job of one launch, grid.y = job
hipError_t launch_bvh_knn_packet_jobs(int K, const KnnArgs* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxKnnJobs) {
    const int cnt = n - b < kMaxKnnJobs ? n - b : kMaxKnnJobs;
    KnnJobs J;
    int max_q = 0;
    for (int i = 0; i < cnt; ++i) { J.job[i] = jobs[b + i]; max_q = jobs[b + i].q_count > max_q ? jobs[b + i].q_count : max_q; }
    if (max_q <= 0) continue;
    const int packets = (max_q + 15) / 16;
#define SICP_PKJ(KK, W) hipLaunchKernelGGL((bvh_knn_packet_jobs_kernel<KK, W>), dim3((packets + W - 1) / W, cnt), dim3(64 * W), 0, st, J)
    switch (K) {
      case 1: SICP_PKJ(1, 4); break;
      case 4: SICP_PKJ(4, 4); break;
      case 20: SICP_PKJ(20, 2); break;
      default: return hipErrorInvalidValue;
    }
#undef SICP_PKJ
  }
  return hipGetLastError();
}

hipError_t launch_cov_jobs(const CovArgs* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxSmallJobs) {
    const int cnt = n - b < kMaxSmallJobs ? n - b : kMaxSmallJobs;
    CovJobs J;
    int mx = 0;
    for (int i = 0; i < cnt; ++i) { J.job[i] = jobs[b + i]; mx = jobs[b + i].n > mx ? jobs[b + i].n : mx; }
    if (mx <= 0) continue;
    hipLaunchKernelGGL(cov_jobs_kernel, dim3((mx + 255) / 256, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_proj_jobs(const ProjArgs* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxSmallJobs) {
    const int cnt = n - b < kMaxSmallJobs ? n - b : kMaxSmallJobs;
    ProjJobs J;
    int mx = 0;
    for (int i = 0; i < cnt; ++i) { J.job[i] = jobs[b + i]; const int t = jobs[b + i].n * jobs[b + i].C; mx = t > mx ? t : mx; }
    if (mx <= 0) continue;
    hipLaunchKernelGGL(proj_jobs_kernel, dim3((mx + 255) / 256, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_em_weight_jobs(const WeightArgs* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxKnnJobs) {
    const int cnt = n - b < kMaxKnnJobs ? n - b : kMaxKnnJobs;
    WeightJobs J;
    int mx = 0;
    for (int i = 0; i < cnt; ++i) { J.job[i] = jobs[b + i]; const int t = jobs[b + i].n_s * jobs[b + i].K; mx = t > mx ? t : mx; }
    if (mx <= 0) continue;
    hipLaunchKernelGGL(em_weight_jobs_kernel, dim3((mx + 255) / 256, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_fused_labels(const WeightArgs& a, uint32_t* out, hipStream_t st) {
  if (a.n_s <= 0) return hipSuccess;
  hipLaunchKernelGGL(fused_label_kernel, dim3((a.n_s + 255) / 256), dim3(256), 0, st, a, out);
  return hipGetLastError();
}

int accumulate_blocks(int total) {
  // each lane sums several slots in registers before the (LDS-bound) wave reduction; the grid
  // still covers every CU.  SICP_ACC_SLOTS_PER_LANE is a tuning aid.
  static const int per_lane = [] { const char* e = getenv("SICP_ACC_SLOTS_PER_LANE"); return e ? atoi(e) : 8; }();
  const int bs = 256;
  int nb = (total + bs * per_lane - 1) / (bs * per_lane);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb;
}

static hipError_t launch_accumulate_only(const AccArgs& a, int nb, hipStream_t st) {
  switch (a.K) {
    case 1: hipLaunchKernelGGL((accumulate_kernel<1, 256, false>), dim3(nb), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((accumulate_kernel<4, 256, false>), dim3(nb), dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL((accumulate_kernel<20, 256, false>), dim3(nb), dim3(256), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// batched evaluation: `batch` holds n BatchArgs in HBM, max_nb = largest block count among them
hipError_t launch_accumulate_batch(int K, const BatchArgs* batch, int n, int max_nb, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const dim3 grid(max_nb, n);
  static const int pf = [] { const char* e = getenv("SICP_ACC_BATCH_PF"); return e ? atoi(e) : 4; }();  // tuning aid
#define SICP_AB(KK) \
  do { \
    if (pf == 8) hipLaunchKernelGGL((accumulate_batch_kernel<KK, 256, 8>), grid, dim3(256), 0, st, batch); \
    else hipLaunchKernelGGL((accumulate_batch_kernel<KK, 256, 4>), grid, dim3(256), 0, st, batch); \
  } while (0)
  switch (K) {
    case 1: SICP_AB(1); break;
    case 4: SICP_AB(4); break;
    case 20: SICP_AB(20); break;
    default: return hipErrorInvalidValue;
  }
#undef SICP_AB
  return hipGetLastError();
}

hipError_t launch_lm_step_batch(const BatchArgs* batch, int n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(lm_step_batch_kernel, dim3(n), dim3(64), 0, st, batch);
  return hipGetLastError();
}

hipError_t launch_finalize_batch(const BatchArgs* batch, int n, double* out28, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_batch_kernel, dim3(n), dim3(64), 0, st, batch, out28);
  return hipGetLastError();
}

// chained solve: a.lm = state in, a.lm_step = state out, a.partials_in / a.partials = previous / this
// evaluation's partials (same grid for every launch of a solve)
hipError_t launch_accumulate_chain(const AccArgs& a, hipStream_t st) {
  const int nb = accumulate_blocks(a.n_s * a.K);
  switch (a.K) {
    case 1: hipLaunchKernelGGL((accumulate_chain_kernel<1, 256>), dim3(nb), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((accumulate_chain_kernel<4, 256>), dim3(nb), dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL((accumulate_chain_kernel<20, 256>), dim3(nb), dim3(256), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// one LM evaluation as one kernel (a.lm, a.lm_step, a.ticket set)
hipError_t launch_accumulate_fused(const AccArgs& a, hipStream_t st) {
  const int nb = accumulate_blocks(a.n_s * a.K);
  switch (a.K) {
    case 1: hipLaunchKernelGGL((accumulate_kernel<1, 256, true>), dim3(nb), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((accumulate_kernel<4, 256, true>), dim3(nb), dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL((accumulate_kernel<20, 256, true>), dim3(nb), dim3(256), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_accumulate_kernel(const AccArgs& a, hipStream_t st) {
  return launch_accumulate_only(a, accumulate_blocks(a.n_s * a.K), st);
}

hipError_t launch_finalize(const AccArgs& a, double* out28, hipStream_t st) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, st, a.partials, accumulate_blocks(a.n_s * a.K), out28);
  return hipGetLastError();
}

hipError_t launch_accumulate_lm(const AccArgs& a, LmState* lm, hipStream_t st) {
  const int nb = accumulate_blocks(a.n_s * a.K);
  hipError_t e = launch_accumulate_only(a, nb, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(64), 0, st, lm, a.partials, nb);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void count_active_jobs_kernel(CountJobs jobs) {
  const CountJob& J = jobs.job[blockIdx.y];
  const int* idx = J.idx;
  unsigned long long cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < J.n; i += gridDim.x * blockDim.x) cnt += idx[i] >= 0;
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(J.out, cnt);
}

hipError_t launch_count_active_jobs(const CountJob* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxSmallJobs) {
    const int cnt = n - b < kMaxSmallJobs ? n - b : kMaxSmallJobs;
    CountJobs J;
    int mx = 0;
    for (int i = 0; i < cnt; ++i) { J.job[i] = jobs[b + i]; mx = jobs[b + i].n > mx ? jobs[b + i].n : mx; }
    if (mx <= 0) continue;
    const int gx = (mx + 255) / 256 < 256 ? (mx + 255) / 256 : 256;
    hipLaunchKernelGGL(count_active_jobs_kernel, dim3(gx, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_count_active(const int* idx, int n, unsigned long long* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(count_active_kernel, dim3(min(256, (n + 255) / 256)), dim3(256), 0, st, idx, n, out);
  return hipGetLastError();
}

hipError_t launch_transform_float(int n, const float* x, const float* y, const float* z, const Mat4f& M,
                                  float* ox, float* oy, float* oz, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(transform_float_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, x, y, z, M, ox, oy, oz);
  return hipGetLastError();
}

}  // namespace sicp
