// knn_kernels.hip -- exact k-nearest-neighbour search on gfx950 (CDNA4, wave64).
//
// Replaces pcl::transformPointCloud + KdTreeFLANN::nearestKSearch + the dist^2 < 250 gate of the
// reference (em_icp.hpp:46-65, gicp.hpp:54-70, semantic_icp.hpp:53-69) and the k = 20
// self-search of ComputeCovariances (em_icp.hpp:283-296).  Three engines with bit-identical results:
// an LDS-tiled brute force, a box tree walked per query, and the same tree walked by packets of 16
// queries (default).
//
// Design notes (MI355X): clouds live in HBM in Hilbert-curve order (SoA float32 + a packed float4
// x,y,z,caller-index copy for the search kernels).  Top-K lists are 64-bit (distance, caller index)
// keys in statically indexed VGPRs.  No floating-point atomics anywhere, so every result is
// run-to-run reproducible.  Nothing here is GEMM shaped: no MFMA.  All three kernel files are
// compiled with -ffp-contract=off; fused multiply-adds are re-enabled per function where the float64
// algebra only needs tolerance-level parity.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define SICP_HD __host__ __device__
#include "kernels.h"
#include "device_geometry.hpp"

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

// The same distance for the lane's four candidates of a leaf, with the x and y differences and squares formed by the
// packed float32 instructions of gfx950 (v_pk_add_f32 / v_pk_mul_f32 operate on a register pair: every half is an
// ordinary IEEE float32 operation, so the bits are l2_simple's): 6 instead of 8 instructions per point.
typedef float knn_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float l2_simple_pk(knn_v2f qxy, float qz, float bx, float by, float bz) {
#pragma clang fp contract(off)
  const knn_v2f b = {bx, by};
  const knn_v2f d = qxy - b;
  const knn_v2f s = d * d;
  const float dz = __fsub_rn(qz, bz);
  float r = __fadd_rn(s.x, s.y);
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
// any list length (the padded lists of uncommon k): one two-instruction asm statement per element
template <int K>
__device__ __forceinline__ void key_insert(u64 (&bk)[K], u64 key) {
  double c = __longlong_as_double((long long)key);
  double* b = reinterpret_cast<double*>(bk);
#pragma unroll
  for (int j = K - 1; j >= 1; --j) asm("v_max_f64 %1, %2, %0\n\tv_min_f64 %0, %2, %0" : "+v"(c), "=&v"(b[j]) : "v"(b[j - 1]));
  b[0] = c;
}
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

// ---- the four lanes of a quad merge their ascending lists in registers ---------------------------------
// Two rounds (partner = lane ^ 1, then lane ^ 2), after which every lane of the quad holds the K smallest keys of
// all four lists, ascending.  A round is the bitonic merge of two sorted lists: min(a[i], b[N-1-i]) are the N
// smallest of both as a bitonic sequence, which log2(N) stages of compare-exchanges sort (N = K for 1 and 4, 32
// for the longer lists, padded with the empty key).  The lists are disjoint up to identical padding keys, so the
// result is the one the serial four-way merge through LDS gave (one lane per query, 58 instructions per output
// entry: 1160 of the K = 20 wave's 6118) -- in ~420 instructions for K = 20, ~50 for K = 4, all lanes working.
template <int CTRL>
__device__ __forceinline__ u64 quad_perm_key(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 key_min(u64 a, u64 b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (u64)__double_as_longlong(r);
}
template <int K, int CTRL>
__device__ __forceinline__ void quad_merge_round(u64 (&bk)[K]) {
  if constexpr (K == 1) {
    bk[0] = key_min(bk[0], quad_perm_key<CTRL>(bk[0]));
  } else if constexpr (K == 4) {
    u64 pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = quad_perm_key<CTRL>(bk[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) bk[i] = key_min(bk[i], pb[3 - i]);
    key_cswap(bk[0], bk[2]); key_cswap(bk[1], bk[3]); key_cswap(bk[0], bk[1]); key_cswap(bk[2], bk[3]);
  } else {
    static_assert(K > 4 && K <= 32, "list lengths 20 and 32 (and the 8 / 12 / 16 of timing experiments)");
    constexpr int N = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
    u64 l[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int j = N - 1 - i;  // the partner's entry
      if (i < K && j < K) l[i] = key_min(bk[i], quad_perm_key<CTRL>(bk[j]));
      else if (i < K) l[i] = bk[i];
      else if (j < K) l[i] = quad_perm_key<CTRL>(bk[j]);
      else l[i] = KEY_EMPTY;
    }
    // the log2(N) stages that sort the bitonic sequence, pruned to what the first K outputs depend on: a compare-
    // exchange whose upper output is never read again keeps only its minimum (one instruction instead of two), one
    // of which neither output is read disappears.  need[s][i] = "entry i is read after stage s" is known at compile
    // time: after the last stage the entries below K, before that whatever feeds a needed entry.
    // (K = 20: 116 instead of 160 instructions per round.)
#pragma unroll
    for (int j = N / 2; j >= 1; j >>= 1)
#pragma unroll
      for (int i = 0; i < N; ++i)
        if ((i & j) == 0) {
          // after this stage the block [i & ~(2j-1), +2j) is split into halves of j that never mix again, and
          // within the sorted result the block's entries keep their block: entry e ends up inside its final
          // j-block, so it is needed iff that block starts below K
          const bool lo_needed = (i & ~(j - 1)) < K;
          const bool hi_needed = ((i + j) & ~(j - 1)) < K;
          if (lo_needed && hi_needed) key_cswap(l[i], l[i + j]);
          else if (lo_needed) l[i] = key_min(l[i], l[i + j]);
        }
#pragma unroll
    for (int i = 0; i < K; ++i) bk[i] = l[i];
  }
}
template <int K>
__device__ __forceinline__ void quad_merge(u64 (&bk)[K]) {
  quad_merge_round<K, 0xB1>(bk);  // quad_perm [1,0,3,2]
  quad_merge_round<K, 0x4E>(bk);  // quad_perm [2,3,0,1]
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
// k_out <= K entries are written (the k nearest are the first k of a longer exact list)
template <int K>
__device__ __forceinline__ void emit(const u64 (&bk)[K], const int* inv, float gate_sq, int* out_i, float* out_d, size_t o, int k_out) {
  // fully unrolled (the list lives in registers: no dynamic index), entries beyond k_out predicated off
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const unsigned orig = (unsigned)bk[k];
    const float d = bk[k] == KEY_EMPTY ? INFINITY : key_dist(bk[k]);
    const bool keep = orig != 0xffffffffu && d < gate_sq;  // strict <, float compare (em_icp.hpp:65)
    if (k < k_out) {
      out_i[o + k] = keep ? inv[orig] : -1;
      if (out_d) out_d[o + k] = d;
    }
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
  emit<K>(bk, a.inv, a.gate_sq, a.out_i, a.out_d, (size_t)(a.q_begin + q) * a.k_out, a.k_out);
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
  emit<K>(bk, a.inv, a.gate_sq, a.out_i, a.out_d, (size_t)(a.q_begin + q) * a.k_out, a.k_out);
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
// Pruning bound of a quad: an upper bound of the query's true K-th smallest distance from the four
// partial lists (each the K best of a quarter of the candidates seen so far).
//   (1) any lane's own K-th distance: its K candidates are among all candidates;
//   (2) max over lanes of the lane's ceil(K/4)-th distance: 4 * ceil(K/4) >= K candidates lie
//       within it.  The quarters are a quasi-random split, so (2) is close to the true K-th.
// Computed on the distances' BIT PATTERNS as unsigned integers: a distance is a non-negative float or +inf, so
// the unsigned order is the float order, and the empty entry's pattern (0x7fefffff, a NaN) lies above +inf --
// "not enough entries yet" is simply the largest value: the minimum in (1) skips it, the maximum in (2) keeps
// it, and a bound that is still that NaN prunes nothing (`!(d > NaN)` holds).  Integer min / max need no
// canonicalising copies and fold into their DPP operand: 5 instructions (round 4: 22, with the +inf selects).
__device__ __forceinline__ unsigned quad_min_u(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  return min(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
}
__device__ __forceinline__ unsigned quad_max_u(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));
  return max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float quad_bound(const u64 (&bk)[K]) {
  constexpr int M = (K + 3) / 4;
  return __uint_as_float(min(quad_min_u((unsigned)(bk[K - 1] >> 32)), quad_max_u((unsigned)(bk[M - 1] >> 32))));
}

// K = 4: branch-free.  The lane's four candidates are sorted among themselves (5 compare-exchanges) and
// merged with its ascending list the bitonic way: min(list[i], cand[3 - i]) are the four smallest of the
// eight, as a bitonic sequence that two more rounds of compare-exchanges sort -- 22 v_min / v_max_f64
// instead of four guarded insertions (a 64-bit compare, an exec-mask branch and a 7-instruction carry
// chain each; some lane of the wave almost always inserts, so the wave almost always ran them all).
// Candidates beyond the quad bound may now enter a lane's list; they are beyond the true K-th
// distance, so the merged result -- the K smallest keys of everything seen -- is the same.
__device__ __forceinline__ void scan_points_quad4(const float4 (&t)[4], float px, float py, float pz, u64 (&bk)[4], float& wd) {
  u64 c[4];
  const knn_v2f qxy = {px, py};
#pragma unroll
  for (int p = 0; p < 4; ++p) c[p] = make_key(l2_simple_pk(qxy, pz, t[p].x, t[p].y, t[p].z), __float_as_uint(t[p].w));
  key_cswap(c[0], c[1]); key_cswap(c[2], c[3]); key_cswap(c[0], c[2]); key_cswap(c[1], c[3]); key_cswap(c[1], c[2]);
  double l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double a = __longlong_as_double((long long)bk[i]), b = __longlong_as_double((long long)c[3 - i]);
    asm("v_min_f64 %0, %1, %2" : "=v"(l[i]) : "v"(a), "v"(b));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) bk[i] = (u64)__double_as_longlong(l[i]);
  key_cswap(bk[0], bk[2]); key_cswap(bk[1], bk[3]); key_cswap(bk[0], bk[1]); key_cswap(bk[2], bk[3]);
  wd = quad_bound<4>(bk);
}

// t = the lane's four points of the leaf: lane s of the quad takes points s, s+4, s+8, s+12 (a mixed quarter)
template <int K>
__device__ __forceinline__ void scan_points_quad(const float4 (&t)[4], float px, float py, float pz, u64 (&bk)[K], float& wd) {
  if constexpr (K == 4) { scan_points_quad4(t, px, py, pz, bk, wd); return; }
  if constexpr (K >= 8) {
    // Long lists: an insertion is 2 K instructions and the WAVE runs it whenever any of its 64 lanes
    // inserts.  The lane's 4 candidates are sorted first (10 instructions); a lane then inserts them in
    // ascending order, and a candidate that fails (beyond the quad bound or the lane's own K-th key) is
    // followed only by larger ones -- so the wave stops at the first round in which no lane inserts,
    // instead of running all four insertions almost every time.
    u64 c[4];
    const knn_v2f qxy = {px, py};
#pragma unroll
    for (int p = 0; p < 4; ++p) c[p] = make_key(l2_simple_pk(qxy, pz, t[p].x, t[p].y, t[p].z), __float_as_uint(t[p].w));
    key_cswap(c[0], c[1]); key_cswap(c[2], c[3]); key_cswap(c[0], c[2]); key_cswap(c[1], c[3]); key_cswap(c[1], c[2]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool pass = !(key_dist(c[r]) > wd) && c[r] < bk[K - 1];
      if (__ballot(pass) == 0) break;
      if (pass) key_insert<K>(bk, c[r]);
    }
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float d = l2_simple(px, py, pz, t[p].x, t[p].y, t[p].z);
      if (!(d > wd)) {  // beyond the quad bound it cannot be among the K nearest
        const u64 key = make_key(d, __float_as_uint(t[p].w));
        if (key < bk[K - 1]) key_insert<K>(bk, key);
      }
    }
  }
  wd = quad_bound<K>(bk);
}

template <int K>
__device__ __forceinline__ void scan_leaf_quad(const float4* __restrict__ pts, float px, float py, float pz, u64 (&bk)[K], float& wd) {
  float4 t[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) t[p] = pts[4 * p];
  scan_points_quad<K>(t, px, py, pz, bk, wd);
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
  const float4* __restrict__ pts_all = a.tree.pts4 + a.tree.pt_begin;
  const float4* __restrict__ pts = pts_all + sub;
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
    const size_t o = (size_t)(a.q_begin + q) * a.k_out;
    for (int k = 0; k < a.k_out; ++k) {
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

// (the tree is complete -- bvh.hpp: make_levels -- so every child exists; the ones without a real point
// below them have inverted boxes, i.e. lb = +inf)
__device__ __forceinline__ unsigned child_mask_packet(const float4* __restrict__ blo, const float4* __restrict__ bhi, int child_off,
                                                      int parent, int sub, float px, float py, float pz, float wd, float& lb) {
  const int node = child_off + parent * kFan + sub;
  lb = box_lb(blo[node], bhi[node], px, py, pz);
  const bool ok = !(lb > wd);  // lb == wd may hide an equal distance with a lower index
  u64 b = __ballot(ok);                         // bit 4*q + c
  b |= b >> 32; b |= b >> 16; b |= b >> 8; b |= b >> 4;
  return (unsigned)b & 15u;
}

// The leaf a search starts at: the last leaf whose first curve index is <= qc (leaf 0 when there is none).
// 16-ary search of the sorted first indices of the 4^top leaves of the complete tree (the padding leaves
// carry the largest index): lanes 1..15 probe one separator each, a ballot counts the ones at or below qc --
// two tree levels per dependent load instead of the one bit of a binary search (13 loads at 100K points).
__device__ __forceinline__ int locate_leaf(const u64* __restrict__ codes, int top, u64 qc, int lane) {
  int lo = 0, shift = 2 * top;  // the answer lies in [lo, lo + 2^shift)
  const int i = lane & 15;
  while (shift > 0) {
    const int s = shift >= 4 ? shift - 4 : 0;  // this round's step is 2^s leaves: 16 parts (4 in a last odd round)
    const bool probe = i >= 1 && i < (1 << (shift - s));
    const u64 c = probe ? codes[lo + (i << s)] : ~0ull;
    const u64 below = __ballot(probe && c <= qc) & 0xfffeull;  // sorted: the set bits are a prefix of lanes 1..15
    lo += __popcll(below) << s;
    shift = s;
  }
  return lo;
}

// (Round 5, built and measured, not kept: a walk that PREFETCHES below a level-2 node -- the leaf boxes and leaf groups of all
// level-1 children that pass, requested in one round trip, their bounds parked in LDS -- to shorten the chain of dependent
// loads of a search that runs alone.  Same decisions, bit-identical results, and slower everywhere: one search alone 66 -> 89 us
// (first search 70 -> 97), 24 -> 39 us per search in a 16-job launch: 102 VGPRs / 31 KB of LDS per workgroup halve the waves in
// flight, and the loads for children that are pruned before their turn cost more than the round trips saved.
// profiles/r05/prefetching_walk.json.)
constexpr int kWalkLevels = kMaxLevels - 1;  // levels that carry sibling bounds (every level but the root)

// The four leaves below a level-1 node are 64 consecutive points = 1 KB: ONE LDS-DMA instruction (global_load_lds,
// 16 bytes per lane, no destination registers) brings them into a wave-private buffer together with the load of
// their four boxes, so the leaves that pass the box test are scanned from LDS without another round trip to memory.
#ifndef SICP_KNN_DMA_MAXK
#define SICP_KNN_DMA_MAXK 4  // list lengths whose walk buffers leaf groups in LDS (tuning aid; longer lists measured slower with it)
#endif
#define KNN_LDS __attribute__((address_space(3)))
#define KNN_GLOBAL __attribute__((address_space(1)))
typedef float knn_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_leaf_group(const float4* __restrict__ pts_all, unsigned group, KNN_LDS knn_v4f* buf, int lane) {
  const float4* p = pts_all + (size_t)group * (kFan * kLeaf) + lane;
  __builtin_amdgcn_global_load_lds((const KNN_GLOBAL void*)p, (KNN_LDS void*)buf, 16, 0, 0);
}
// the calling lane's four points (sub, sub + 4, sub + 8, sub + 12) of leaf c of the buffered group
__device__ __forceinline__ void lds_leaf_points(const KNN_LDS knn_v4f* buf, unsigned c, int sub, float4 (&t)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const knn_v4f v = buf[c * kLeaf + sub + 4 * p];
    t[p] = make_float4(v.x, v.y, v.z, v.w);
  }
}

// WPB waves (packets) per workgroup: one-wave workgroups are launched too slowly to fill the chip
// (6250 of them at 100K queries: ~1.5 waves per SIMD resident on average)
template <int K, int WPB>
__device__ __forceinline__ void knn_packet_body(const KnnArgs& a, int wg, int n_wg) {
  // One wave-private LDS region: s_lb[L][lane] = lane (q, c)'s lower bound of query q to child c of the node of
  // level L + 1 the walk is below -- the bounds of the siblings that wait on every level of the current path.
  // (The quad's four lists are merged in registers at the end: quad_merge.)
  constexpr size_t kWaveBytes = sizeof(float) * kWalkLevels * 64;
  __shared__ __attribute__((aligned(16))) unsigned char s_wave_all[WPB][kWaveBytes];
  constexpr bool kDma = K <= SICP_KNN_DMA_MAXK;
  __shared__ __attribute__((aligned(16))) float4 s_pts_all[kDma ? WPB : 1][kDma ? kFan * kLeaf : 1];  // the leaf group the walk is in (dma_leaf_group)
  KNN_LDS knn_v4f* const s_pts = (KNN_LDS knn_v4f*)s_pts_all[kDma ? (threadIdx.x >> 6) : 0];
  float (&s_lb)[kWalkLevels][64] = *reinterpret_cast<float (*)[kWalkLevels][64]>(s_wave_all[threadIdx.x >> 6]);
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
  const float4* __restrict__ pts_all = a.tree.pts4 + a.tree.pt_begin;
  const float4* __restrict__ pts = pts_all + sub;
  const float4* __restrict__ blo = a.tree.box_lo + a.tree.node_begin;
  const float4* __restrict__ bhi = a.tree.box_hi + a.tree.node_begin;
  const int top = a.tree.lv.n_levels - 1;

  // --- One shared depth-first walk; all of its state is wave-uniform (scalar registers): the level, the first
  // sibling's index and 4 sibling bits per level.  A launch lasts as long as its longest chain of DEPENDENT
  // loads (PMC, profiles/r03: 4.4 of 8 waves per SIMD resident, a third of their time waiting for memory), so
  // the walk is built to have few of them:
  //   (1) the seed leaf -- the packet's own leaf for the covariance self-search, else the leaf at the middle
  //       query's curve position, found by a 16-ary search (locate_leaf: 4 loads at 100K points, was 13);
  //   (2) the whole root-to-seed path AT ONCE: the node indices on it are arithmetic, so the sibling boxes of
  //       every level are loaded together (one round trip per 4 levels), tested, and their lower bounds left
  //       in LDS -- the walk starts standing on the seed leaf with the complete stack of waiting siblings;
  //   (3) the seed leaf's points give every query a first bound;
  //   (4) from then on a waiting sibling is RE-TESTED with its stored bounds against the queries' current
  //       bounds when its turn comes (no load), and a node is only loaded when some query still needs it.
  // (Round 2: binary search, four seed leaves one after the other, then a walk from the root that entered
  // siblings whose test was out of date: ~41 dependent loads per packet in the median.  A first attempt at
  // (1) in this round -- descend from the root into the child nearest to the middle query -- halved the
  // chain and doubled the visits: Hilbert-ordered boxes overlap, lb = 0 does not say where the neighbours are.)
  int n_box = 0, n_scan = 0;  // statistics (a.dbg), dead code otherwise
  if (top == 0) {
    scan_leaf_quad<K>(pts, px, py, pz, bk, wd);
    ++n_scan;
  } else {
    int seed_leaf;
    if (a.self) {
      seed_leaf = bid;  // the 16 queries ARE leaf `bid`
    } else {
      // the previous search of these queries, when there is one (outer iterations move the pose little): the
      // leaf of the middle query's previous nearest neighbour -- one load instead of the curve transform and
      // the search.  Any leaf is a legal seed; a stale one only costs visits.
      int prev = -1;
      if (a.seed_hint) prev = __builtin_amdgcn_readlane(a.seed_hint[(size_t)(a.q_begin + q) * a.hint_K], 32);
      if (prev >= a.t_begin && prev < a.t_begin + a.tree.n) {
        seed_leaf = (prev - a.t_begin) / kLeaf;
      } else {
        const u64 qc_lane = curve_code_coarse<10>(px, py, pz, a.tree.lo[0], a.tree.lo[1], a.tree.lo[2], a.tree.scale);
        const u64 qc = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(qc_lane >> 32), 32) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)qc_lane, 32);
        seed_leaf = locate_leaf(a.tree.leaf_code + a.tree.code_begin, top, qc, lane);
      }
    }
    seed_leaf = __builtin_amdgcn_readfirstlane(seed_leaf);
    u64 masks = 0;  // 4 sibling bits per level
    float cur_lb = 0.f;  // this lane's bound for child `sub` on the level the walk stands on
    constexpr int PB = 4;  // levels per round trip of the path phase
    if constexpr (kDma) dma_leaf_group(pts_all, (unsigned)seed_leaf >> 2, s_pts, lane);  // the seed's leaf group travels with the path's boxes
    for (int L0 = 0; L0 < top; L0 += PB) {
      float4 lo[PB], hi[PB];
#pragma unroll
      for (int i = 0; i < PB; ++i) {  // (levels past the top are clamped: loaded, not used)
        const int L = min(L0 + i, top - 1);
        const int node = level_offset(top, L) + (((seed_leaf >> (2 * L)) & ~3) + sub);
        lo[i] = blo[node];
        hi[i] = bhi[node];
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int L = L0 + i;
        if (L < top) {
          const float lb = box_lb(lo[i], hi[i], px, py, pz);
          s_lb[L][lane] = lb;
          if (L == 0) cur_lb = lb;
          u64 b = __ballot(lb < INFINITY);  // (a node without a real point below it has lb = +inf)
          b |= b >> 32; b |= b >> 16; b |= b >> 8; b |= b >> 4;
          const unsigned m4 = ((unsigned)b & 15u) & ~(1u << ((seed_leaf >> (2 * L)) & 3));  // the path's own node is not waiting
          masks |= (u64)m4 << (4 * L);
          ++n_box;
        }
      }
    }
    // the seed: one leaf; for the long lists the whole group of four, sorted by a network
    if constexpr (K >= 16) {
      // the 16 seed candidates of a lane all enter its (empty) list.  Sixteen insertions are 16 x 39
      // instructions; writing them into the first 16 slots and sorting those with a 63-comparator network is
      // 126 (same list: the keys are unique up to identical padding keys).
      if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group has landed in LDS
#pragma unroll
      for (int l = 0; l < kFan; ++l) {
        float4 t[4];
        if constexpr (kDma) {
          lds_leaf_points(s_pts, (unsigned)l, sub, t);
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p) t[p] = pts[(size_t)((seed_leaf & ~3) + l) * kLeaf + 4 * p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) bk[4 * l + p] = make_key(l2_simple_pk(knn_v2f{px, py}, pz, t[p].x, t[p].y, t[p].z), __float_as_uint(t[p].w));
      }
      key_sort16<K>(bk);
      wd = quad_bound<K>(bk);
      masks &= ~15ull;
      n_scan += 4;
    } else {
      if constexpr (kDma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the group has landed in LDS
        float4 t[4];
        lds_leaf_points(s_pts, (unsigned)seed_leaf & 3u, sub, t);
        scan_points_quad<K>(t, px, py, pz, bk, wd);
      } else {
        scan_leaf_quad<K>(pts + (size_t)seed_leaf * kLeaf, px, py, pz, bk, wd);
      }
      ++n_scan;
    }
    unsigned sh = 0u, ubase = (unsigned)seed_leaf & ~3u;
    const unsigned t4 = 1u << (2 * (top + 1));  // level_offset(top, l) = (t4 - (t4 >> 2 l)) / 3
    // The bounds only tighten, so a waiting sibling that no query needs any more never has to be visited: whenever the
    // walk ARRIVES on a level (back from a subtree, or after a leaf scan on the leaf level) the level's waiting siblings
    // are re-tested all at once with their stored bounds -- lane (q, c) holds query q's bound to sibling c: one
    // compare, one ballot folded to 4 bits -- and the failed ones leave the mask for good.  Every sibling that is
    // popped afterwards is needed, so the loop below only turns for real visits (round 4 popped and re-tested
    // them one by one: a whole turn of scalar bookkeeping per skipped sibling, ~15 per median packet), and a
    // climb goes straight to the nearest level that still has someone waiting.
    auto drop_unneeded = [&]() {
      u64 b = __ballot(!(cur_lb > wd));  // bit 4 q + c: query q still needs sibling c  (lb == wd may hide a tie with a lower index)
      b |= b >> 32; b |= b >> 16; b |= b >> 8; b |= b >> 4;
      masks &= ~((u64)(~(unsigned)b & 15u) << sh);
    };
    drop_unneeded();
    // Two nested loops: the inner one only moves through the tree (scalar state, box tests) until it stands
    // on a leaf some query needs, the outer one scans that leaf -- so the K-entry lists are carried by
    // exactly one loop with one back edge (with `continue`s in a single loop the compiler keeps several
    // copies of the lists alive and moves them at every edge: 218 instead of 52 registers at K = 20).
    for (;;) {
      int leaf = -1;
      for (;;) {
        const u64 rest = masks >> sh;  // the waiting siblings of this level and of every level above it
        if (rest == 0) break;          // nobody waits anywhere: the walk is over
        const unsigned m = (unsigned)rest & 15u;
        if (m == 0) {  // this level is done: up to the nearest level with a waiting sibling
          const unsigned up = (unsigned)__builtin_ctzll(rest) & ~3u;  // 4 bits per level
          sh += up;
          ubase = (ubase >> (up >> 1)) & ~3u;
          cur_lb = s_lb[sh >> 2][lane];
          drop_unneeded();
          continue;
        }
        const unsigned c = (unsigned)__ffs(m) - 1u;
        masks ^= 1ull << (sh + c);
        const unsigned node = ubase + c;
        if (sh == 0u) { leaf = (int)node; break; }
        ++n_box;
        sh -= 4u;
        ubase = node * kFan;
        if constexpr (kDma)
          if (sh == 0u) dma_leaf_group(pts_all, node, s_pts, lane);  // the children are leaves: their points travel with their boxes
        const unsigned cm = child_mask_packet(blo, bhi, (int)((t4 - (t4 >> (sh >> 1))) / 3u), (int)node, sub, px, py, pz, wd, cur_lb);
        masks |= (u64)cm << sh;  // (a level's sibling bits are zero when the walk descends into it)
        s_lb[sh >> 2][lane] = cur_lb;
      }
      if (leaf < 0) break;
      if constexpr (kDma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the group's DMA was issued ahead of the box loads that put the walk here)
        float4 t[4];
        lds_leaf_points(s_pts, (unsigned)leaf & 3u, sub, t);
        scan_points_quad<K>(t, px, py, pz, bk, wd);
      } else {
        scan_leaf_quad<K>(pts + (size_t)leaf * kLeaf, px, py, pz, bk, wd);
      }
      ++n_scan;
      drop_unneeded();  // (the walk stands on the leaf level: cur_lb is the bound to this leaf's siblings)
    }
  }
  if (a.dbg && sub == 0 && q_raw < a.q_count) { a.dbg[2 * q] = n_box; a.dbg[2 * q + 1] = n_scan; }

  // --- merge the quad's four ascending lists (registers: quad_merge); lane `sub` of the quad emits the entries
  // sub, sub + 4, ... -- [query][K] output: consecutive lanes write consecutive addresses
  quad_merge<K>(bk);
  {
    // [query][K], or [K][out_stride] (consecutive queries -> consecutive addresses)
    const size_t o = a.out_stride > 0 ? (size_t)(a.q_begin + q) : (size_t)(a.q_begin + q) * a.k_out;
    const size_t ks = a.out_stride > 0 ? (size_t)a.out_stride : 1;
    unsigned live = 0;  // neighbours of this wave that passed the gate (statistics)
#pragma unroll
    for (int t = 0; t < (K + 3) / 4; ++t) {
      // entry k = 4 t + sub of the merged list (static register indices: a select per lane of the quad)
      u64 best = bk[4 * t < K ? 4 * t : K - 1];
#pragma unroll
      for (int s2 = 1; s2 < 4; ++s2)
        if (4 * t + s2 < K) best = sub == s2 ? bk[4 * t + s2] : best;
      const int k = 4 * t + sub;
      const bool mine = q_raw < a.q_count && k < K && k < a.k_out;
      const unsigned orig = (unsigned)best;
      const float d = best == KEY_EMPTY ? INFINITY : key_dist(best);
      const bool keep = orig != 0xffffffffu && d < a.gate_sq;  // strict <, float compare (em_icp.hpp:65)
      int nb = -1;  // device index of the neighbour, -1 = none / gated out
      if (mine) {
        if (keep) nb = a.inv[orig];
        a.out_i[o + k * ks] = nb;
        if (a.out_d) a.out_d[o + k * ks] = d;
      }
      if (a.live_cnt) live += (unsigned)__popcll(__ballot(mine && keep));
      if constexpr (K == 4) {
        // The slot's EM weight, here rather than in a kernel of its own (em_weight_rows4_kernel: one more launch per
        // search that re-reads the indices and is bound by its gathers): prob = sum_s (t_dist . CM[:, s]) (s_dist .
        // CM[:, s]) from the two projection rows, in ascending s with every product rounded on its own
        // (em_icp.hpp:84-89), times Probability()'s bool (:108) -- operation for operation what that kernel does.
        // The gathers of this wave wait while the other waves of the SIMD walk.
        if (a.w_out != nullptr) {
          double w = 0.0;
          if (mine && nb >= 0) {
            const int C = a.w_C, PS = proj_stride(C), NP = PS / 2;
            typedef double v2d_t __attribute__((ext_vector_type(2)));
            const v2d_t* __restrict__ psrc = reinterpret_cast<const v2d_t*>(a.w_sproj + (size_t)(a.q_begin + q) * PS);
            const v2d_t* __restrict__ ptgt = reinterpret_cast<const v2d_t*>(a.w_tproj + (size_t)nb * PS);
            const PointRec sr = a.w_srec[a.q_begin + q], tr = a.w_trec[nb];
            double prob = 0.0;
            for (int kk = 0; kk < NP; ++kk) {
              const v2d_t tv = ptgt[kk], sv = psrc[kk];
              { double temp = tv.x; temp *= sv.x; prob += temp; }
              if (2 * kk + 1 < C) { double temp = tv.y; temp *= sv.y; prob += temp; }
            }
            Pose P;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              P.R[3 * r + 0] = a.M[4 * r + 0]; P.R[3 * r + 1] = a.M[4 * r + 1]; P.R[3 * r + 2] = a.M[4 * r + 2];
              P.t[r] = a.M[4 * r + 3];
            }
            Corr cr;
            corr_eval<false>(P, a.w_one_m_eps, sr.x, sr.y, sr.z, sr.nx, sr.ny, sr.nz, tr.x, tr.y, tr.z, tr.nx, tr.ny, tr.nz, cr);
            w = prob * geometric_gate(cr, a.w_bool_probability);
          }
          if (mine) a.w_out[o + k * ks] = w;
        }
      }
    }
    if (a.live_cnt && lane == 0 && live) atomicAdd(a.live_cnt + (bid & (kLiveCounters - 1)), (unsigned long long)live);
  }
}

template <int K, int WPB>
__global__ __launch_bounds__(64 * WPB) void bvh_knn_packet_kernel(KnnArgs a) {
  knn_packet_body<K, WPB>(a, (int)blockIdx.x, (int)gridDim.x);
}

// Several searches in one launch (lock-step batch: all pairs' searches of a phase): blockIdx.y picks
// the job from an array passed BY VALUE -- kernel arguments keep their pointers typed as HBM and
// are read with scalar loads.  The jobs' long tails overlap inside the one launch.
// (SICP_KNN20_WAVES, build-time experiment: waves per SIMD the long-list kernels are compiled for.  Their 122 VGPRs give 4;
// forced to 5 (96 VGPRs, 72 bytes of spills) a k = 20 self-search takes 39.7 instead of 40.4 us in a 16-job launch, forced to 6
// (80 VGPRs, 136 bytes) 41.3 us: not worth the spills.)
#ifndef SICP_KNN20_WAVES
#define SICP_KNN20_WAVES 1
#endif
template <int K, int WPB>
__global__ __launch_bounds__(64 * WPB, (K >= 16 ? SICP_KNN20_WAVES : 1)) void bvh_knn_packet_jobs_kernel(KnnJobs jobs) {
  const KnnArgs& a = jobs.job[blockIdx.y];
  const int n_wg = ((a.q_count + 15) / 16 + WPB - 1) / WPB;
  if ((int)blockIdx.x >= n_wg) return;
  knn_packet_body<K, WPB>(a, (int)blockIdx.x, n_wg);
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
    case 32: return launch_nn_partial<32, 1>(a, n_chunks, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_nn_merge(int K, const MergeArgs& m, hipStream_t st) {
  dim3 grid((m.q_count + 255) / 256);
  switch (K) {
    case 1: hipLaunchKernelGGL((nn_merge_kernel<1>), grid, dim3(256), 0, st, m); break;
    case 4: hipLaunchKernelGGL((nn_merge_kernel<4>), grid, dim3(256), 0, st, m); break;
    case 20: hipLaunchKernelGGL((nn_merge_kernel<20>), grid, dim3(256), 0, st, m); break;
    case 32: hipLaunchKernelGGL((nn_merge_kernel<32>), grid, dim3(256), 0, st, m); break;
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
    case 32: hipLaunchKernelGGL((bvh_knn_quad_kernel<32>), grid, dim3(64), 0, st, a); break;
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
    case 32: SICP_PK(32, 2); break;
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
    case 32: hipLaunchKernelGGL((bvh_knn_kernel<32>), grid, dim3(64), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

bool nn_k_supported(int K) { return K == 1 || K == 4 || K == 20; }  // correspondences per source point (accumulate kernels)
int nn_list_len(int k) { return k < 1 ? 0 : (k == 1 ? 1 : (k <= 4 ? 4 : (k <= 20 ? 20 : (k <= 32 ? 32 : 0)))); }

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
#if defined(SICP_EXPERIMENT_K20_LISTS)  // timing experiment only: the k = 20 self-search with SHORTER per-lane lists -- wrong beyond that many entries
      case 20: SICP_PKJ(SICP_EXPERIMENT_K20_LISTS, 2); break;
#else
      case 20: SICP_PKJ(20, 2); break;
#endif
      case 32: SICP_PKJ(32, 2); break;
      default: return hipErrorInvalidValue;
    }
#undef SICP_PKJ
  }
  return hipGetLastError();
}

}  // namespace sicp
