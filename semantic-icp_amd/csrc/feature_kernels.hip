// feature_kernels.hip -- per-point covariance normals and label histograms, the EM label-posterior
// weights, fused labels and the small utility kernels (gfx950, wave64).
//
//   cov_kernel       : ComputeCovariances body                 em_icp.hpp:298-340
//   proj / em_weight : label posterior * Probability           em_icp.hpp:77-89,108
//   fused_label      : getFusedLabels                          em_icp.hpp:224-266
//   transform_float  : the final_cloud of align()              em_icp.hpp:192-198
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
  // The neighbours in batches of 8: their indices, then their coordinates (and labels), are requested together and
  // only then summed -- in list order, so the sums are the ones a one-at-a-time loop gives.  (That loop was a chain of
  // 2 dependent round trips to memory per neighbour, and the histogram below another 2 with a byte read-modify-write
  // in global memory each: ~80 per point; a 100K-point cloud took 26-31 us in a 16-job launch, most of it waiting.)
  // The label counts (k <= 32 < 256) are kept as bytes of two 64-bit registers for up to 16 classes.
  const bool hist_regs = a.hist != nullptr && a.C <= 16;
  unsigned long long cnt_lo = 0ull, cnt_hi = 0ull;
  constexpr int NB = 8;
  for (int j0 = 0; j0 < a.k; j0 += NB) {
    int g[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) g[t] = j0 + t < a.k ? nn[(size_t)(j0 + t) * js] : -1;
    float px[NB], py[NB], pz[NB];
    uint32_t lb[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int gi = g[t] > 0 ? g[t] : 0;
      px[t] = a.x[gi]; py[t] = a.y[gi]; pz[t] = a.z[gi];
      lb[t] = hist_regs ? a.label[gi] : 0u;
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      if (g[t] < 0) continue;
      const float x = px[t], y = py[t], z = pz[t];
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
      const uint32_t l = lb[t];
      if (hist_regs && l >= 1u && l <= (uint32_t)a.C) {  // em_icp.hpp:301: dist(label-1) += 1/k, as a count
        const unsigned long long one = 1ull << (8u * ((l - 1u) & 7u));
        if (l - 1u < 8u) cnt_lo += one; else cnt_hi += one;
      }
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
  {
    PointRec r;
    r.x = a.x[i]; r.y = a.y[i]; r.z = a.z[i];
    r.nx = nx; r.ny = ny; r.nz = nz;
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0u;
    a.rec[i] = r;
    if (a.rec_dense) {
      typedef double dense_v2d __attribute__((ext_vector_type(2)));
      const size_t n = (size_t)a.rec_dense_n;
      const char* src = reinterpret_cast<const char*>(&r);
      *reinterpret_cast<dense_v2d*>(a.rec_dense + 16 * (size_t)i) = *reinterpret_cast<const dense_v2d*>(src);
      *reinterpret_cast<dense_v2d*>(a.rec_dense + 16 * n + 16 * (size_t)i) = *reinterpret_cast<const dense_v2d*>(src + 16);
      *reinterpret_cast<float*>(a.rec_dense + 32 * n + 4 * (size_t)i) = r.z;
    }
  }
  if (a.hist) {
    // label histogram as neighbour counts (em_icp.hpp:301: dist(label-1) += 1/k)
    const int HS = hist_stride(a.C);
    uint8_t* h = a.hist + (size_t)i * HS;  // this lane owns the row (16-byte aligned)
    if (hist_regs) {  // byte c of the two registers is the count of class c + 1; the bytes past C are zero
      typedef unsigned long long hist_v2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<hist_v2*>(h) = hist_v2{cnt_lo, cnt_hi};
    } else {
      for (int c = 0; c < HS; ++c) h[c] = 0;
      for (int j = 0; j < a.k; ++j) {
        const int g = nn[j * js];
        if (g < 0) continue;
        const uint32_t l = a.label[g];
        if (l >= 1u && l <= (uint32_t)a.C) h[l - 1] = (uint8_t)(h[l - 1] + 1);
      }
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
  const uint8_t* h = a.hist + (size_t)i * hist_stride(a.C);
  double temp = 0.0;
  for (int r = 0; r < a.C; ++r) temp += a.hval[h[r]] * a.cm[r * a.C + s];
  const int PS = proj_stride(a.C);
  a.proj[(size_t)i * PS + s] = temp;
  if (s == 0 && PS != a.C) a.proj[(size_t)i * PS + a.C] = 0.0;  // the padding entry
}

// The same projection, one LANE PER POINT (C <= PROJ_CMAX): the point's C counts are read once and turned
// into their table values, CM and the table sit in LDS (broadcast reads), the C results of a point are
// formed in registers -- every one the same ascending sum of separately rounded products as above, so the
// same bits -- and leave through LDS as one coalesced block.  The lane-per-output form above issues three
// loads per term (33 per output, 36 M per 100K-point cloud) and was bound by that: 18 us per cloud.
constexpr int PROJ_CMAX = 16;
__device__ __forceinline__ void proj_rows_body(const ProjArgs& a) {
  __shared__ double s_cm[PROJ_CMAX * PROJ_CMAX];
  __shared__ double s_hval[64];
  __shared__ double s_out[256 * PROJ_CMAX];
  const int C = a.C, PS = proj_stride(C);
  for (int k = threadIdx.x; k < C * C; k += 256) s_cm[k] = a.cm[k];
  for (int k = threadIdx.x; k <= 32; k += 256) s_hval[k] = a.hval[k];  // counts are at most k_cov <= 32 (the buffer carries slack: DevBuf)
  __syncthreads();
  const int i0 = blockIdx.x * 256, i = i0 + (int)threadIdx.x;
  if (i0 >= a.n) return;
  if (i < a.n) {
    const uint8_t* h = a.hist + (size_t)i * hist_stride(C);
    double hv[PROJ_CMAX];
#pragma unroll
    for (int r = 0; r < PROJ_CMAX; ++r) hv[r] = r < C ? s_hval[h[r]] : 0.0;
    for (int s = 0; s < C; ++s) {
      double temp = 0.0;
#pragma unroll
      for (int r = 0; r < PROJ_CMAX; ++r)
        if (r < C) temp += hv[r] * s_cm[r * C + s];
      s_out[threadIdx.x * PS + s] = temp;
    }
    if (PS != C) s_out[threadIdx.x * PS + C] = 0.0;  // the padding entry
  }
  __syncthreads();
  const int valid = min(256, a.n - i0) * PS;
  double* out = a.proj + (size_t)i0 * PS;
  for (int k = threadIdx.x; k < valid; k += 256) out[k] = s_out[k];
}

// ------------------------------------------------------------------------------------------
// EM weight: label posterior from the confusion matrix x the (bool) geometric gate
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_kernel(ProjArgs a) { proj_body(a); }
__global__ __launch_bounds__(256) void proj_jobs_kernel(ProjJobs jobs) { proj_body(jobs.job[blockIdx.y]); }
__global__ __launch_bounds__(256) void proj_rows_kernel(ProjArgs a) { proj_rows_body(a); }
__global__ __launch_bounds__(256) void proj_rows_jobs_kernel(ProjJobs jobs) { proj_rows_body(jobs.job[blockIdx.y]); }

// one 16-byte piece of a projection row (rows are padded to an even number of doubles: proj_stride)
typedef double v2d_t __attribute__((ext_vector_type(2)));

// One lane per SOURCE point, its K slots one after the other: the source's record and projection row are
// read once (they used to be read by each of the K lanes of a lane-per-slot kernel), rows move as 16-byte
// pieces (half the load instructions of 8-byte ones), the K weights leave as one contiguous run.
// em_icp.hpp:84-89 with the two dot products of each term taken from the per-point projections:
// prob = sum_s (t_dist . CM[:, s]) * (s_dist . CM[:, s]), accumulated in ascending s, every product rounded
// on its own -- the same operations in the same order as before, hence the same bits.
constexpr int WEIGHT_CMAX = 16;  // classes the register-resident row holds; beyond, the generic kernel below
template <int K>
__device__ __forceinline__ void em_weight_rows_body(const WeightArgs& a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_s) return;
  const int C = a.C, PS = proj_stride(C), NP = PS / 2;
  int j[K];
#pragma unroll
  for (int c = 0; c < K; ++c) j[c] = a.idx[(size_t)i * K + c];
  v2d_t ps[WEIGHT_CMAX / 2];
  const v2d_t* __restrict__ psrc = reinterpret_cast<const v2d_t*>(a.s_proj + (size_t)i * PS);
#pragma unroll
  for (int k = 0; k < WEIGHT_CMAX / 2; ++k) ps[k] = k < NP ? psrc[k] : v2d_t{0.0, 0.0};
  const PointRec sr = a.srec[i];
  double w[K];
#pragma unroll
  for (int c = 0; c < K; ++c) {
    w[c] = 0.0;
    if (j[c] >= 0) {
      const v2d_t* __restrict__ pt = reinterpret_cast<const v2d_t*>(a.t_proj + (size_t)j[c] * PS);
      v2d_t t[WEIGHT_CMAX / 2];
#pragma unroll
      for (int k = 0; k < WEIGHT_CMAX / 2; ++k) t[k] = k < NP ? pt[k] : v2d_t{0.0, 0.0};
      const PointRec tr = a.trec[j[c]];
      double prob = 0.0;
#pragma unroll
      for (int k = 0; k < WEIGHT_CMAX / 2; ++k) {
        if (2 * k < C) { double temp = t[k].x; temp *= ps[k].x; prob += temp; }
        if (2 * k + 1 < C) { double temp = t[k].y; temp *= ps[k].y; prob += temp; }
      }
      Corr cr;
      corr_eval<false>(a.pose, a.one_m_eps, sr.x, sr.y, sr.z, sr.nx, sr.ny, sr.nz, tr.x, tr.y, tr.z, tr.nx, tr.ny, tr.nz, cr);
      w[c] = prob * geometric_gate(cr, a.bool_probability);  // em_icp.hpp:108
    }
  }
#pragma unroll
  for (int c = 0; c < K; ++c) a.w[(size_t)i * K + c] = w[c];
}

// The same weights WITHOUT the projection arrays (K = 4, C <= 16).  em_weight_rows_body gathers, per slot, the target's
// projection row -- 8 (C + C % 2) = 96 bytes at C = 11 -- and is bound by exactly that gather traffic (77 MB through the
// L1s per 100K x 4 search).  A point's label distribution is 16 BYTES of neighbour counts; here a lane gathers those
// and forms the projections it needs itself: proj[s] = sum_r hval[count[r]] * CM[r][s], r ascending, every product rounded
// on its own -- the very sums proj_kernel forms (same operations, same order: same bits), with CM^T and the count table in
// LDS (broadcast reads).  The source point's projections are formed once and parked in LDS (they are indexed by the class
// loop's counter), then slot after slot: counts -> table values, class after class the target's projection, times the
// source's, added up in the reference's order (em_icp.hpp:84-89).  ~300 float64 operations per slot instead of a
// 96-byte gather, and the projection kernel + arrays (9.6 MB per cloud) drop out of align().  Measured (round 4): 37 us per
// search alone against 13, the same 256-pair step time -- see stages.cpp: weights_from_histograms; behind SICP_WEIGHTS_FROM_HIST.
constexpr int HW_CMAX = 16;
__device__ __forceinline__ double hist_value(const uint4& row, int r, const double* s_hval) {
  const unsigned word = r < 4 ? row.x : (r < 8 ? row.y : (r < 12 ? row.z : row.w));
  return s_hval[(word >> (8 * (r & 3))) & 0xffu];
}

__device__ __forceinline__ void em_weight_hist4_body(const WeightArgs& a) {
  __shared__ __attribute__((aligned(16))) double s_cmT[HW_CMAX * HW_CMAX];  // CM transposed: [s][r], rows of 16
  __shared__ double s_hval[256];
  __shared__ double s_ps[HW_CMAX * 256];  // the source points' projections: [s][thread]
  const int C = a.C;
  for (int k = threadIdx.x; k < HW_CMAX * HW_CMAX; k += 256) {
    const int s = k / HW_CMAX, r = k - s * HW_CMAX;
    s_cmT[k] = (s < C && r < C) ? a.cm[r * C + s] : 0.0;
  }
  for (int k = threadIdx.x; k < 256; k += 256) s_hval[k] = k <= 32 ? a.hval[k] : 0.0;  // counts are at most k_cov <= 32
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_s) return;
  // ---- the source point: its own projections (what proj_kernel would have stored for it)
  {
    const uint4 row = *reinterpret_cast<const uint4*>(a.s_hist + (size_t)i * 16);
    double hv[HW_CMAX];
#pragma unroll
    for (int r = 0; r < HW_CMAX; ++r) hv[r] = hist_value(row, r, s_hval);
    for (int s = 0; s < C; ++s) {
      const double* cm = s_cmT + s * HW_CMAX;
      double temp = 0.0;
#pragma unroll
      for (int r = 0; r < HW_CMAX; ++r)
        if (r < C) temp += hv[r] * cm[r];
      s_ps[s * 256 + threadIdx.x] = temp;
    }
  }
  const int4 jj = *reinterpret_cast<const int4*>(a.idx + (size_t)i * 4);
  const PointRec sr = a.srec[i];
  double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
  for (int c = 0; c < 4; ++c) {
    const int j = c == 0 ? jj.x : (c == 1 ? jj.y : (c == 2 ? jj.z : jj.w));
    double w = 0.0;
    if (j >= 0) {
      const uint4 row = *reinterpret_cast<const uint4*>(a.t_hist + (size_t)j * 16);
      const PointRec tr = a.trec[j];
      double hv[HW_CMAX];
#pragma unroll
      for (int r = 0; r < HW_CMAX; ++r) hv[r] = hist_value(row, r, s_hval);
      double prob = 0.0;
      for (int s = 0; s < C; ++s) {
        const double* cm = s_cmT + s * HW_CMAX;
        double temp = 0.0;  // the target's projection on class s (proj_kernel's sum)
#pragma unroll
        for (int r = 0; r < HW_CMAX; ++r)
          if (r < C) temp += hv[r] * cm[r];
        temp *= s_ps[s * 256 + threadIdx.x];  // em_icp.hpp:86-88
        prob += temp;
      }
      Corr cr;
      corr_eval<false>(a.pose, a.one_m_eps, sr.x, sr.y, sr.z, sr.nx, sr.ny, sr.nz, tr.x, tr.y, tr.z, tr.nx, tr.ny, tr.nz, cr);
      w = prob * geometric_gate(cr, a.bool_probability);  // em_icp.hpp:108
    }
    w0 = c == 0 ? w : w0; w1 = c == 1 ? w : w1; w2 = c == 2 ? w : w2; w3 = c == 3 ? w : w3;
  }
  typedef double w_v2 __attribute__((ext_vector_type(2)));
  w_v2* out = reinterpret_cast<w_v2*>(a.w + (size_t)i * 4);
  out[0] = w_v2{w0, w1};
  out[1] = w_v2{w2, w3};
}

// any K and C: one lane per slot
__device__ __forceinline__ void em_weight_body(const WeightArgs& a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_s * a.K) return;
  const int i = e / a.K;
  const int j = a.idx[e];
  if (j < 0) { a.w[e] = 0.0; return; }
  const int PS = proj_stride(a.C);
  const double* __restrict__ ps = a.s_proj + (size_t)i * PS;
  const double* __restrict__ pt = a.t_proj + (size_t)j * PS;
  double prob = 0.0;
  for (int s = 0; s < a.C; ++s) {
    double temp = pt[s];
    temp *= ps[s];
    prob += temp;
  }
  Corr c;
  const PointRec sr = a.srec[i], tr = a.trec[j];
  corr_eval<false>(a.pose, a.one_m_eps, sr.x, sr.y, sr.z, sr.nx, sr.ny, sr.nz, tr.x, tr.y, tr.z, tr.nx, tr.ny, tr.nz, c);
  a.w[e] = prob * geometric_gate(c, a.bool_probability);
}

__global__ __launch_bounds__(256) void em_weight_kernel(WeightArgs a) { em_weight_body(a); }
__global__ __launch_bounds__(256) void em_weight_jobs_kernel(WeightJobs jobs) { em_weight_body(jobs.job[blockIdx.y]); }
__global__ __launch_bounds__(256) void em_weight_rows4_kernel(WeightArgs a) { em_weight_rows_body<4>(a); }
__global__ __launch_bounds__(256) void em_weight_rows4_jobs_kernel(WeightJobs jobs) { em_weight_rows_body<4>(jobs.job[blockIdx.y]); }
__global__ __launch_bounds__(256) void em_weight_hist4_kernel(WeightArgs a) { em_weight_hist4_body(a); }
__global__ __launch_bounds__(256) void em_weight_hist4_jobs_kernel(WeightJobs jobs) { em_weight_hist4_body(jobs.job[blockIdx.y]); }

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
  const PointRec sr = a.srec[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = c < a.K ? a.idx[(size_t)i * a.K + c] : -1;
    jj[c] = j;
    gprob[c] = 0.0;
    if (j >= 0) {
      Corr cr;
      const PointRec tr = a.trec[j];
      corr_eval<false>(a.pose, a.one_m_eps, sr.x, sr.y, sr.z, sr.nx, sr.ny, sr.nz, tr.x, tr.y, tr.z, tr.nx, tr.ny, tr.nz, cr);
      gprob[c] = geometric_gate(cr, a.bool_probability);
    }
  }
  const int PS = proj_stride(a.C);
  const double* __restrict__ ps = a.s_proj + (size_t)i * PS;
  double max_prob = 0.0;
  int max_s = 0;
  for (int s = 0; s < a.C; ++s) {
    double sprob = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (jj[c] < 0) continue;
      double temp = a.t_proj[(size_t)jj[c] * PS + s];
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


hipError_t launch_cov(const CovArgs& a, hipStream_t st) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cov_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

// sicp_set_covariances: the records of a cloud whose normals the caller supplied (device order), written like cov_body's
__global__ __launch_bounds__(256) void set_normals_kernel(int n, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                          const double* __restrict__ normal3, PointRec* __restrict__ rec, char* __restrict__ rec_dense,
                                                          int rec_dense_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PointRec r;
  r.x = x[i]; r.y = y[i]; r.z = z[i];
  r.nx = normal3[3 * (size_t)i]; r.ny = normal3[3 * (size_t)i + 1]; r.nz = normal3[3 * (size_t)i + 2];
  r.pad_[0] = r.pad_[1] = r.pad_[2] = 0u;
  rec[i] = r;
  if (rec_dense) {
    typedef double dense_v2d __attribute__((ext_vector_type(2)));
    const size_t m = (size_t)rec_dense_n;
    const char* src = reinterpret_cast<const char*>(&r);
    *reinterpret_cast<dense_v2d*>(rec_dense + 16 * (size_t)i) = *reinterpret_cast<const dense_v2d*>(src);
    *reinterpret_cast<dense_v2d*>(rec_dense + 16 * m + 16 * (size_t)i) = *reinterpret_cast<const dense_v2d*>(src + 16);
    *reinterpret_cast<float*>(rec_dense + 32 * m + 4 * (size_t)i) = r.z;
  }
}

__global__ __launch_bounds__(256) void cov9_caller_order_kernel(int n, const PointRec* __restrict__ rec, const double* __restrict__ cov6,
                                                                const int* __restrict__ perm, double one_m_eps, double* __restrict__ out9) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  double* o = out9 + 9 * (size_t)perm[d];
  if (cov6) {
    const double* q = cov6 + 6 * (size_t)d;
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[1]; o[4] = q[3]; o[5] = q[4]; o[6] = q[2]; o[7] = q[4]; o[8] = q[5];
  } else {
    const double v[3] = {rec[d].nx, rec[d].ny, rec[d].nz};
    // the upper triangle, mirrored: bit-symmetric (rows == columns), the host loop's expression and bits
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = a; b < 3; ++b) {
        const double e = (a == b ? 1.0 : 0.0) - one_m_eps * v[a] * v[b];
        o[3 * a + b] = e;
        o[3 * b + a] = e;
      }
  }
}

hipError_t launch_cov9_caller_order(int n, const PointRec* rec, const double* cov6, const int* perm, double one_m_eps, double* out9, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(cov9_caller_order_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, rec, cov6, perm, one_m_eps, out9);
  return hipGetLastError();
}

hipError_t launch_set_normals(int n, const float* x, const float* y, const float* z, const double* normal3, PointRec* rec, char* rec_dense,
                              int rec_dense_n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(set_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, x, y, z, normal3, rec, rec_dense, rec_dense_n);
  return hipGetLastError();
}

hipError_t launch_proj(const ProjArgs& a, hipStream_t st) {
  const int total = a.n * a.C;
  if (total <= 0) return hipSuccess;
  if (a.C <= PROJ_CMAX) hipLaunchKernelGGL(proj_rows_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(proj_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_em_weight(const WeightArgs& a, hipStream_t st) {
  const int total = a.n_s * a.K;
  if (total <= 0) return hipSuccess;
  if (a.s_hist && a.K == 4 && a.C <= HW_CMAX) hipLaunchKernelGGL(em_weight_hist4_kernel, dim3((a.n_s + 255) / 256), dim3(256), 0, st, a);
  else if (a.K == 4 && a.C <= WEIGHT_CMAX) hipLaunchKernelGGL(em_weight_rows4_kernel, dim3((a.n_s + 255) / 256), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(em_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

// job-array launches (lock-step batch): every job of one launch, grid.y = job
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
    int mx = 0, mx_n = 0, mx_c = 0;
    for (int i = 0; i < cnt; ++i) {
      J.job[i] = jobs[b + i];
      const int t = jobs[b + i].n * jobs[b + i].C;
      mx = t > mx ? t : mx; mx_n = jobs[b + i].n > mx_n ? jobs[b + i].n : mx_n; mx_c = jobs[b + i].C > mx_c ? jobs[b + i].C : mx_c;
    }
    if (mx <= 0) continue;
    if (mx_c <= PROJ_CMAX) hipLaunchKernelGGL(proj_rows_jobs_kernel, dim3((mx_n + 255) / 256, cnt), dim3(256), 0, st, J);
    else hipLaunchKernelGGL(proj_jobs_kernel, dim3((mx + 255) / 256, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_em_weight_jobs(const WeightArgs* jobs, int n, hipStream_t st) {
  for (int b = 0; b < n; b += kMaxKnnJobs) {
    const int cnt = n - b < kMaxKnnJobs ? n - b : kMaxKnnJobs;
    WeightJobs J;
    int mx = 0, mx_n = 0;
    bool rows = true, hist = true;
    for (int i = 0; i < cnt; ++i) {
      J.job[i] = jobs[b + i];
      const int t = jobs[b + i].n_s * jobs[b + i].K;
      mx = t > mx ? t : mx; mx_n = jobs[b + i].n_s > mx_n ? jobs[b + i].n_s : mx_n;
      rows = rows && jobs[b + i].K == 4 && jobs[b + i].C <= WEIGHT_CMAX;
      hist = hist && jobs[b + i].s_hist != nullptr && jobs[b + i].K == 4 && jobs[b + i].C <= HW_CMAX;
    }
    if (mx <= 0) continue;
    if (hist) hipLaunchKernelGGL(em_weight_hist4_jobs_kernel, dim3((mx_n + 255) / 256, cnt), dim3(256), 0, st, J);
    else if (rows) hipLaunchKernelGGL(em_weight_rows4_jobs_kernel, dim3((mx_n + 255) / 256, cnt), dim3(256), 0, st, J);
    else hipLaunchKernelGGL(em_weight_jobs_kernel, dim3((mx + 255) / 256, cnt), dim3(256), 0, st, J);
  }
  return hipGetLastError();
}

hipError_t launch_fused_labels(const WeightArgs& a, uint32_t* out, hipStream_t st) {
  if (a.n_s <= 0) return hipSuccess;
  hipLaunchKernelGGL(fused_label_kernel, dim3((a.n_s + 255) / 256), dim3(256), 0, st, a, out);
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
