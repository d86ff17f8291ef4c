// solve_kernels.hip -- one LM evaluation of the inner solve and the device-resident LM step
// (gfx950, wave64).
//
//   accumulate*  : GICPCostFunction::Evaluate + LocalParameterizationSE3 + losses + Ceres' Corrector,
//                  summed to 28 doubles (6x6 J^T J upper triangle, J^T r, cost)
//                                                              gicp_cost_function.h:27-73
//   lm_step*     : ceres::Solve's trust-region step (csrc/lm.hpp) em_icp.hpp:162-177
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
#include <string.h>

#define SICP_HD __host__ __device__
#include "kernels.h"
#include "device_geometry.hpp"

namespace sicp {
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

// ---- the batched inner solve as a graph with explicit nodes --------------------------------------
static void* accumulate_batch_fn(int K) {
  static const int pf = [] { const char* e = getenv("SICP_ACC_BATCH_PF"); return e ? atoi(e) : 4; }();  // tuning aid
  switch (K) {
    case 1: return pf == 8 ? (void*)accumulate_batch_kernel<1, 256, 8> : (void*)accumulate_batch_kernel<1, 256, 4>;
    case 4: return pf == 8 ? (void*)accumulate_batch_kernel<4, 256, 8> : (void*)accumulate_batch_kernel<4, 256, 4>;
    case 20: return pf == 8 ? (void*)accumulate_batch_kernel<20, 256, 8> : (void*)accumulate_batch_kernel<20, 256, 4>;
    default: return nullptr;
  }
}

static void batch_node_params(int K, const BatchArgs** arg, void** slot, int n, int max_nb, hipKernelNodeParams& acc, hipKernelNodeParams& step) {
  slot[0] = (void*)arg;
  memset(&acc, 0, sizeof acc);
  acc.func = accumulate_batch_fn(K);
  acc.gridDim = dim3(max_nb, n);
  acc.blockDim = dim3(256);
  acc.kernelParams = slot;
  memset(&step, 0, sizeof step);
  step.func = (void*)lm_step_batch_kernel;
  step.gridDim = dim3(n);
  step.blockDim = dim3(64);
  step.kernelParams = slot;
}

void batch_graph_destroy(BatchGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr; g.graph = nullptr; g.len = 0;
}

hipError_t batch_graph_prepare(BatchGraph& g, int K, const BatchArgs* batch, int n, int max_nb, int len, int* built) {
  *built = 0;
  if (n <= 0 || max_nb <= 0 || len < 1 || len > kMaxBatchLen || !accumulate_batch_fn(K)) return hipErrorInvalidValue;
  const BatchArgs* arg = batch;
  void* slot[1];
  hipKernelNodeParams pa, ps;
  batch_node_params(K, &arg, slot, n, max_nb, pa, ps);
  if (g.exec && g.K == K && g.len == len && g.batch == batch) {
    if (g.n == n && g.max_nb == max_nb) return hipSuccess;
    hipError_t e = hipSuccess;
    for (int b = 0; b < len && e == hipSuccess; ++b) {
      e = hipGraphExecKernelNodeSetParams(g.exec, g.acc[b], &pa);
      if (e == hipSuccess) e = hipGraphExecKernelNodeSetParams(g.exec, g.step[b], &ps);
    }
    if (e == hipSuccess) { g.n = n; g.max_nb = max_nb; return hipSuccess; }
    (void)hipGetLastError();  // fall through: rebuild
  }
  batch_graph_destroy(g);
  hipError_t e = hipGraphCreate(&g.graph, 0);
  if (e != hipSuccess) return e;
  hipGraphNode_t prev = nullptr;
  for (int b = 0; b < len; ++b) {
    e = hipGraphAddKernelNode(&g.acc[b], g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &pa);
    if (e != hipSuccess) return e;
    e = hipGraphAddKernelNode(&g.step[b], g.graph, &g.acc[b], 1, &ps);
    if (e != hipSuccess) return e;
    prev = g.step[b];
  }
  e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) return e;
  g.K = K; g.len = len; g.n = n; g.max_nb = max_nb; g.batch = batch;
  *built = 1;
  return hipSuccess;
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

// test hook (sicp_se3_device): the SE(3) code of the device-resident solve, one lane per item
__global__ __launch_bounds__(64) void se3_ops_kernel(int op, int n, const double* __restrict__ in, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a[14], o[7];
  const int n_in = op == 0 ? 6 : (op == 2 ? 13 : (op == 3 ? 14 : 7)), n_out = op == 1 ? 6 : 7;
  for (int k = 0; k < n_in; ++k) a[k] = in[(size_t)i * n_in + k];
  switch (op) {
    case 0: se3::exp(a, o); break;
    case 1: se3::log(a, o); break;
    case 2: se3::plus(a, a + 7, o); break;
    case 3: se3::mul(a, a + 7, o); break;
    default: se3::inverse(a, o); break;
  }
  for (int k = 0; k < n_out; ++k) out[(size_t)i * n_out + k] = o[k];
}

hipError_t launch_se3_ops(int op, int n, const double* in, double* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(se3_ops_kernel, dim3((n + 63) / 64), dim3(64), 0, st, op, n, in, out);
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

}  // namespace sicp
