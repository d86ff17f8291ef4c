// solve_kernels.hip -- one LM evaluation of the inner solve and the device-resident LM step
// (gfx950, wave64).
//
//   accumulate*  : GICPCostFunction::Evaluate + LocalParameterizationSE3 + losses + Ceres' Corrector,
//                  summed to 28 doubles (6x6 J^T J upper triangle, J^T r, cost)
//                                                              gicp_cost_function.h:27-73
//   lm_step*     : ceres::Solve's trust-region step (csrc/lm.hpp) em_icp.hpp:162-177
//
// Design notes (MI355X).  The correspondence slots of a pair are cut into CHUNKS of 512 m groups of
// SG slots (SG = 4 slots of one source point for K = 4 / 20, 2 source points for K = 1; m = 1 up to
// 2M slots): a chunk is what one 256-lane workgroup sums -- every lane takes 2 m groups, 256 apart, in
// ascending order; every wave then sums its own lanes (LDS transpose + DPP), and the four waves' sums
// are joined in a fixed order into column `chunk` of partials[28][n_chunks].  The chunk is the unit of
// reproducibility: ONE kernel evaluates it for every caller (a pair alone is a batch of one), so a
// pair gets the same bits alone and in a batch, run after run.  No floating-point atomics.  Points are
// read as 48-byte records {x, y, z f32 | nx, ny, nz f64} -- one gather per target instead of six.
// Nothing here is GEMM shaped: no MFMA.  The file is compiled with -ffp-contract=off; fused
// multiply-adds are re-enabled per function where the float64 algebra only needs tolerance-level parity.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SICP_HD __host__ __device__
#include "kernels.h"
#include "device_geometry.hpp"
#include "fast_log.hpp"

namespace sicp {
// The float64 arithmetic of an evaluation is written with EXPLICIT fused multiply-adds (the file is compiled
// with -ffp-contract=off): the same source then is the same instruction sequence in every kernel that inlines
// it -- the batched kernel and the persistent one-pair solve agree bit for bit by construction.  (With
// `#pragma clang fp contract(fast)` each instantiation picks its own FMA pairs: round 2 saw a second kernel
// over the same chunks differ in the last bits.)
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

__device__ __forceinline__ double rcp_newton(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = FMA(r, FMA(-d, r, 1.0), r);
  r = FMA(r, FMA(-d, r, 1.0), r);
  return r;
}

// Everything that only depends on the SOURCE point (shared by the slots of one source point) is
// computed once per group.
struct SrcTerms {
  double qx, qy, qz;  // R p_s + t
  double mx, my, mz;  // m = R n_s
};

__device__ __forceinline__ void src_terms(const Pose& P, double psx, double psy, double psz, double nsx, double nsy, double nsz, SrcTerms& s) {
  const double* R = P.R;
  s.qx = FMA(R[2], psz, FMA(R[1], psy, FMA(R[0], psx, P.t[0])));
  s.qy = FMA(R[5], psz, FMA(R[4], psy, FMA(R[3], psx, P.t[1])));
  s.qz = FMA(R[8], psz, FMA(R[7], psy, FMA(R[6], psx, P.t[2])));
  s.mx = FMA(R[2], nsz, FMA(R[1], nsy, R[0] * nsx));
  s.my = FMA(R[5], nsz, FMA(R[4], nsy, R[3] * nsx));
  s.mz = FMA(R[8], nsz, FMA(R[7], nsy, R[6] * nsx));
}

// residual r = res^T A^-1 res and its local Jacobian for one correspondence (gicp_cost_function.h:27-73
// chained with LocalParameterizationSE3, closed form for C = I - (1-eps) n n^T: SURVEY appendix B).
//
// A = C_t + R C_s R^T = 2 I - k (n n^T + m m^T), k = 1 - eps, n = n_t, m = R n_s: twice the identity minus a
// rank-2 term, so a = A^-1 res needs no 3x3 inverse (the reference's Matrix3d::inverse(): 27 operations for
// cofactors, determinant and the adjugate product).  With U = [n m] and unit n, m (Woodbury):
//     A^-1 = 1/2 I + 1/2 U G^-1 U^T,   G = (2/k) I_2 - U^T U = [[gw, -d], [-d, gw]],   gw = 2/k - 1,  d = n.m,
//     G^-1 = 1 / ((gw - d)(gw + d)) [[gw, d], [d, gw]]
//     2 a = res + h ((gw al + d be) n + (d al + gw be) m),   al = n.res, be = m.res, h = 1 / ((gw - d)(gw + d))
// -- 29 operations instead of 44 and 6 fewer live values per source point, with the same conditioning: both
// forms lose cond(A) ~ 1/(2 eps) digits when n and m are parallel (checked against 40-digit arithmetic: equal
// worst-case and median errors at eps = 1e-3 and 1e-6).  (gw - d)(gw + d) is formed as a product of the two
// factors: gw^2 - d^2 would cancel.
__device__ __forceinline__ void corr_eval_src(const Pose& P, double one_m_eps, double gw, const SrcTerms& s, double psx, double psy,
                                              double psz, double nsx, double nsy, double nsz, double ptx, double pty,
                                              double ptz, double ntx, double nty, double ntz, Corr& o) {
  const double* R = P.R;
  const double rx = ptx - s.qx, ry = pty - s.qy, rz = ptz - s.qz;
  const double al = FMA(ntz, rz, FMA(nty, ry, ntx * rx));
  const double be = FMA(s.mz, rz, FMA(s.my, ry, s.mx * rx));
  const double d = FMA(ntz, s.mz, FMA(nty, s.my, ntx * s.mx));
  // Everything from here to the Jacobian is carried at TWICE its value (a2 = 2 a, b2 = 2 b: the factors 1/2 of
  // A^-1 are never applied), and powers of two are exact: J = 2 [-b; b x c] = [-b2; b2 x c] comes out
  // unscaled, r = res . a = 1/2 res . a2 costs one multiplication by 1/2.
  const double h = rcp_newton((gw - d) * (gw + d));  // (gw - d)(gw + d) in [~4 eps, ~(2/k)^2]
  const double ga = h * FMA(d, be, gw * al), de = h * FMA(gw, be, d * al);
  const double ax = FMA(de, s.mx, FMA(ga, ntx, rx));
  const double ay = FMA(de, s.my, FMA(ga, nty, ry));
  const double az = FMA(de, s.mz, FMA(ga, ntz, rz));
  o.r = 0.5 * FMA(rz, az, FMA(ry, ay, rx * ax));
  const double bx = FMA(R[6], az, FMA(R[3], ay, R[0] * ax));  // b2 = R^T a2
  const double by = FMA(R[7], az, FMA(R[4], ay, R[1] * ax));
  const double bz = FMA(R[8], az, FMA(R[5], ay, R[2] * ax));
  // b x c with c = p_s + C_s b = p_s + b - (1-eps)(n_s . b) n_s: b x b = 0, so c may leave out its b
  const double nb = (0.5 * one_m_eps) * FMA(nsz, bz, FMA(nsy, by, nsx * bx));
  const double cx = FMA(-nb, nsx, psx);
  const double cy = FMA(-nb, nsy, psy);
  const double cz = FMA(-nb, nsz, psz);
  o.J[0] = -bx; o.J[1] = -by; o.J[2] = -bz;
  o.J[3] = FMA(by, cz, -(bz * cy));
  o.J[4] = FMA(bz, cx, -(bx * cz));
  o.J[5] = FMA(bx, cy, -(by * cx));
}

// g = sqrt(v) and h = 1 / (2 sqrt(v)) together, for v > 0: v_rsq_f64 and two Goldschmidt steps (9
// instructions; a library rsqrt plus the two products is 12 and needs the special-case compares)
__device__ __forceinline__ void sqrt_and_half_rsqrt(double v, double& g, double& h) {
  const double y = __builtin_amdgcn_rsq(v);
  g = v * y;
  h = 0.5 * y;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
}

// The robustifier of one residual at s = r^2 (em_icp.hpp:109-117, gicp.hpp:98-104, semantic_icp.hpp:96;
// Ceres CauchyLoss / ScaledLoss / ComposedLoss, sqloss.h); b = a^2, c = 1 / b.  rho2 < 0 for all of the
// reference's stacks, so Ceres' Corrector scales residual and Jacobian by sqrt(rho1).  Split in two: rho1
// (what the 27 Gauss-Newton sums need) now, and the argument `sum` of the logarithm of rho0 = w b log(sum),
// which the caller evaluates for all slots of a group together (log_group below).
template <bool SQLOSS>
__device__ __forceinline__ void loss_rho1(double c, double s, double w, double& sum, double& rho1) {
  if (SQLOSS) {
    const double v = s + 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
    double g0, g1;
    sqrt_and_half_rsqrt(v, g0, g1);
    sum = FMA(g0, c, 1.0);
    rho1 = (w * fmax(2.2250738585072014e-308, rcp_newton(sum))) * g1;
  } else {
    sum = FMA(s, c, 1.0);
    rho1 = fmax(2.2250738585072014e-308, rcp_newton(sum));
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

// Hand-offs between workgroups of one launch (the persistent solve; the LM step inside the accumulate launch): write-through stores and agent-scope loads of
// at most 8 bytes -- visible across CUs and XCDs without a release / acquire fence (cdna_hip_programming.md
// guideline 16, recipe R1: payload sc1 -> the storing wave drains -> one lane stores the flag).
typedef unsigned long long u64;
__device__ __forceinline__ void coherent_store_f64(SICP_GLOBAL double* p, double v) {
  __hip_atomic_store((SICP_GLOBAL u64*)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coherent_load_f64(const SICP_GLOBAL double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const SICP_GLOBAL u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// a value every lane holds identically -> scalar registers (it costs no VGPRs and feeds the scalar
// operand of the vector instructions that use it)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void pose_to_sgprs(Pose& P) {
#pragma unroll
  for (int k = 0; k < 9; ++k) P.R[k] = uniform_f64(P.R[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) P.t[k] = uniform_f64(P.t[k]);
}

// ---- one group of SG consecutive slots: what a lane loads, and what it computes from it -----------
template <int K>
struct GroupShape {
  static constexpr int SG = acc_slots_per_group(K);  // slots per group
  static constexpr int NS = (K % 4 == 0) ? 1 : 2;    // source points per group
};

template <int K>
struct Group {
  static constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS;
  int j[SG];
  double w[SG];
  float sx[NS], sy[NS], sz[NS], tx[SG], ty[SG], tz[SG];
  double snx[NS], sny[NS], snz[NS], tnx[SG], tny[SG], tnz[SG];
};

// uniform (per pair) inputs of the loads and of the arithmetic
struct LoadCtx {
  const SICP_GLOBAL int* idx;
  const SICP_GLOBAL double* w;       // nullable: weight 1
  const SICP_GLOBAL PointRec* srec;
  const SICP_GLOBAL PointRec* trec;
  const SICP_GLOBAL char* sdense;    // nullable: the source records as dense arrays (kernels.h: dense_rec_*), pitch n_s
  int n_s, total;
};
struct MathCtx {
  Pose P;
  double one_m_eps, gw, loss_b, loss_c;  // gw = 2 / (1 - eps) - 1 (corr_eval_src)
  unsigned log_table;                    // LDS byte address of the logarithm's table (fast_log.hpp)
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// The streams of a pass (indices, weights, source records) are read once; loading them non-temporal
// (-DSICP_STREAM_NT), so that they would not push the gathered target records out of L2, measured SLOWER:
// 110.5 vs 108.0 us over 32 pairs, 876 vs 810 us over 256.
#if defined(SICP_STREAM_NT)
#define SICP_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SICP_STREAM_LOAD(p) (*(p))
#endif
struct RecMid { double nz; float x, y; };
__device__ __forceinline__ void load_rec(const SICP_GLOBAL PointRec* r, float& x, float& y, float& z, double& nx, double& ny, double& nz) {
  const SICP_GLOBAL char* p = (const SICP_GLOBAL char*)r;
  const v2d a = SICP_STREAM_LOAD((const SICP_GLOBAL v2d*)p);          // nx ny
  const v4f b = SICP_STREAM_LOAD((const SICP_GLOBAL v4f*)(p + 16));   // nz (two floats) x y
  const float c = SICP_STREAM_LOAD((const SICP_GLOBAL float*)(p + 32));
  nx = a.x; ny = a.y;
  nz = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
  x = b.z; y = b.w; z = c;
}

// Unconditional form for the pipelined kernel: always one vector load (from the last group when g lies
// past the end; the index buffer is allocated with slack, so the SG - 1 entries a ragged last group
// reads beyond `total` exist), and the "past the end -> -1" fix-up is applied where the values are
// consumed (fix_idx) -- a select here would be a use of the load and make the compiler wait for it.
template <int K>
__device__ __forceinline__ void load_idx_raw(const LoadCtx& L, int g, int (&j)[GroupShape<K>::SG]) {
  constexpr int SG = GroupShape<K>::SG;
  const int last = ((L.total - 1) / SG) * SG;
  const int e0 = max(min(g * SG, last), 0);
  if (SG == 4) {
    const v4i v = SICP_STREAM_LOAD((const SICP_GLOBAL v4i*)(L.idx + e0));
    j[0] = v.x; j[1] = v.y; j[SG - 2] = v.z; j[SG - 1] = v.w;
  } else {
    const v2i v = SICP_STREAM_LOAD((const SICP_GLOBAL v2i*)(L.idx + e0));
    j[0] = v.x; j[SG - 1] = v.y;
  }
}
template <int K>
__device__ __forceinline__ void fix_idx(int total, int g, int (&j)[GroupShape<K>::SG]) {
  constexpr int SG = GroupShape<K>::SG;
#pragma unroll
  for (int c = 0; c < SG; ++c) j[c] = g * SG + c < total ? j[c] : -1;
}

// log(sum) for the SG slots of a group at once (fast_log.hpp).  The table entries are fetched from LDS by
// ONE inline-asm statement that also waits for them: the compiler neither sees an LDS access it would have
// to order against the LDS-DMA prefetch in flight (it would drain that prefetch with s_waitcnt vmcnt(0) at
// every logarithm), nor gets to touch the destination registers before the data has landed.  The table is
// never written after the kernel's prologue, so there is nothing to order.
template <int SG>
__device__ __forceinline__ void log_group(unsigned table, const double (&x)[SG], double (&lg)[SG]) {
  static_assert(SG == 2 || SG == 4, "group shapes of the accumulate kernel");
  unsigned a[SG];
  v4i e[SG];
#pragma unroll
  for (int c = 0; c < SG; ++c) a[c] = table + log_entry_offset(x[c]);
  if constexpr (SG == 4) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
  } else {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(e[0]), "=&v"(e[1]) : "v"(a[0]), "v"(a[1]));
  }
#pragma unroll
  for (int c = 0; c < SG; ++c) lg[c] = log_from_entry(x[c], __hiloint2double(e[c].y, e[c].x), __hiloint2double(e[c].w, e[c].z));
}

template <int K, bool SQLOSS>
__device__ __forceinline__ void compute_group(const MathCtx& M, const Group<K>& G, double (&acc)[28]) {
  constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS;
  SrcTerms st;
  double sum[SG], w[SG];
  // A gated-out slot (index -1; also the slots of a ragged last group past the end, whose weights are
  // whatever lies behind the buffer) is evaluated on target 0 and WEIGHTED by exactly zero instead of being
  // branched around: x + (+-0 * finite) == x bit for bit, and a divergent skip makes the compiler copy
  // all 28 accumulators at the join.  (Without SQLoss the reference has no ScaledLoss -- w is 1 -- and the
  // weight only carries the gate.)
#pragma unroll
  for (int c = 0; c < SG; ++c) w[c] = G.j[c] < 0 ? 0.0 : G.w[c];
#pragma unroll
  for (int c = 0; c < SG; ++c) {
    const int s = NS == 1 ? 0 : c / (SG / NS);
    if (c % (SG / NS) == 0) src_terms(M.P, G.sx[s], G.sy[s], G.sz[s], G.snx[s], G.sny[s], G.snz[s], st);
    Corr cr;
    corr_eval_src(M.P, M.one_m_eps, M.gw, st, G.sx[s], G.sy[s], G.sz[s], G.snx[s], G.sny[s], G.snz[s], G.tx[c], G.ty[c], G.tz[c], G.tnx[c],
                  G.tny[c], G.tnz[c], cr);
    double rho1;
    loss_rho1<SQLOSS>(M.loss_c, cr.r * cr.r, w[c], sum[c], rho1);
    if (!SQLOSS) rho1 *= w[c];
    int o = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const double jp = rho1 * cr.J[p];
#pragma unroll
      for (int q = p; q < 6; ++q, ++o) acc[o] = FMA(jp, cr.J[q], acc[o]);
      acc[21 + p] = FMA(jp, cr.r, acc[21 + p]);
    }
  }
  // cost = 1/2 sum rho0, rho0 = w b log(sum) (w = 1 without SQLoss: semantic_icp.hpp:96 has no ScaledLoss)
  double lg[SG];
  log_group<SG>(M.log_table, sum, lg);
#pragma unroll
  for (int c = 0; c < SG; ++c) acc[27] = FMA(0.5 * w[c], M.loss_b * lg[c], acc[27]);
}

// ------------------------------------------------------------------------------------------
// The batched evaluation (every path: a pair alone is a batch of one).  ONE launch evaluates the
// current LM pose of every pair of the tick that still iterates.
//
// Work split.  gridDim.x persistent workgroups (SICP_ACC_OCC per CU) split the chunks of the RUNNING
// pairs -- counted on the device from the LM states, so a pair that finished inside a tick costs
// nothing and unbalances nothing -- into equal CONTIGUOUS ranges.  A range is a few segments (runs of
// chunks of one pair; mostly one).  Inside a segment the pair's constants live in SGPRs and a lane's
// group index advances by the workgroup size: nothing is located, re-read or re-broadcast per step.
//
// Pipeline.  At two waves per SIMD plain wave interleaving hides about half of an index -> gather
// chain, and a second register set for the next group does not fit beside the 28 accumulators.  So the
// next group's TARGET records (36 gathered bytes x SG per lane) are fetched by LDS-DMA
// (global_load_lds: no destination registers) into a wave-private staging area while the current
// group is computed from registers; only the next group's indices, weights and source record travel
// through registers.  Per lane and step:
//     wait for everything issued a step ago | staging area -> registers | issue: LDS-DMA of group
//     t+1's targets, loads of its weights / source, indices of group t+2 | compute group t
//
// Reduction.  A wave sums its OWN 64 lanes' 28 accumulators at the end of every chunk: transposed through
// a wave-private LDS tile, RED_ROWS rows at a time -- lane (r, q) adds up quarter q of row r in ascending
// order, one DPP quad reduction joins the quarters -- and parks the 28 sums in LDS.  Every COMB_CHUNKS
// chunks (and at the end of a segment) ONE workgroup barrier lets the 256 threads join the four waves'
// sums of each parked chunk as (w0 + w1) + (w2 + w3) and write column `chunk` of partials[28][n_chunks].
// A block-wide reduction per chunk (barrier | write | barrier | read, twice) cost 21 % of the launch, far
// more than its instructions: every barrier re-synchronises four waves whose memory waits differ.
// ------------------------------------------------------------------------------------------
// what the walk needs of every pair of the launch (LDS, 12 bytes per pair); everything else is read
// from the argument array when a workgroup enters a segment of the pair
struct PairSlot {
  int running, n_chunks, item_begin;
};

#define SICP_LDS __attribute__((address_space(3)))
__device__ const double kLogTable[2 * kLogTableEntries] = {
#include "log_table.inc"
};
constexpr int LOG_TABLE_BYTES = 16 * kLogTableEntries;  // the workgroup's copy in LDS
constexpr int STAGE_SLOT_BYTES = 64 * 36;  // one target record of every lane of a wave: 1024 + 1024 + 256

__device__ __forceinline__ void lds_dma_rec(const SICP_GLOBAL PointRec* r, SICP_LDS char* slot) {
  const SICP_GLOBAL char* p = (const SICP_GLOBAL char*)r;
  __builtin_amdgcn_global_load_lds((const SICP_GLOBAL void*)p, (SICP_LDS void*)slot, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const SICP_GLOBAL void*)(p + 16), (SICP_LDS void*)(slot + 1024), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const SICP_GLOBAL void*)(p + 32), (SICP_LDS void*)(slot + 2048), 4, 0, 0);
}

__device__ __forceinline__ void lds_read_rec(const SICP_LDS char* slot, int lane, float& x, float& y, float& z, double& nx, double& ny, double& nz) {
  const v2d a = *(const SICP_LDS v2d*)(slot + 16 * lane);
  const v4f b = *(const SICP_LDS v4f*)(slot + 1024 + 16 * lane);
  const float c = *(const SICP_LDS float*)(slot + 2048 + 4 * lane);
  nx = a.x; ny = a.y;
  nz = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
  x = b.z; y = b.w; z = c;
}

// the part of a group that travels through registers
template <int K>
struct GroupRegs {
  static constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS;
  int j[SG];
  double w[SG];
  float sx[NS], sy[NS], sz[NS];
  double snx[NS], sny[NS], snz[NS];
};

template <int K>
__device__ __forceinline__ void load_regs(const LoadCtx& L, int last, int g, GroupRegs<K>& G) {
  constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS;
  const int e0 = max(min(g * SG, last), 0);  // whole vectors: the buffers carry slack past `total` (DevBuf)
  if (L.w) {
    if (SG == 4) {
      const v2d a = SICP_STREAM_LOAD((const SICP_GLOBAL v2d*)(L.w + e0)), b = SICP_STREAM_LOAD((const SICP_GLOBAL v2d*)(L.w + e0 + 2));
      G.w[0] = a.x; G.w[1] = a.y; G.w[SG - 2] = b.x; G.w[SG - 1] = b.y;
    } else {
      const v2d a = SICP_STREAM_LOAD((const SICP_GLOBAL v2d*)(L.w + e0));
      G.w[0] = a.x; G.w[SG - 1] = a.y;
    }
  } else {
#pragma unroll
    for (int c = 0; c < SG; ++c) G.w[c] = 1.0;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int i = max(min((e0 + s * (SG / NS)) / K, L.n_s - 1), 0);
    if (L.sdense) {  // (wave-uniform) the stream of source points from the dense arrays: 36 instead of 48 bytes per point
      const SICP_GLOBAL char* p = L.sdense + 16 * (size_t)i;
      const size_t n = (size_t)L.n_s;
      const v2d a = SICP_STREAM_LOAD((const SICP_GLOBAL v2d*)p);
      const v4f b = SICP_STREAM_LOAD((const SICP_GLOBAL v4f*)(p + 16 * n));
      const float c = SICP_STREAM_LOAD((const SICP_GLOBAL float*)(L.sdense + 32 * n + 4 * (size_t)i));
      G.snx[s] = a.x; G.sny[s] = a.y;
      G.snz[s] = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
      G.sx[s] = b.z; G.sy[s] = b.w; G.sz[s] = c;
    } else {
      load_rec(L.srec + i, G.sx[s], G.sy[s], G.sz[s], G.snx[s], G.sny[s], G.snz[s]);
    }
  }
}

// The 28 sums of ONE WAVE -> column `col` of partials[28][n_cols].  tile = this wave's
// [RED_ROWS][RED_STRIDE] doubles of LDS.  Lane l writes its accumulators into column l of the tile; lane
// (r = l / 4, q = l % 4) then adds elements 16 q .. 16 q + 15 of row r in ascending order, and the four
// quarters of a row are joined as (q0 + q1) + (q2 + q3).  Same-wave LDS accesses execute in program
// order, so no barrier is involved: the fences only pin the compiler.  RED_STRIDE = 66: even (16-byte
// reads stay aligned) and not a multiple of the 32 8-byte banks (the quarter reads of the 56 active
// lanes spread over all banks: ~4 lanes per 16-byte slot, the minimum for 64 lanes x 16 bytes).
#ifndef SICP_RED_ROWS
#define SICP_RED_ROWS 14
#endif
constexpr int RED_ROWS = SICP_RED_ROWS;  // divides 28; RED_ROWS * 4 <= 64 lanes
constexpr int RED_STRIDE = 66;
static_assert(28 % RED_ROWS == 0 && RED_ROWS * 4 <= 64, "one lane per (row, quarter)");
#ifndef SICP_COMB_CHUNKS
#define SICP_COMB_CHUNKS 8
#endif
constexpr int COMB_CHUNKS = SICP_COMB_CHUNKS;  // chunks whose four wave sums wait in LDS for one combining pass
constexpr unsigned kRunsPerWorkgroupForDynamic = 8u;  // accumulate launches with at least this many 8-chunk runs per workgroup hand their chunks out dynamically
__device__ __forceinline__ void wave_reduce(const double (&acc)[28], SICP_LDS double* tile, SICP_LDS double* out28, int lane) {
  const int r = lane >> 2, q = lane & 3;
  const SICP_LDS v2d* mine = (const SICP_LDS v2d*)(tile + min(r, RED_ROWS - 1) * RED_STRIDE + 16 * q);
#pragma unroll
  for (int p0 = 0; p0 < 28; p0 += RED_ROWS) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int k = 0; k < RED_ROWS; ++k) tile[k * RED_STRIDE + lane] = acc[p0 + k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    v2d v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = mine[m];
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) { s += v[m].x; s += v[m].y; }
    s += dpp_f64<0xB1>(s);  // quad_perm:[1,0,3,2]
    s += dpp_f64<0x4E>(s);  // quad_perm:[2,3,0,1]
    if (q == 0 && r < RED_ROWS) out28[p0 + r] = s;
    __builtin_amdgcn_wave_barrier();
  }
}

// One SEGMENT: chunks [chunk_lo, chunk_lo + n_here) of one pair, evaluated by the calling workgroup at the pose
// in M; their 28 sums go to columns chunk_lo ... of partials[28][n_chunks].  The one routine both the batched
// kernel and the persistent one-pair solve run, so a chunk's sums are the same bits in either.
template <int K, bool SQLOSS, int BS>
__device__ __forceinline__ void accumulate_segment(const LoadCtx& L, const MathCtx& M, int chunk_lo, int n_here, int n_chunks, int steps, int chunk_groups,
                                                   SICP_GLOBAL double* partials, SICP_LDS char* stage, SICP_LDS double* tile,
                                                   double (&comb)[COMB_CHUNKS][BS / 64][28], int lane, int wave) {
  constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS;
  const int nsteps = n_here * steps;
  const int last = ((L.total - 1) / SG) * SG;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  // Two sets of the register-borne part alternate as "current" and "next" (the step below is
  // instantiated twice with the roles swapped): a copy `current = next` would be scheduled into the
  // arithmetic and wait there for the very loads it is meant to overlap.
  GroupRegs<K> R0, R1;
  int g = chunk_lo * chunk_groups + (int)threadIdx.x;  // this lane's group; + BS per step
  int t = 0, in_chunk = 0, chunk = chunk_lo, parked = 0;  // parked: chunks whose wave sums sit in comb[]
  // join the four waves' sums of the parked chunks [chunk - parked, chunk) -> their columns.  Raw barriers:
  // a __syncthreads would also wait for the LDS-DMA / loads in flight.
  auto flush_parked = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int e = threadIdx.x; e < parked * 28; e += BS) {
      const int c = e / 28, k = e - 28 * c;
#if defined(SICP_LM_STEP_IN_LAUNCH)  // (write-through: the workgroup that finishes a pair LAST sums the columns of all of them inside this launch)
      coherent_store_f64(partials + (size_t)k * n_chunks + (chunk - parked + c), (comb[c][0][k] + comb[c][1][k]) + (comb[c][2][k] + comb[c][3][k]));
#else
      partials[(size_t)k * n_chunks + (chunk - parked + c)] = (comb[c][0][k] + comb[c][1][k]) + (comb[c][2][k] + comb[c][3][k]);
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    parked = 0;
  };
  auto issue_targets = [&](const int (&j)[SG]) {
#pragma unroll
#if defined(SICP_DEBUG_NOGATHER)  // developer aid: every gather reads target 0 (results are wrong)
    for (int c = 0; c < SG; ++c) lds_dma_rec(L.trec + (max(j[c], 0) & 0), stage + c * STAGE_SLOT_BYTES);
#else
    for (int c = 0; c < SG; ++c) lds_dma_rec(L.trec + max(j[c], 0), stage + c * STAGE_SLOT_BYTES);
#endif
  };
  // fill: indices of the first two groups, then targets / weights / source of the first
  load_idx_raw<K>(L, g, R0.j);
  load_idx_raw<K>(L, g + BS, R1.j);
  fix_idx<K>(L.total, g, R0.j);
  issue_targets(R0.j);
  load_regs<K>(L, last, g, R0);

  auto step = [&](GroupRegs<K>& cur, GroupRegs<K>& nxt) -> bool {
    // (A) everything issued a step ago has landed.  The empty asm statements "use" every register a
    // load of the previous step wrote: hipcc places its own (conservative, vmcnt(0)) wait for them
    // here, where nothing is in flight, instead of at their first arithmetic use below -- where it
    // would drain the loads issued in (B).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < SG; ++c) { asm volatile("" : "+v"(cur.w[c]), "+v"(nxt.j[c])); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      asm volatile("" : "+v"(cur.sx[s]), "+v"(cur.sy[s]), "+v"(cur.sz[s]), "+v"(cur.snx[s]), "+v"(cur.sny[s]), "+v"(cur.snz[s]));
    }
    Group<K> Gc;
#pragma unroll
    for (int c = 0; c < SG; ++c) {
      Gc.j[c] = cur.j[c]; Gc.w[c] = cur.w[c];
      lds_read_rec(stage + c * STAGE_SLOT_BYTES, lane, Gc.tx[c], Gc.ty[c], Gc.tz[c], Gc.tnx[c], Gc.tny[c], Gc.tnz[c]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) { Gc.sx[s] = cur.sx[s]; Gc.sy[s] = cur.sy[s]; Gc.sz[s] = cur.sz[s]; Gc.snx[s] = cur.snx[s]; Gc.sny[s] = cur.sny[s]; Gc.snz[s] = cur.snz[s]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area may be overwritten
    // (B) the next group's traffic, and the indices of the group after it.  Unconditional: a load
    // under a branch merges with the "not taken" value in a register copy, which is a use of the load
    // -- the compiler would wait for it right here.  Past the end of the segment the fetch is target
    // 0 of dead slots, and is never computed.
    const bool more = t + 1 < nsteps;
    fix_idx<K>(more ? L.total : 0, g + BS, nxt.j);
    issue_targets(nxt.j);
    load_regs<K>(L, last, g + BS, nxt);
    load_idx_raw<K>(L, g + 2 * BS, cur.j);
    // nothing of (B) may sink into the arithmetic: left alone, the scheduler issues the source-record
    // and index loads half way through / near the end of the step (shorter live ranges), which leaves
    // them a fraction of a step to land before (A) of the next step waits for them
    __builtin_amdgcn_sched_barrier(0);
    // (C) the arithmetic of the current group
#if defined(SICP_DEBUG_NOCOMPUTE)  // developer aid: the memory pipeline alone (every loaded value is consumed once)
#pragma unroll
    for (int c = 0; c < SG; ++c) acc[c] += (double)(Gc.tx[c] + Gc.ty[c] + Gc.tz[c]) + Gc.tnx[c] + Gc.tny[c] + Gc.tnz[c] + Gc.w[c] + (double)Gc.j[c];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[8 + s] += (double)(Gc.sx[s] + Gc.sy[s] + Gc.sz[s]) + Gc.snx[s] + Gc.sny[s] + Gc.snz[s];
#else
    compute_group<K, SQLOSS>(M, Gc, acc);
#endif
    if (++in_chunk == steps) {  // the chunk is complete
#if defined(SICP_DEBUG_NOREDUCE)  // developer aid: no LDS transpose / DPP, all 28 sums stay live
      {
        double chk = 0.0;
#pragma unroll
        for (int k = 0; k < 28; ++k) chk += acc[k];
        if (chk == 1.2345e300) partials[chunk] = chk;
      }
#else
      wave_reduce(acc, tile, (SICP_LDS double*)&comb[parked][wave][0], lane);
#endif
#pragma unroll
      for (int k = 0; k < 28; ++k) acc[k] = 0.0;
      in_chunk = 0;
      ++chunk;
#if !defined(SICP_DEBUG_NOREDUCE)
      if (++parked == COMB_CHUNKS || t + 1 == nsteps) flush_parked();
#endif
    }
#if !defined(SICP_DEBUG_NOSTREAM)  // developer aid: with it, every step re-reads the segment's first groups (cache hits)
    g += BS;
#endif
    ++t;
    return more;
  };
  for (;;) {
    if (!step(R0, R1)) break;
    if (!step(R1, R0)) break;
  }
}

// Fixed-order sum of the partial columns (layout [28][n]) by one workgroup of
// REDUCE_THREADS / 64 waves: wave (cq = wave / 4, rg = wave % 4) sums the rows rg, rg + 4, ... (7 of
// them) over the column part cq (all columns with 256 threads) -- lane l takes the columns l, l + 64,
// ... in ascending order, four columns' loads in flight before the first add -- and reduces across
// its lanes with wave_sum; with more than one part the parts of a row are joined in a fixed order.
// Everyone who needs the 28 sums of a pair (the LM step, sicp_accumulate, the host-loop solve) goes
// through this one routine, so they see the same bits.  Must be called by all REDUCE_THREADS threads
// of the block; the result is valid for thread 0 after the call (s_part is block-shared scratch).
#ifndef SICP_REDUCE_THREADS
#define SICP_REDUCE_THREADS 256  // more threads leave lm_feed (312 VGPRs) too few registers: 512 -> spills
#endif
constexpr int REDUCE_THREADS = SICP_REDUCE_THREADS, REDUCE_PARTS = REDUCE_THREADS / 256;
static_assert(REDUCE_PARTS == 1, "the column count need not divide");
// COHERENT: the columns were written by other workgroups of the SAME launch with write-through (sc1) stores and are
// read with agent-scope loads, which no cache of this CU / XCD can serve stale (the persistent solve's master).
template <bool COHERENT = false>
__device__ __forceinline__ void reduce_partials_block(const double* __restrict__ partials, int n, double (&s_part)[4][28], double (&o)[28]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rg = wave & 3, cq = wave >> 2;
  const int nq = n / REDUCE_PARTS, c0 = cq * nq, c1 = c0 + nq;
  const double* __restrict__ base = partials + (size_t)rg * n;  // row rg + 4 r lives at base + 4 r n
  double s[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) s[r] = 0.0;
  constexpr int DEEP = 4;  // column blocks in flight per trip (clamped loads past the end are issued too: keep it near n / 64)
  for (int b0 = c0 + lane; b0 < c1; b0 += 64 * DEEP) {
    double v[DEEP][7];
#pragma unroll
    for (int t = 0; t < DEEP; ++t) {
      const int b = min(b0 + 64 * t, c1 - 1);  // clamped: loaded unconditionally, added only when in range
#pragma unroll
      for (int r = 0; r < 7; ++r)
        v[t][r] = COHERENT ? coherent_load_f64((const SICP_GLOBAL double*)(base + (size_t)(4 * r) * n + b)) : __builtin_nontemporal_load(base + (size_t)(4 * r) * n + b);
    }
#pragma unroll
    for (int t = 0; t < DEEP; ++t)
      if (b0 + 64 * t < c1) {
#pragma unroll
        for (int r = 0; r < 7; ++r) s[r] += v[t][r];
      }
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const double sum = wave_sum(s[r]);
    if (lane == 0) s_part[cq][rg + 4 * r] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 28; ++k)
      o[k] = REDUCE_PARTS == 4 ? (s_part[0][k] + s_part[1][k]) + (s_part[2][k] + s_part[3][k])
           : REDUCE_PARTS == 2 ? s_part[0][k] + s_part[1][k] : s_part[0][k];
  }
}

// lm_feed's first test -- are all 28 sums finite -- by the 64 lanes of one wave from the block-shared sums reduce_partials_block
// has just left in s_part[0] (o[k] IS s_part[0][k]): the same predicate per sum, one compare instead of 28 in the one lane
// that steps the machine (179 of the LM step's instructions; one pair alone: -0.035 ms per align).  Call with the whole wave,
// after reduce_partials_block.
__device__ __forceinline__ int sums_all_finite(const double (&s_part)[4][28]) {
  static_assert(REDUCE_PARTS == 1, "s_part[0] holds the sums themselves");
  const int lane = threadIdx.x & 63;
  const double v = s_part[0][min(lane, 27)];
  return __all(v - v == 0.0) ? 1 : 0;
}

// ---- build-time experiment of round 6 (-DSICP_LM_STEP_IN_LAUNCH + SICP_LM_STEP_IN_LAUNCH=1 in the environment): the LM step
// INSIDE the accumulate launch -- the workgroup that delivers a pair's last columns sums them all and advances the pair's
// trust-region machine, the tick is [tick_prepare, accumulate x len] instead of [accumulate, lm_step_batch] x len.  Built,
// bit-identical to the product (all GPU tests), and SLOWER everywhere: 256-pair stream step 184.2 against 183.1 ms, 16 full-size
// pairs 59.5 against 48.9 ms per step, SE3-GICP 144 against 129 (profiles/r06/lm_step_in_launch_ab.json).  Every workgroup's range
// is about half a pair, so every pair completes at the END of the launch and its step (reduce 44 KB + ~9 us of one lane) extends
// the launch by what the separate kernel took; the drain + two barriers + one agent-scope add per segment and the 532 bytes of
// scratch lm_feed's 310 registers cost the kernel come on top.  The step's kernel was never the loss it looked like in the
// trace (55 us "present" per launch): while it waits for a SIMD the searches that share the chip make progress.
// Not part of the product build.
#if defined(SICP_LM_STEP_IN_LAUNCH)
// One LM step of a pair from inside the accumulate launch (one lane).  Not inlined: lm_feed wants ~310 registers, the
// accumulate kernel is compiled for two waves per SIMD (256), and this runs once per pair and evaluation -- whatever it
// spills must stay out of the kernel's hot loop.
__device__ __attribute__((noinline)) void lm_step_in_launch(LmState* lm, EvalIn* next, unsigned next_epoch, const double (&o)[28]) {
  LmCore st = *lm;
  lm_feed(st, lm->opt, o);
  st.pending = 0;  // the arrival counter of the NEXT launch starts from zero
  *static_cast<LmCore*>(lm) = st;
#pragma unroll
  for (int k = 0; k < 7; ++k) next->pose[k] = st.pose[k];
  next->status = st.status;
  next->epoch = next_epoch;
}

#endif

template <int K, bool SQLOSS, int BS>
__global__ __launch_bounds__(BS, SICP_ACC_OCC) void accumulate_staged_kernel(const BatchHeader* __restrict__ hdr, const BatchArgs* __restrict__ batch, int node) {
  constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS, NW = BS / 64;
  extern __shared__ __attribute__((aligned(16))) double smem[];  // ONE shared object: [reduction tiles | staging | logarithm table | per-pair walk state]
  __shared__ int total_running;
#if defined(SICP_LM_STEP_IN_LAUNCH)
  __shared__ int s_last;
#endif
  __shared__ double comb[COMB_CHUNKS][NW][28];
  static_assert(COMB_CHUNKS * NW >= 4 && BS == REDUCE_THREADS, "comb[] doubles as reduce_partials_block's scratch; the workgroup is its block");
  char* stage_all = reinterpret_cast<char*>(smem + NW * RED_ROWS * RED_STRIDE);
  char* log_tab = stage_all + NW * SG * STAGE_SLOT_BYTES;
  PairSlot* ctx = reinterpret_cast<PairSlot*>(log_tab + LOG_TABLE_BYTES);
  const int n_pairs = hdr->n_pairs;
  if (n_pairs <= 0) return;
  // this launch's epoch (pairs whose LM machine is stepped inside the launch: AccArgs::ein, see EvalIn in kernels.h)
  const unsigned epoch = (unsigned)uniform_i32((int)(hdr->epoch_base + (unsigned)(node & 0xffff)));
  (void)epoch;
  for (int k = threadIdx.x; k < kLogTableEntries; k += BS) reinterpret_cast<v2d*>(log_tab)[k] = reinterpret_cast<const v2d*>(kLogTable)[k];
  for (int p = threadIdx.x; p < n_pairs; p += BS) {
    const AccArgs& a = batch[p].a;
    PairSlot c;
#if defined(SICP_LM_STEP_IN_LAUNCH)
    if (a.ein) {  // (never the state itself: another workgroup of this launch may already have stepped it)
      const EvalIn& in = a.ein[epoch & 1u];
      c.running = in.epoch == epoch && in.status == LM_RUNNING;
    } else
#endif
    c.running = a.lm ? a.lm->status == LM_RUNNING : 1;
    c.n_chunks = acc_geometry(a.n_s * a.K, SG).n_chunks;
    c.item_begin = 0;
    ctx[p] = c;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = uniform_i32((int)(threadIdx.x >> 6));
  // item_begin[p] := number of chunks of RUNNING pairs before pair p (wave 0: a few pairs per lane, then
  // a scan over the lanes)
  if (threadIdx.x < 64) {
    const int per = (n_pairs + 63) / 64, p0 = lane * per, p1 = min(p0 + per, n_pairs);
    int mine = 0;
    for (int p = p0; p < p1; ++p) mine += ctx[p].running ? ctx[p].n_chunks : 0;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    int base = incl - mine;
    for (int p = p0; p < p1; ++p) {
      ctx[p].item_begin = base;
      base += ctx[p].running ? ctx[p].n_chunks : 0;
    }
    if (lane == 63) total_running = incl;
  }
  __syncthreads();
  // equal ranges; with fewer chunks than workgroups, one chunk each for the FIRST workgroups (they are
  // dispatched to different CUs; spreading b * T / G would put two working groups on some CUs and none
  // on others).  T <= 512 pairs x 1024 chunks and gridDim.x < 2^12: the products fit 32 bits.
  const unsigned T = (unsigned)total_running, G = gridDim.x, b = blockIdx.x;
  SICP_LDS char* stage = (SICP_LDS char*)stage_all + wave * (SG * STAGE_SLOT_BYTES);
  SICP_LDS double* tile = (SICP_LDS double*)smem + wave * (RED_ROWS * RED_STRIDE);
#if !defined(SICP_ACC_STATIC_RANGES) && !defined(SICP_DEV_PROBES)
  // Work split (round 6): the chunks of the running pairs are handed out in RUNS of consecutive chunks by an agent-scope
  // counter, not as equal static ranges.  In a tick the accumulate workgroups are dispatched beside the search kernels of the
  // side stream and start up to hundreds of microseconds apart (an 80 KB / 4 x 196-register workgroup only gets onto a CU as
  // search workgroups drain), and with static ranges the launch lasts as long as its latest starter: 184.0 -> 177.2 ms per
  // 256-pair step (+3.8 % correspondences/s), and 805 -> 793 us for the launch alone (profiles/r06/dynamic_runs_ab.json).
  // A chunk's column does not depend on who computes it: the same bits.  A run is 8 chunks (what is parked in LDS between two
  // combining passes); the next run's index is requested before the current run is processed (its round trip hides behind
  // the pipeline fill).  ONLY launches with at least 8 runs per workgroup are split this way: every run pays a pipeline fill and
  // two barriers, and with few runs per workgroup their granularity unbalances the launch -- shorter runs were measured too
  // (T / 4G chunks): 16 full-size pairs 56.9 against 44.4 ms per step, the open stream's ramp-up 1.9 K against 2.0 K pairs/s --
  // so smaller launches keep the equal contiguous ranges of rounds 2-5.  The two counters live in the header
  // (BatchHeader::pad_) and are reset by the last workgroup to leave -- everyone has made its last fetch by then.
  // (-DSICP_ACC_STATIC_RANGES: the equal contiguous ranges of rounds 2-5, for A/B.)
  unsigned* const next_run = reinterpret_cast<unsigned*>(const_cast<int*>(&hdr->pad_[0]));
  unsigned* const left = reinterpret_cast<unsigned*>(const_cast<int*>(&hdr->pad_[1]));
  constexpr unsigned run = (unsigned)COMB_CHUNKS;
  // (the same for every workgroup of the launch; kAccStaticRanges in `node`: the host asks for equal ranges -- a stream while scans
  //  are being uploaded: a launch whose workgroups all stay to the end leaves the tree-build kernels of the upload stream, whose sorts
  //  need LDS, no CU to get onto except between launches: 1.9 K instead of 2.0 K pairs/s end to end)
  const bool by_runs = !(node & kAccStaticRanges) && T >= kRunsPerWorkgroupForDynamic * run * G;
  __shared__ unsigned s_run;
  unsigned fetched = 0u;
  if (by_runs && threadIdx.x == 0) fetched = __hip_atomic_fetch_add(next_run, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (bool first_pass = true;; first_pass = false) {
  int item, item_end;
  if (by_runs) {
    if (threadIdx.x == 0) s_run = fetched;
    __syncthreads();
    item = (int)s_run;
    if ((unsigned)item >= T) break;
    if (threadIdx.x == 0) fetched = __hip_atomic_fetch_add(next_run, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    item_end = (int)min((unsigned)item + run, T);
  } else {
    // equal contiguous ranges; with fewer chunks than workgroups, one chunk each for the FIRST workgroups (they are dispatched
    // to different CUs; spreading b * T / G would put two working groups on some CUs and none on others)
    if (!first_pass) break;
    item = T < G ? (int)b : (int)(b * T / G);
    item_end = T < G ? (b < T ? (int)b + 1 : (int)b) : (int)((b + 1) * T / G);
    if (item >= item_end) break;
  }
#else
  int item = T < G ? (int)b : (int)(b * T / G);
  const int item_end = T < G ? (b < T ? (int)b + 1 : (int)b) : (int)((b + 1) * T / G);
  if (item >= item_end) return;
#endif

#ifdef SICP_DEV_PROBES
  // developer build only (-DSICP_DEV_PROBES, tools/corun_probe.py): hdr->pad_[0] > 1 repeats the workgroup's whole range
  // that many times inside ONE launch -- the same sums every time -- which keeps the persistent workgroups resident the
  // way a fused multi-evaluation kernel would, so that what co-resides with them can be measured.  The product kernel
  // has neither the loop nor the header read.
  const int item_first = item;
  for (int rep = uniform_i32(hdr->pad_[0] > 1 ? hdr->pad_[0] : 1); rep > 0; --rep) {
  item = item_first;
#endif
  while (item < item_end) {
    // the pair this item belongs to: the last one that begins at or before it (always a running one)
    int lo = 0, hi = n_pairs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (ctx[mid].item_begin <= item) lo = mid; else hi = mid - 1;
    }
    const int pair = uniform_i32(lo);
    const int chunk_lo = uniform_i32(item - ctx[pair].item_begin);
    // the pair's constants: uniform addresses, i.e. scalar loads straight into SGPRs
    const AccArgs& a = batch[pair].a;
    LoadCtx L;
    L.idx = (const SICP_GLOBAL int*)uniform_ptr(a.idx);
    L.w = (const SICP_GLOBAL double*)uniform_ptr(a.w);
    L.srec = (const SICP_GLOBAL PointRec*)uniform_ptr(a.srec);
    L.trec = (const SICP_GLOBAL PointRec*)uniform_ptr(a.trec);
    L.sdense = (const SICP_GLOBAL char*)uniform_ptr(a.srec_dense);
    L.n_s = uniform_i32(a.n_s);
    L.total = uniform_i32(a.n_s * a.K);
    const AccGeometry geo = acc_geometry(L.total, SG);
    const int n_chunks = uniform_i32(geo.n_chunks), steps = uniform_i32(geo.steps);
    const int n_here = min(n_chunks - chunk_lo, item_end - item);
    SICP_GLOBAL double* partials = (SICP_GLOBAL double*)uniform_ptr(a.partials);
    MathCtx M;
    {
      Pose P;
#if defined(SICP_LM_STEP_IN_LAUNCH)
      if (a.ein) {
        const double* q = a.ein[epoch & 1u].pose;
        se3::rotation(q, P.R);
        P.t[0] = q[4]; P.t[1] = q[5]; P.t[2] = q[6];
      } else
#endif
      if (a.lm) {
        se3::rotation(a.lm->pose, P.R);
        P.t[0] = a.lm->pose[4]; P.t[1] = a.lm->pose[5]; P.t[2] = a.lm->pose[6];
      } else {
        P = a.pose;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) M.P.R[k] = uniform_f64(P.R[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) M.P.t[k] = uniform_f64(P.t[k]);
      const double loss_b = a.loss.cauchy_a * a.loss.cauchy_a;
      M.one_m_eps = uniform_f64(a.one_m_eps); M.loss_b = uniform_f64(loss_b); M.loss_c = uniform_f64(1.0 / loss_b);
      M.gw = uniform_f64(2.0 / a.one_m_eps - 1.0);
      M.log_table = (unsigned)uniform_i32((int)(unsigned)(unsigned long)(SICP_LDS char*)log_tab);
    }

    accumulate_segment<K, SQLOSS, BS>(L, M, chunk_lo, n_here, n_chunks, steps, uniform_i32(geo.chunk_groups), partials, stage, tile, comb, lane, wave);
    item += n_here;
#if defined(SICP_LM_STEP_IN_LAUNCH)
    // ---- the LM step, inside this launch: the workgroup that delivers a pair's LAST columns sums them all and advances the
    // pair's trust-region machine (csrc/lm.hpp: the code lm_step_batch_kernel runs, in the order it runs it: the same bits)
    // while the other workgroups are still evaluating other pairs -- no second launch per evaluation, and nothing waits for
    // a 312-register block to find a SIMD between the search waves that share the chip with a tick.
    // Hand-off (cdna_hip_programming.md recipe R1): the columns were stored write-through, every wave drains its stores,
    // ONE agent-scope add on the pair's arrival counter (LmCore::pending) tells who is last; the last one reads the
    // columns with agent-scope loads.  The stepped state goes to the pair's LmState (what the host reads back after the tick)
    // and, as pose + status + epoch, to the OTHER EvalIn entry: the next launch's input, never this one's.
    EvalIn* const ein = uniform_ptr(a.ein);
    if (ein != nullptr) {
      LmState* const lm = uniform_ptr(a.lm_step);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (threadIdx.x == 0) {
        const int before = __hip_atomic_fetch_add(&lm->pending, n_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before + n_here == n_chunks;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s_last) {
        double o[28];
        reduce_partials_block<true>((const double*)partials, n_chunks, *reinterpret_cast<double (*)[4][28]>(&comb[0][0][0]), o);
        if (threadIdx.x == 0) lm_step_in_launch(lm, ein + ((epoch + 1u) & 1u), epoch + 1u, o);
        __syncthreads();  // (comb[] is the next segment's again)
      }
    }
#endif
  }
#ifdef SICP_DEV_PROBES
  }
#endif
#if !defined(SICP_ACC_STATIC_RANGES) && !defined(SICP_DEV_PROBES)
  __syncthreads();  // (s_run is rewritten at the top)
  }
  if (by_runs && threadIdx.x == 0) {
    const unsigned gone = __hip_atomic_fetch_add(left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone + 1u == G) {  // the last one out: the next launch finds the counters at zero
      __hip_atomic_store(next_run, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}

// device-resident solve: reduce the partial columns and advance the LM machine by one evaluation
// (lm.hpp: the same lm_feed the host loop runs).  One workgroup per pair; the grid is the capacity of
// the batch buffers, the blocks beyond the number of active pairs leave at once.  Thread 0 takes the
// trust-region step (serial: ~10 us, the longest link of a pair-alone evaluation).
#ifndef SICP_LM_STEP_WAVES
#define SICP_LM_STEP_WAVES 1  // waves per SIMD the LM step is compiled for (1: lm_feed gets the 312 VGPRs it wants)
#endif
__global__ __launch_bounds__(REDUCE_THREADS, SICP_LM_STEP_WAVES) void lm_step_batch_kernel(const BatchHeader* __restrict__ hdr, const BatchArgs* __restrict__ batch) {
  if ((int)blockIdx.x >= hdr->n_pairs) return;
  const BatchArgs& B = batch[blockIdx.x];
  LmState* lm = B.a.lm_step;
  // One pair alone is a chain of ~120 (accumulate, LM step) launches, so every dependent memory round trip
  // in here is paid ~120 times: the state (thread 0) and the status are requested BEFORE the reduction's own
  // loads and arrive while it runs, instead of status -> partial columns -> state one after the other.
  // The machine is stepped by the WHOLE first wave, every lane on its own copy of the same state (the same instructions as one
  // lane): the independent pieces of a step with one instruction sequence -- the six sqrt(diag / radius), the two sincos of a
  // Plus -- then run in different lanes at once (lm.hpp / se3.hpp, WAVE: ~320 of ~1500 instructions).
  LmCore st;
  if (threadIdx.x < 64) st = *lm;  // the options stay in memory: uniform, read with scalar loads
  const int status = lm->status;
  __shared__ double s_part[4][28];
  double o[28];
  reduce_partials_block(B.a.partials, B.nb, s_part, o);
  if (status != LM_RUNNING) return;  // (uniform; a finished pair's partial columns are read for nothing: 44 KB)
  if (threadIdx.x < 64) {
    const int finite = sums_all_finite(s_part);
#pragma unroll
    for (int k = 0; k < 28; ++k) o[k] = s_part[0][k];
    lm_feed<true>(st, lm->opt, o, finite);
    if (threadIdx.x == 0) *static_cast<LmCore*>(lm) = st;
  }
}

// ------------------------------------------------------------------------------------------
// One pair alone: the whole inner solve in ONE launch.
//
// A pair alone is a chain of ~120 evaluations, and as [accumulate launch, LM-step launch] each costs two launch
// boundaries (a no-op launch of either kernel is 4-4.5 us) on top of its work, and every evaluation streams the
// same 5.7 MB again.  Here the grid is one WORKER workgroup per chunk plus one MASTER, all resident (one per CU):
//   * worker b loads chunk b ONCE -- indices, weights, source records, the gathered target records: 8 slots per
//     lane, ~114 registers -- and keeps it in registers for the whole solve: the correspondences do not change
//     inside an inner solve, only the pose does.  An evaluation is compute_group over the resident groups (the
//     batched kernel's arithmetic, the same order: the same bits) -> wave_reduce -> column b -> flag b := e + 1;
//   * the master polls the flags (one lane per worker: parallel loads, no contended atomic), sums the columns
//     (reduce_partials_block: the LM-step kernel's routine), advances the trust-region state (lm_feed, state in
//     the registers of its thread 0), publishes pose + status and raises the epoch;
//   * the workers poll the epoch, read the pose, go again.
// Two hand-offs per evaluation, without a fence (cdna_hip_programming.md guideline 16, recipe R1): the payload --
// a column's 28 doubles, the pose -- is stored write-through (agent-scope 8-byte stores), the storing wave drains
// (s_waitcnt vmcnt(0)), ONE lane stores the flag; the consumer polls the flag relaxed and reads the payload with
// agent-scope loads, which neither its L1 nor its XCD's L2 can serve stale.  (With plain stores + a release fence
// and an acquire fence + plain loads each hand-off cost ~4 us more: buffer_wbl2 / buffer_inv are ~1.7 us each.)
// Columns, pose and flags need no double buffering: a worker writes column e + 1 only after epoch e + 1, which the master raises after it has summed the columns of e; the master writes
// pose e + 2 only after every worker has flagged e + 1, i.e. has read pose e + 1.  Every spin is bounded: on a
// timeout the abort word is raised, every workgroup leaves, and the host reports an error instead of waiting
// for a grid that is not resident.
// ------------------------------------------------------------------------------------------
// sync[0]: abort word (raised = the launch's first tag); sync[2 .. 31]: the published pose as 15 granules {tag = evaluations fed, 32 bits of payload}
// (7 doubles in halves + the status: the data is its own flag -- recipe R2 -- so a worker learns the new pose in ONE
// round trip); sync[kSoloFlags + b]: worker b's flag (= evaluations it has delivered).
constexpr int kSoloFlags = 32, kSoloGranules = 15;
static_assert(kSoloFlags + 256 <= kSoloSyncWords, "flags of up to 256 workers");

template <int K, bool SQLOSS, int BS>
__global__ __launch_bounds__(BS, 1) void solve_one_kernel(const SoloArgs A) {
  constexpr int SG = GroupShape<K>::SG, NS = GroupShape<K>::NS, NW = BS / 64, STEPS = 8 / SG;
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [reduction tiles | logarithm table]
  __shared__ double comb[NW][28];
  __shared__ double s_part[4][28];
  __shared__ unsigned s_gran[kSoloGranules + 1];
  __shared__ int s_status, s_abort;
  char* log_tab = reinterpret_cast<char*>(smem + NW * RED_ROWS * RED_STRIDE);
  const int lane = threadIdx.x & 63, wave = uniform_i32((int)(threadIdx.x >> 6));
  const AccArgs& a = A.a;
  const int max_evals = A.max_evals, wait_ticks = A.wait_ticks;
  const unsigned tag_base = A.tag_base;
  // The abort word is raised to a value only THIS launch uses (its first hand-off tag, never 0): a worker that gives
  // up after the master has already ended regularly leaves a word behind that no later launch mistakes for its own.
  const unsigned abort_tag = tag_base + 1u;
  unsigned* const sync = A.sync;
  LoadCtx L;
  L.idx = (const SICP_GLOBAL int*)uniform_ptr(a.idx);
  L.w = (const SICP_GLOBAL double*)uniform_ptr(a.w);
  L.srec = (const SICP_GLOBAL PointRec*)uniform_ptr(a.srec);
  L.trec = (const SICP_GLOBAL PointRec*)uniform_ptr(a.trec);
  L.sdense = nullptr;  // (a worker loads its chunk once)
  L.n_s = uniform_i32(a.n_s);
  L.total = uniform_i32(a.n_s * a.K);
  const AccGeometry geo = acc_geometry(L.total, SG);
  const int n_chunks = uniform_i32(geo.n_chunks), chunk_groups = uniform_i32(geo.chunk_groups);
  // (the host launches one workgroup per chunk + the master, and only for pairs whose chunks are 8 slots per lane)
  if (n_chunks + 1 != (int)gridDim.x || geo.steps != STEPS || n_chunks > BS) return;
  SICP_GLOBAL double* col = (SICP_GLOBAL double*)uniform_ptr(a.partials);
  LmState* lm = a.lm_step;
  SICP_GLOBAL unsigned* const abort_word = (SICP_GLOBAL unsigned*)sync;
  SICP_GLOBAL u64* const gran = (SICP_GLOBAL u64*)(abort_word + 2);
  SICP_GLOBAL unsigned* const flags = abort_word + kSoloFlags;
#if defined(SICP_SOLO_TIMING)  // developer aid: cycles per phase of the master and of worker 0, summed over the evaluations
  unsigned long long tm[4] = {0, 0, 0, 0}, tprev = 0;
#define SOLO_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm[i] += now_ - tprev; tprev = now_; } while (0)
#else
#define SOLO_MARK(i) do { } while (0)
#endif
  if (threadIdx.x == 0) s_abort = 0;

  if (blockIdx.x == 0) {
    // ------------------------------------------------------------------ the master
    LmCore st;  // (a copy in every lane of the first wave: lm_feed<true>)
    if (threadIdx.x < 64) {
      if (A.init) {
        LmState fresh;
        lm_init(fresh, A.opt, A.start);
        st = fresh;
      } else {
        st = *lm;
      }
      if (threadIdx.x == 0) s_status = st.status;
    }
    __syncthreads();
    for (int e = 0; e < max_evals; ++e) {
      if (s_status != LM_RUNNING) break;
      const unsigned want = tag_base + (unsigned)(e + 1);
#if defined(SICP_SOLO_TIMING)
      tprev = __builtin_readcyclecounter();
#endif
      if ((int)threadIdx.x < n_chunks) {
        unsigned long long t0 = 0;
        for (int spins = 0; __hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want; ++spins) {
          __builtin_amdgcn_s_sleep(1);
          if ((spins & 15) != 15 && wait_ticks > 0) continue;  // the clock and the abort word every 16th poll
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          if (t0 == 0) t0 = now;
          if (now - t0 >= (unsigned long long)wait_ticks || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_tag) {
            __hip_atomic_store(abort_word, abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_abort = 1;
            break;
          }
        }
      }
      __syncthreads();
      if (s_abort) return;
      SOLO_MARK(0);
      double o[28];
      reduce_partials_block<true>((const double*)col, n_chunks, s_part, o);
      SOLO_MARK(1);
      if (threadIdx.x < 64) {
        // the first wave steps the machine (every lane the same state: lm_step_batch_kernel); the new pose leaves as ONE store
        // instruction of 15 lanes (15 stores of one lane reach the other XCDs one after the other: ~1.5 us)
        const int finite = sums_all_finite(s_part);
#pragma unroll
        for (int k = 0; k < 28; ++k) o[k] = s_part[0][k];
        lm_feed<true>(st, A.opt, o, finite);
        if (threadIdx.x == 0) {
          SOLO_MARK(2);
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            const u64 bits = (u64)__double_as_longlong(st.pose[k]);
            s_gran[2 * k] = (unsigned)bits;
            s_gran[2 * k + 1] = (unsigned)(bits >> 32);
          }
          s_gran[14] = (unsigned)st.status;
          s_status = st.status;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < kSoloGranules)
          __hip_atomic_store(gran + lane, ((u64)want << 32) | (u64)s_gran[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SOLO_MARK(3);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      st.pad_ = A.seq;  // the launch ran to its regular end (after a timed-out wait the state in HBM is left as it was)
      *static_cast<LmCore*>(lm) = st;
      if (A.init) lm->opt = A.opt;
      if (A.host_state) {  // the host's copy, then the word it polls (system scope: the CPU sees the state before the flag)
        *A.host_state = st;
        __threadfence_system();
        __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#if defined(SICP_SOLO_TIMING)
      for (int i = 0; i < 4; ++i) { sync[kSoloSyncWords + 2 * i] = (unsigned)tm[i]; sync[kSoloSyncWords + 1 + 2 * i] = (unsigned)(tm[i] >> 32); }
#endif
    }
    return;
  }

  // -------------------------------------------------------------------- a worker
  const int b = (int)blockIdx.x - 1;
  for (int k = threadIdx.x; k < kLogTableEntries; k += BS) reinterpret_cast<v2d*>(log_tab)[k] = reinterpret_cast<const v2d*>(kLogTable)[k];
  MathCtx M;
  {
    const double loss_b = a.loss.cauchy_a * a.loss.cauchy_a;
    M.one_m_eps = uniform_f64(a.one_m_eps); M.loss_b = uniform_f64(loss_b); M.loss_c = uniform_f64(1.0 / loss_b);
    M.gw = uniform_f64(2.0 / a.one_m_eps - 1.0);
    M.log_table = (unsigned)uniform_i32((int)(unsigned)(unsigned long)(SICP_LDS char*)log_tab);
  }
  SICP_LDS double* tile = (SICP_LDS double*)smem + wave * (RED_ROWS * RED_STRIDE);
  // the chunk, once: what accumulate_segment loads step by step (same clamps, same dead-slot rule)
  Group<K> G[STEPS];
  {
    const int last = ((L.total - 1) / SG) * SG;
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      const int g = b * chunk_groups + (int)threadIdx.x + t * BS;
      load_idx_raw<K>(L, g, G[t].j);
      fix_idx<K>(L.total, g, G[t].j);
      GroupRegs<K> R;
      load_regs<K>(L, last, g, R);
#pragma unroll
      for (int c = 0; c < SG; ++c) {
        G[t].w[c] = R.w[c];
        load_rec(L.trec + max(G[t].j[c], 0), G[t].tx[c], G[t].ty[c], G[t].tz[c], G[t].tnx[c], G[t].tny[c], G[t].tnz[c]);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) { G[t].sx[s] = R.sx[s]; G[t].sy[s] = R.sy[s]; G[t].sz[s] = R.sz[s]; G[t].snx[s] = R.snx[s]; G[t].sny[s] = R.sny[s]; G[t].snz[s] = R.snz[s]; }
    }
  }
  if (threadIdx.x < 7) {
    const u64 bits = (u64)__double_as_longlong(A.init ? A.start[threadIdx.x] : lm->pose[threadIdx.x]);
    s_gran[2 * threadIdx.x] = (unsigned)bits;
    s_gran[2 * threadIdx.x + 1] = (unsigned)(bits >> 32);
  }
  if (threadIdx.x == 0) s_gran[14] = (unsigned)(A.init ? (int)LM_RUNNING : lm->status);
  __syncthreads();
  for (int e = 0; e < max_evals; ++e) {
    if ((int)s_gran[14] != LM_RUNNING) break;
#if defined(SICP_SOLO_TIMING)
    tprev = __builtin_readcyclecounter();
#endif
    {
      double pose[7], R[9];
#pragma unroll
      for (int k = 0; k < 7; ++k) pose[k] = __hiloint2double((int)s_gran[2 * k + 1], (int)s_gran[2 * k]);
      se3::rotation(pose, R);
#pragma unroll
      for (int k = 0; k < 9; ++k) M.P.R[k] = uniform_f64(R[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) M.P.t[k] = uniform_f64(pose[4 + k]);
    }
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < STEPS; ++t) compute_group<K, SQLOSS>(M, G[t], acc);
    wave_reduce(acc, tile, (SICP_LDS double*)&comb[wave][0], lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 28)
      coherent_store_f64(col + (size_t)threadIdx.x * n_chunks + b, (comb[0][threadIdx.x] + comb[1][threadIdx.x]) + (comb[2][threadIdx.x] + comb[3][threadIdx.x]));
    SOLO_MARK(0);
    // ---- publish the column (its 28 write-through stores are wave 0's: that wave drains, one lane raises the flag),
    //      then wave 0 sweeps the pose granules until all 15 carry this evaluation's tag
    // (every thread read this evaluation's pose out of s_gran before the barrier above)
    if (threadIdx.x < 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned want = tag_base + (unsigned)(e + 1);
      if (threadIdx.x == 0) __hip_atomic_store(&flags[b], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SOLO_MARK(1);
      const int gl = min(lane, kSoloGranules - 1);
      unsigned long long t0 = 0;
      for (int spins = 0;; ++spins) {
        const u64 x = __hip_atomic_load(gran + gl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(x >> 32) == want)) {
          if (lane < kSoloGranules) s_gran[lane] = (unsigned)x;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 15) != 15 && wait_ticks > 0) continue;  // the clock and the abort word every 16th poll
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) t0 = now;
        if (now - t0 >= (unsigned long long)wait_ticks || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == abort_tag) {
          if (lane == 0) {
            __hip_atomic_store(abort_word, abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_abort = 1;
          }
          break;
        }
      }
    }
    __syncthreads();
    if (s_abort) return;
    SOLO_MARK(2);
  }
#if defined(SICP_SOLO_TIMING)
  if (b == 0 && threadIdx.x == 0)
    for (int i = 0; i < 3; ++i) { sync[kSoloSyncWords + 8 + 2 * i] = (unsigned)tm[i]; sync[kSoloSyncWords + 9 + 2 * i] = (unsigned)(tm[i] >> 32); }
#endif
}

// ------------------------------------------------------------------------------------------
// The evaluation on FULL covariance matrices: a pair whose covariances are the caller's own and not of the form
// I - (1-eps) n n^T (sicp_set_covariances; impl/semantic_icp.hpp:73,77 registers with whatever sits in labeledCovariances).
// gicp_cost_function.h:27-73 as written there -- A = C_t + R C_s R^T, its inverse by cofactors, r = res^T A^-1 res -- chained
// with LocalParameterizationSE3 in the closed form that holds for SYMMETRIC covariances:
// a = A^-1 res, b = R^T a, c = p_s + C_s b, J = [-2 b ; 2 (b x c)].  Losses, Corrector and the 28 sums as in compute_group.
// One pair at a time, a plain lane-per-slot loop (no staging, no pipelining): the path of an exotic input, not of the metric.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_cov6(const double* __restrict__ cov6, const PointRec& r, int i, double one_m_eps, double (&C)[6]) {
  if (cov6) {
#pragma unroll
    for (int k = 0; k < 6; ++k) C[k] = cov6[6 * (size_t)i + k];
  } else {  // I - (1 - eps) n n^T
    C[0] = 1.0 - one_m_eps * r.nx * r.nx; C[1] = -one_m_eps * r.nx * r.ny; C[2] = -one_m_eps * r.nx * r.nz;
    C[3] = 1.0 - one_m_eps * r.ny * r.ny; C[4] = -one_m_eps * r.ny * r.nz; C[5] = 1.0 - one_m_eps * r.nz * r.nz;
  }
}

template <bool SQLOSS>
__global__ __launch_bounds__(256) void accumulate_general_kernel(GenAccArgs G) {
  __shared__ double s_w[4][28];
  const AccArgs& a = G.a;
  const int total = a.n_s * a.K, chunk = blockIdx.x;
  const int per = (total + G.n_chunks - 1) / G.n_chunks;
  const int s0 = chunk * per, s1 = min(total, s0 + per);
  const double* R = a.pose.R;
  const double loss_b = a.loss.cauchy_a * a.loss.cauchy_a, loss_c = 1.0 / loss_b;
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  for (int s = s0 + (int)threadIdx.x; s < s1; s += 256) {
    const int j = a.idx[s];
    if (j < 0) continue;
    const int i = s / a.K;
    const double w = a.w ? a.w[s] : 1.0;
    const PointRec sr = a.srec[i], tr = a.trec[j];
    double Cs[6], Ct[6];
    load_cov6(G.scov6, sr, i, a.one_m_eps, Cs);
    load_cov6(G.tcov6, tr, j, a.one_m_eps, Ct);
    const double ps[3] = {(double)sr.x, (double)sr.y, (double)sr.z};
    const double res[3] = {(double)tr.x - (R[0] * ps[0] + R[1] * ps[1] + R[2] * ps[2] + a.pose.t[0]),
                           (double)tr.y - (R[3] * ps[0] + R[4] * ps[1] + R[5] * ps[2] + a.pose.t[1]),
                           (double)tr.z - (R[6] * ps[0] + R[7] * ps[1] + R[8] * ps[2] + a.pose.t[2])};
    // T = R C_s (C_s symmetric: xx xy xz yy yz zz), A = C_t + T R^T
    const double S[3][3] = {{Cs[0], Cs[1], Cs[2]}, {Cs[1], Cs[3], Cs[4]}, {Cs[2], Cs[4], Cs[5]}};
    double T[3][3], A[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[r][c] = R[3 * r] * S[0][c] + R[3 * r + 1] * S[1][c] + R[3 * r + 2] * S[2][c];
    const double Tm[3][3] = {{Ct[0], Ct[1], Ct[2]}, {Ct[1], Ct[3], Ct[4]}, {Ct[2], Ct[4], Ct[5]}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) A[r][c] = Tm[r][c] + (T[r][0] * R[3 * c] + T[r][1] * R[3 * c + 1] + T[r][2] * R[3 * c + 2]);
    // inverse by cofactors (Eigen's Matrix3d::inverse())
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02, inv = 1.0 / det;
    const double I00 = c00 * inv, I01 = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * inv, I02 = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * inv;
    const double I10 = c01 * inv, I11 = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * inv, I12 = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * inv;
    const double I20 = c02 * inv, I21 = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * inv, I22 = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * inv;
    const double av[3] = {I00 * res[0] + I01 * res[1] + I02 * res[2], I10 * res[0] + I11 * res[1] + I12 * res[2],
                          I20 * res[0] + I21 * res[1] + I22 * res[2]};
    const double r = res[0] * av[0] + res[1] * av[1] + res[2] * av[2];
    const double b[3] = {R[0] * av[0] + R[3] * av[1] + R[6] * av[2], R[1] * av[0] + R[4] * av[1] + R[7] * av[2],
                         R[2] * av[0] + R[5] * av[1] + R[8] * av[2]};
    const double cv[3] = {ps[0] + (S[0][0] * b[0] + S[0][1] * b[1] + S[0][2] * b[2]), ps[1] + (S[1][0] * b[0] + S[1][1] * b[1] + S[1][2] * b[2]),
                          ps[2] + (S[2][0] * b[0] + S[2][1] * b[1] + S[2][2] * b[2])};
    const double J[6] = {-2.0 * b[0], -2.0 * b[1], -2.0 * b[2], 2.0 * (b[1] * cv[2] - b[2] * cv[1]), 2.0 * (b[2] * cv[0] - b[0] * cv[2]),
                         2.0 * (b[0] * cv[1] - b[1] * cv[0])};
    double sum, rho1;
    loss_rho1<SQLOSS>(loss_c, r * r, w, sum, rho1);
    if (!SQLOSS) rho1 *= w;
    int o = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const double jp = rho1 * J[p];
#pragma unroll
      for (int q = p; q < 6; ++q, ++o) acc[o] = FMA(jp, J[q], acc[o]);
      acc[21 + p] = FMA(jp, r, acc[21 + p]);
    }
    const unsigned off = log_entry_offset(sum) >> 3;  // (doubles)
    acc[27] = FMA(0.5 * w, loss_b * log_from_entry(sum, kLogTable[off], kLogTable[off + 1]), acc[27]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_w[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 28) a.partials[(size_t)threadIdx.x * G.n_chunks + chunk] = (s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + (s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
}

hipError_t launch_accumulate_general(const GenAccArgs& g, hipStream_t st) {
  if (g.n_chunks <= 0) return hipSuccess;
  if (g.a.loss.use_sqloss) hipLaunchKernelGGL(accumulate_general_kernel<true>, dim3(g.n_chunks), dim3(256), 0, st, g);
  else hipLaunchKernelGGL(accumulate_general_kernel<false>, dim3(g.n_chunks), dim3(256), 0, st, g);
  return hipGetLastError();
}

__global__ __launch_bounds__(REDUCE_THREADS) void finalize_batch_kernel(const BatchArgs* __restrict__ batch, double* out28) {
  const BatchArgs& B = batch[blockIdx.x];
  __shared__ double s_part[4][28];
  double o[28];
  reduce_partials_block(B.a.partials, B.nb, s_part, o);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 28; ++k) out28[28 * blockIdx.x + k] = o[k];
  }
}

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

// test hook: the trust-region machine on the device, fed with a GIVEN sequence of evaluations (lm_feed is a pure function of
// state and evaluation, so any sequence will do: indefinite H, rejected steps, non-finite sums).  One wave per item; WAVE = the
// form the kernels run (lm_feed<true>: every lane its own copy of the state, the finite test by ballot), otherwise the one-lane
// form the host runs.  out = kLmSeqOut doubles of the final state.
template <bool WAVE>
__global__ __launch_bounds__(64, 1) void lm_feed_sequence_kernel(int n, const double* __restrict__ in, double* __restrict__ out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n || (!WAVE && lane != 0)) return;
  const double* item = in + (size_t)i * kLmSeqIn;
  LmState s;
  LmOptions opt;
  double x0[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) x0[k] = item[k];
  lm_init(s, opt, x0);
  for (int e = 0; e < kLmSeqEvals && s.status == LM_RUNNING; ++e) {
    double o[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) o[k] = item[7 + 28 * e + k];
    if (WAVE) {
      const double v = item[7 + 28 * e + min(lane, 27)];
      lm_feed<true>(s, opt, o, __all(v - v == 0.0) ? 1 : 0);
    } else {
      lm_feed<false>(s, opt, o);
    }
  }
  if (lane == 0) {
    double* r = out + (size_t)i * kLmSeqOut;
    int k = 0;
    for (int j = 0; j < 7; ++j) r[k++] = s.pose[j];
    for (int j = 0; j < 7; ++j) r[k++] = s.x[j];
    for (int j = 0; j < 6; ++j) r[k++] = s.diag[j];
    for (int j = 0; j < 6; ++j) r[k++] = s.scale[j];
    r[k++] = s.radius; r[k++] = s.cost; r[k++] = s.model_change; r[k++] = s.decrease_factor; r[k++] = s.x_norm;
    r[k++] = s.status; r[k++] = s.iterations; r[k++] = s.evaluations; r[k++] = s.invalid; r[k++] = s.reuse_diagonal; r[k++] = s.phase;
  }
}

hipError_t launch_se3_ops(int op, int n, const double* in, double* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (op == 5) hipLaunchKernelGGL(lm_feed_sequence_kernel<true>, dim3(n), dim3(64), 0, st, n, in, out);
  else if (op == 6) hipLaunchKernelGGL(lm_feed_sequence_kernel<false>, dim3(n), dim3(64), 0, st, n, in, out);
  else hipLaunchKernelGGL(se3_ops_kernel, dim3((n + 63) / 64), dim3(64), 0, st, op, n, in, out);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
// columns of partials[28][.]: one per chunk
int accumulate_blocks(int total, int K) { return acc_geometry(total, acc_slots_per_group(K)).n_chunks; }

// ---- the batched evaluation --------------------------------------------------------------------
static void* accumulate_fn(int K, int use_sqloss) {
  switch (K) {
    case 1: return use_sqloss ? (void*)accumulate_staged_kernel<1, true, 256> : (void*)accumulate_staged_kernel<1, false, 256>;
    case 4: return use_sqloss ? (void*)accumulate_staged_kernel<4, true, 256> : (void*)accumulate_staged_kernel<4, false, 256>;
    case 20: return use_sqloss ? (void*)accumulate_staged_kernel<20, true, 256> : (void*)accumulate_staged_kernel<20, false, 256>;
    default: return nullptr;
  }
}

int accumulate_grid() {
  // persistent workgroups: two per CU (two waves per SIMD).  SICP_ACC_GRID is a tuning aid.
  static const int grid = [] {
    const char* e = getenv("SICP_ACC_GRID");
    if (e && atoi(e) > 0) return atoi(e);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return SICP_ACC_OCC * (cus > 0 ? cus : 256);
  }();
  return grid;
}

static size_t stream_smem_bytes(int capacity, int K) {
  const size_t staging = (size_t)4 * acc_slots_per_group(K) * STAGE_SLOT_BYTES;
  return sizeof(double) * 4 * RED_ROWS * RED_STRIDE + staging + LOG_TABLE_BYTES + sizeof(PairSlot) * (size_t)capacity;
}

hipError_t launch_accumulate_batch(int K, int use_sqloss, const BatchHeader* hdr, const BatchArgs* batch, int capacity, hipStream_t st, int node) {
  void* fn = accumulate_fn(K, use_sqloss);
  if (!fn) return hipErrorInvalidValue;
  void* args[] = {(void*)&hdr, (void*)&batch, (void*)&node};
  return hipLaunchKernel(fn, dim3(accumulate_grid()), dim3(256), args, stream_smem_bytes(capacity, K), st);
}

static size_t solo_smem_bytes(int) { return sizeof(double) * 4 * RED_ROWS * RED_STRIDE + LOG_TABLE_BYTES; }

// whether a pair of `total` slots can run as one persistent solve: one workgroup per chunk and the master, all of
// them resident (one per CU), a chunk being 8 slots per lane (m = 1 in acc_geometry: always, at this size)
bool solve_one_fits(int total, int K) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static const bool off = getenv("SICP_NO_SOLO") != nullptr;  // developer switch: always the [accumulate, LM step] graph
  return !off && total > 0 && accumulate_blocks(total, K) + 1 <= (cus < 257 ? cus : 257) && acc_geometry(total, acc_slots_per_group(K)).steps == 8 / acc_slots_per_group(K);
}

// How long a wait of the persistent solve may last before it gives up, in ticks of the 100 MHz constant clock: 5 ms
// (a hand-off normally takes microseconds; a grid that is not resident as a whole will not become so while its
// resident part spins).  SICP_SOLO_WAIT_TICKS: test aid (0 = the first unsuccessful poll gives up, which exercises the
// host's fallback to the ticks).
int solo_wait_ticks() {
  static const int v = [] { const char* e = getenv("SICP_SOLO_WAIT_TICKS"); return e ? atoi(e) : 500000; }();
  return v;
}

hipError_t launch_solve_one(int K, int use_sqloss, const SoloArgs& args, int n_chunks, hipStream_t st) {
  void* fn = nullptr;
  switch (K) {
    case 1: fn = use_sqloss ? (void*)solve_one_kernel<1, true, 256> : (void*)solve_one_kernel<1, false, 256>; break;
    case 4: fn = use_sqloss ? (void*)solve_one_kernel<4, true, 256> : (void*)solve_one_kernel<4, false, 256>; break;
    case 20: fn = use_sqloss ? (void*)solve_one_kernel<20, true, 256> : (void*)solve_one_kernel<20, false, 256>; break;
    default: return hipErrorInvalidValue;
  }
  void* kargs[] = {(void*)&args};
  return hipLaunchKernel(fn, dim3(n_chunks + 1), dim3(256), kargs, solo_smem_bytes(K), st);
}

hipError_t launch_lm_step_batch(const BatchHeader* hdr, const BatchArgs* batch, int capacity, hipStream_t st) {
  if (capacity <= 0) return hipSuccess;
  hipLaunchKernelGGL(lm_step_batch_kernel, dim3(capacity), dim3(REDUCE_THREADS), 0, st, hdr, batch);
  return hipGetLastError();
}

// First kernel of a tick whose accumulate launches step the LM machines themselves: a new epoch range for the tick, and every
// pair's pose / status -- from its LmState, whoever wrote it last: lm_init_kernel for a pair that joins, the previous tick's
// last step, a persistent solve that gave up -- as the input of the tick's first evaluation.  ONE workgroup (it owns the
// header word it advances).
__global__ __launch_bounds__(256) void tick_prepare_kernel(BatchHeader* __restrict__ hdr, const BatchArgs* __restrict__ batch) {
  __shared__ unsigned s_epoch;
  if (threadIdx.x == 0) {
    s_epoch = hdr->epoch_base + (unsigned)kMaxBatchLen;  // (an even stride: a tick's first evaluation always has the same parity)
    hdr->epoch_base = s_epoch;
  }
  __syncthreads();
  const unsigned e0 = s_epoch;
  for (int p = threadIdx.x; p < hdr->n_pairs; p += blockDim.x) {
    const AccArgs& a = batch[p].a;
    if (!a.ein || !a.lm_step) continue;
    EvalIn& in = a.ein[e0 & 1u];
#pragma unroll
    for (int k = 0; k < 7; ++k) in.pose[k] = a.lm_step->pose[k];
    in.status = a.lm_step->status;
    in.epoch = e0;
    a.lm_step->pending = 0;
  }
}

hipError_t launch_tick_prepare(BatchHeader* hdr, const BatchArgs* batch, hipStream_t st) {
  hipLaunchKernelGGL(tick_prepare_kernel, dim3(1), dim3(256), 0, st, hdr, batch);
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void lm_init_kernel(const LmJoin* __restrict__ joins, int n, LmState* __restrict__ states) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const LmJoin j = joins[k];
  LmState s;
  lm_init(s, j.opt, j.start);
  states[j.pair] = s;
}

hipError_t launch_lm_init(const LmJoin* joins, int n, LmState* states, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(lm_init_kernel, dim3((n + 63) / 64), dim3(64), 0, st, joins, n, states);
  return hipGetLastError();
}

hipError_t launch_finalize_batch(const BatchArgs* batch, int n, double* out28, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_batch_kernel, dim3(n), dim3(REDUCE_THREADS), 0, st, batch, out28);
  return hipGetLastError();
}

// [accumulate, lm_step_batch] x len as an instantiated graph with explicit kernel nodes.  Both
// grids are fixed (persistent workgroups; one step block per slot of the batch buffers) and the
// kernels read the number of active pairs / items from the header in HBM, so the graph only depends
// on the buffers' addresses: it is built once per batch context.
void batch_graph_destroy(BatchGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr; g.graph = nullptr; g.len = 0;
}

hipError_t batch_graph_prepare(BatchGraph& g, int K, int use_sqloss, const BatchHeader* hdr, const BatchArgs* batch, int capacity, int len,
                               int* built, int fold, int static_ranges) {
  *built = 0;
  void* fn = accumulate_fn(K, use_sqloss);
  if (capacity <= 0 || len < 1 || len > kMaxBatchLen || !fn) return hipErrorInvalidValue;
  if (g.exec && g.K == K && g.sqloss == use_sqloss && g.len == len && g.batch == batch && g.hdr == hdr && g.capacity == capacity && g.fold == fold &&
      g.static_ranges == static_ranges)
    return hipSuccess;
  batch_graph_destroy(g);
  int node = 0;
  void* args[] = {(void*)&hdr, (void*)&batch, (void*)&node};  // (kernel parameters are copied when a node is added)
  hipKernelNodeParams pa, ps;
  memset(&pa, 0, sizeof pa);
  pa.func = fn;
  pa.gridDim = dim3(accumulate_grid());
  pa.blockDim = dim3(256);
  pa.sharedMemBytes = (unsigned)stream_smem_bytes(capacity, K);
  pa.kernelParams = args;
  memset(&ps, 0, sizeof ps);
  ps.func = fold ? (void*)tick_prepare_kernel : (void*)lm_step_batch_kernel;
  ps.gridDim = dim3(fold ? 1 : capacity);
  ps.blockDim = dim3(fold ? 256 : REDUCE_THREADS);
  ps.kernelParams = args;
  hipError_t e = hipGraphCreate(&g.graph, 0);
  if (e != hipSuccess) return e;
  hipGraphNode_t prev = nullptr, acc = nullptr, step = nullptr;
  if (fold) {  // [tick_prepare, accumulate x len]: the accumulate launches step the machines themselves
    e = hipGraphAddKernelNode(&step, g.graph, nullptr, 0, &ps);
    if (e != hipSuccess) return e;
    prev = step;
  }
  for (int b = 0; b < len; ++b) {
    node = b | (static_ranges ? kAccStaticRanges : 0);
    e = hipGraphAddKernelNode(&acc, g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &pa);
    if (e != hipSuccess) return e;
    prev = acc;
    if (!fold) {
      e = hipGraphAddKernelNode(&step, g.graph, &acc, 1, &ps);
      if (e != hipSuccess) return e;
      prev = step;
    }
  }
  e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) return e;
  g.K = K; g.sqloss = use_sqloss; g.len = len; g.batch = batch; g.hdr = hdr; g.capacity = capacity; g.fold = fold; g.static_ranges = static_ranges;
  *built = 1;
  return hipSuccess;
}

}  // namespace sicp
