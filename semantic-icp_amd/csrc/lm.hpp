// lm.hpp -- 6-DoF Levenberg-Marquardt as a re-entrant state machine (product code, host + device).
//
// Replaces the inner ceres::Solve of the reference (em_icp.hpp:162-177, gicp.hpp:138-151,
// semantic_icp.hpp:136-149): one SE3 parameter block (7 ambient / 6 tangent), trust-region +
// Levenberg-Marquardt with Ceres' default step control (Ceres 1.14..2.1
// trust_region_minimizer.cc, levenberg_marquardt_strategy.cc), restated because Ceres is not
// vendored.  The GPU supplies, per evaluation, the 28 numbers a dense Jacobian would be reduced
// to anyway: H = J^T J (robustified), g = J^T r, cost.  DENSE_QR on [J; D] and Cholesky on
// H + D^2 solve the same 6x6 system.
//
// The machine consumes one evaluation at a time:
//     lm_init(s, opt, x0);                 // s.pose = x0 is the first point to evaluate
//     while (s.status == LM_RUNNING) { evaluate out28 at s.pose;  lm_feed(s, out28); }
// so the same code drives the host loop (sicp_api.cpp) and the device-resident solve, where
// lm_feed runs in a one-block kernel right after each accumulate kernel and the host only looks at
// s.status once per batch of launches (solve_kernels.hip: lm_step_batch_kernel).  On the GPU the step is taken by a whole
// wavefront (lm_feed<true>): every lane runs the machine on its own copy of the same state, and the independent pieces with
// one instruction sequence -- the six sqrt(diag / radius), the two sincos of se3::exp -- go to different lanes.  The same
// operations on the same values: the same bits as the one-lane form the host runs.
//
// Unlike Ceres, one evaluation returns cost, gradient and H together, so an accepted step does
// not need a second sweep at the same point (Ceres evaluates the candidate cost first and the
// Jacobian again after accepting); the iterates are the same.
#ifndef SICP_LM_HPP_
#define SICP_LM_HPP_

#include <math.h>
#include <string.h>

#include "se3.hpp"

// the 6x6 loops must unroll completely on the GPU, or their local arrays are indexed dynamically
// and land in scratch memory (the one-lane lm_feed then takes tens of microseconds)
#if defined(__HIP_DEVICE_COMPILE__)
#define SICP_UNROLL _Pragma("unroll")
#else
#define SICP_UNROLL
#endif

namespace sicp {

struct LmOptions {
  int max_iterations = 400;
  double gradient_tolerance = 1e-11;
  double function_tolerance = 1e-11;
  double parameter_tolerance = 1e-8;
  double initial_radius = 1e4;
  double max_radius = 1e16;
  double min_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  int max_consecutive_invalid_steps = 5;
  int jacobi_scaling = 1;
};

enum { LM_RUNNING = -1, LM_CONVERGED = 0, LM_ITERATION_CAP = 1, LM_INVALID_STEPS = 2, LM_EVAL_FAILED = 3 };

// the part of the state that changes (what the GPU step keeps in registers) ...
struct LmCore {
  double pose[7];  // the point whose evaluation lm_feed expects next
  int status;      // LM_* (next to the pose: the accumulate kernel reads both, one cache line)
  int pad_;
  double x[7];     // last accepted iterate (the answer when status != LM_RUNNING)
  double H[36], g[6], cost, x_norm;
  double scale[6], diag[6];
  double radius, decrease_factor, model_change;
  int reuse_diagonal, invalid, iterations, evaluations;
  int phase;   // 0: pose == x (first evaluation), 1: pose is a candidate step
  int pending; // chained device solve: an evaluation at `pose` has been accumulated but not fed yet
};

// ... and the whole state: the options stay in memory (uniform: scalar loads on the GPU)
struct LmState : LmCore {
  LmOptions opt;
};

namespace detail {

// Solves A y = b by Cholesky, IN PLACE: only the lower triangle of A is read, and it is overwritten
// by L (so the caller's 6x6 costs 21 live values, not 36 + 36: on the GPU this routine runs in one
// lane and its register footprint sets the allocation of the whole kernel).  The six reciprocals
// of the diagonal are formed once and multiplied (6 divisions instead of 27: a correctly rounded
// FP64 division is a 14-instruction sequence, and this lane runs alone).
SICP_HD inline bool chol6_solve(double* A, const double* b, double* y) {
  double inv[6];
  SICP_UNROLL
  for (int i = 0; i < 6; ++i)
    SICP_UNROLL
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j];
      SICP_UNROLL
      for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return false;
        A[6 * i + i] = sqrt(s);
        inv[i] = 1.0 / A[6 * i + i];
      } else {
        A[6 * i + j] = s * inv[j];
      }
    }
  double z[6];
  SICP_UNROLL
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    SICP_UNROLL
    for (int k = 0; k < i; ++k) s -= A[6 * i + k] * z[k];
    z[i] = s * inv[i];
  }
  SICP_UNROLL
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    SICP_UNROLL
    for (int k = i + 1; k < 6; ++k) s -= A[6 * k + i] * y[k];
    y[i] = s * inv[i];
  }
  return true;
}

SICP_HD inline void unpack28(const double* o, double* H, double* g, double* cost) {
  int k = 0;
  SICP_UNROLL
  for (int a = 0; a < 6; ++a)
    SICP_UNROLL
    for (int b = a; b < 6; ++b) { H[6 * a + b] = o[k]; H[6 * b + a] = o[k]; ++k; }
  SICP_UNROLL
  for (int a = 0; a < 6; ++a) g[a] = o[21 + a];
  *cost = o[27];
}

// Rigorous lower bounds of gradient_max_norm that cost far less than the SE(3) exp it contains.
// With u = g[0..2], w = g[3..5], t = |w|:
//  * the translation rows of x - Plus(x, -g) are R V(w) u; R is orthonormal (to rounding) and V is
//    normal with singular values 1 and 2|sin(t/2)|/t >= 2/pi for t <= pi, so
//    max-norm >= |V u|_2 / sqrt(3) >= 0.3676 |u|_2;
//  * the quaternion rows are q_x (1 - e) with e = exp's unit quaternion (angle t/2), and quaternion
//    norms multiply: |q_x (1 - e)|_2 = |q_x| 2 |sin(t/4)|, so max-norm >= |q_x| |sin(t/4)|.
// Each is used with a factor 2 of slack for rounding and |q_x| != 1.  When either is clearly above
// the tolerance -- every iteration but the last few of a solve -- the exact evaluation is skipped;
// the decisions are the same.
// The second bound without sin and sqrt (they were ~100 of the ~1500 instructions of a one-lane LM step): for
// x = t/4 in [0, pi/2], sin x >= 2 x / pi, so 0.5 sin(t/4) >= t / (4 pi) > tol whenever t^2 > (4 pi tol)^2 and
// t <= 2 pi.  A sufficient condition of a sufficient condition: where it fails the exact evaluation decides, as before.
SICP_HD inline bool gradient_clearly_above(const double* g, double tol) {
  const double u2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2], w2 = g[3] * g[3] + g[4] * g[4] + g[5] * g[5];
  if (w2 <= 9.0 && 0.18 * 0.18 * u2 > tol * tol) return true;
  const double four_pi_tol = 12.566370614359174 * tol;
  return w2 <= 39.0 && w2 > four_pi_tol * four_pi_tol;
}

// Ceres: ||x - Plus(x, -g)||_inf  (ambient coordinates)
SICP_HD inline double gradient_max_norm(const double* x, const double* g) {
  double ng[6], xp[7], m = 0;
  SICP_UNROLL
  for (int i = 0; i < 6; ++i) ng[i] = -g[i];
  se3::plus(x, ng, xp);
  SICP_UNROLL
  for (int i = 0; i < 7; ++i) m = fmax(m, fabs(x[i] - xp[i]));
  return m;
}

// From the accepted iterate: terminate, or compute the next trust-region step and publish the
// candidate in s.pose.  The retry loop (an invalid step halves the radius and tries again without a
// new evaluation) only contains the 6x6 solve: x and g do not change inside it, so the gradient
// test is done once, before it -- same decisions in the same order as testing it every time.
// WAVE (GPU only): called by a whole wavefront whose 64 lanes hold the same state (se3.hpp "WAVE variants").
template <bool WAVE = false>
SICP_HD inline void lm_propose(LmCore& s, const LmOptions& opt) {
  if (s.iterations >= opt.max_iterations) { s.status = LM_ITERATION_CAP; return; }
  if (!gradient_clearly_above(s.g, opt.gradient_tolerance) && gradient_max_norm(s.x, s.g) <= opt.gradient_tolerance) {
    s.status = LM_CONVERGED;
    return;
  }
  double step[6], model_change;
  for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
    // keeps the compiler from hoisting the loop-invariant half of the state into registers for
    // the (almost never taken) retry: that doubles the register allocation of the kernel
    asm volatile("" ::: "memory");
#endif
    if (s.radius <= opt.min_radius) { s.status = LM_CONVERGED; return; }
    s.iterations++;
    // scaled system Hs = S H S, gs = S g.  Hs is not kept: the lower triangle goes into A (which
    // the factorisation overwrites) and the model change below recomputes the entries it needs.
    double A[36], gs[6], y[6];
    SICP_UNROLL
    for (int a = 0; a < 6; ++a) {
      gs[a] = s.g[a] * s.scale[a];
      SICP_UNROLL
      for (int b = 0; b <= a; ++b) A[6 * a + b] = s.H[6 * a + b] * s.scale[a] * s.scale[b];
    }
    if (!s.reuse_diagonal)
      SICP_UNROLL
      for (int j = 0; j < 6; ++j) s.diag[j] = fmin(fmax(A[6 * j + j], opt.min_lm_diagonal), opt.max_lm_diagonal);
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE) {  // the six independent division + square root pairs: lane j takes diag[j] (~190 -> ~55 instructions)
      const int lane = se3::wave::lane_id();
      double dj = s.diag[0];
      SICP_UNROLL
      for (int j = 1; j < 6; ++j) dj = lane == j ? s.diag[j] : dj;
      const double lm = sqrt(dj / s.radius);
      const double lm2 = lm * lm;
      SICP_UNROLL
      for (int j = 0; j < 6; ++j) A[6 * j + j] += se3::wave::bcast(lm2, j);
    } else
#endif
    SICP_UNROLL
    for (int j = 0; j < 6; ++j) {
      const double lm = sqrt(s.diag[j] / s.radius);  // lm_diagonal_
      A[6 * j + j] += lm * lm;
    }
    s.reuse_diagonal = 1;
    const bool ok = chol6_solve(A, gs, y);
    model_change = 0;
    if (ok) {
      double sg = 0, sHs = 0;
      SICP_UNROLL
      for (int a = 0; a < 6; ++a) step[a] = -y[a];
      SICP_UNROLL
      for (int a = 0; a < 6; ++a) {
        sg += step[a] * gs[a];
        double r = 0;
        SICP_UNROLL
        for (int b = 0; b < 6; ++b) r += (s.H[6 * a + b] * s.scale[a] * s.scale[b]) * step[b];
        sHs += step[a] * r;
      }
      model_change = -(sg + 0.5 * sHs);
    }
    if (ok && model_change > 0.0) break;
    if (++s.invalid >= opt.max_consecutive_invalid_steps) { s.status = LM_INVALID_STEPS; return; }
    s.radius *= 0.5;
    if (s.iterations >= opt.max_iterations) { s.status = LM_ITERATION_CAP; return; }
  }
  s.invalid = 0;
  double delta[6];
  SICP_UNROLL
  for (int j = 0; j < 6; ++j) delta[j] = step[j] * s.scale[j];
  se3::plus<WAVE>(s.x, delta, s.pose);
  s.model_change = model_change;
  s.phase = 1;
}

}  // namespace detail

SICP_HD inline void lm_init(LmState& s, const LmOptions& opt, const double* x0) {
  s.opt = opt;
  SICP_UNROLL
  for (int i = 0; i < 7; ++i) { s.pose[i] = x0[i]; s.x[i] = x0[i]; }
  SICP_UNROLL
  for (int i = 0; i < 36; ++i) s.H[i] = 0;
  SICP_UNROLL
  for (int i = 0; i < 6; ++i) { s.g[i] = 0; s.scale[i] = 1; s.diag[i] = 0; }
  s.cost = 0; s.x_norm = 0;
  s.radius = opt.initial_radius; s.decrease_factor = 2.0; s.model_change = 0;
  s.reuse_diagonal = 0; s.invalid = 0; s.iterations = 0; s.evaluations = 0;
  s.phase = 0;
  s.status = LM_RUNNING;
  s.pending = 0; s.pad_ = 0;
}

// out28 = [H upper 21 | g 6 | cost] evaluated at s.pose.  `finite_known` >= 0: the caller has already tested the 28 sums with
// the predicate below (the GPU does it with 28 lanes at once instead of 84 instructions of the one lane that runs this).
// WAVE (GPU only): the caller is a whole wavefront, all 64 lanes with the same s and o; every lane ends with the same s.
template <bool WAVE = false>
SICP_HD inline void lm_feed(LmCore& s, const LmOptions& opt, const double* o, int finite_known = -1) {
  using namespace detail;
  if (s.status != LM_RUNNING) return;
  s.evaluations++;
  // Ceres rejects an evaluation with a non-finite residual or Jacobian entry (ResidualBlock::Evaluate ->
  // IsArrayValid): at the start point the solve FAILS ("Initial residual and Jacobian evaluation failed", x stays
  // x0); at a candidate the step is "treated as a step with infinite cost" (trust_region_minimizer.cc:
  // candidate_cost = DBL_MAX), i.e. rejected.  Here a non-finite residual shows up as a non-finite sum.  (The cost
  // entry alone would not do: fast_log.hpp maps NaN / Inf to a finite number; H and g carry the NaN.)
  bool finite = finite_known != 0;
  if (finite_known < 0)
    SICP_UNROLL
    for (int k = 0; k < 28; ++k) finite = finite && (o[k] - o[k] == 0.0);
  if (s.phase == 0) {
    if (!finite) { s.status = LM_EVAL_FAILED; return; }
    unpack28(o, s.H, s.g, &s.cost);
    s.x_norm = se3::norm7(s.x);
    SICP_UNROLL
    for (int j = 0; j < 6; ++j) s.scale[j] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(s.H[6 * j + j])) : 1.0;
  } else {
    const double cand_cost = finite ? o[27] : 1.7976931348623157e308;
    double diff[7];
    SICP_UNROLL
    for (int i = 0; i < 7; ++i) diff[i] = s.x[i] - s.pose[i];
    if (se3::norm7(diff) <= opt.parameter_tolerance * (s.x_norm + opt.parameter_tolerance)) { s.status = LM_CONVERGED; return; }
    const double cost_change = s.cost - cand_cost;
    if (fabs(cost_change) <= opt.function_tolerance * s.cost) { s.status = LM_CONVERGED; return; }
    const double rel = cost_change / s.model_change;
    if (rel > opt.min_relative_decrease) {
      SICP_UNROLL
      for (int i = 0; i < 7; ++i) s.x[i] = s.pose[i];
      s.x_norm = se3::norm7(s.x);
      unpack28(o, s.H, s.g, &s.cost);
      const double t = 2.0 * rel - 1.0;
      s.radius = fmin(opt.max_radius, s.radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
      s.decrease_factor = 2.0;
      s.reuse_diagonal = 0;
    } else {
      s.radius /= s.decrease_factor;
      s.decrease_factor *= 2.0;
      s.reuse_diagonal = 1;
    }
  }
  lm_propose<WAVE>(s, opt);
}

SICP_HD inline void lm_feed(LmState& s, const double* o) { lm_feed<false>(s, s.opt, o); }

}  // namespace sicp
#endif
