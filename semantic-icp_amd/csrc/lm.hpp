// lm.hpp -- host-side 6-DoF Levenberg-Marquardt driver (product code).
//
// Replaces the inner ceres::Solve of the reference (em_icp.hpp:162-177, gicp.hpp:138-151,
// semantic_icp.hpp:136-149): one SE3 parameter block (7 ambient / 6 tangent), trust-region +
// Levenberg-Marquardt with Ceres' default step control (Ceres 1.14..2.1
// trust_region_minimizer.cc, levenberg_marquardt_strategy.cc), restated because Ceres is not
// vendored.  The GPU supplies, per evaluation, the 28 numbers a dense Jacobian would be reduced
// to anyway: H = J^T J (robustified), g = J^T r, cost.  DENSE_QR on [J; D] and Cholesky on
// H + D^2 solve the same 6x6 system.
//
// Unlike Ceres, one evaluation returns cost, gradient and H together, so an accepted step does
// not need a second sweep at the same point (Ceres evaluates the candidate cost first and the
// Jacobian again after accepting); the iterates are the same.
#ifndef SICP_LM_HPP_
#define SICP_LM_HPP_

#include <cmath>
#include <cstring>

#include "se3.hpp"

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
  bool jacobi_scaling = true;
};

struct LmResult {
  int status = 0;  // 0 converged, 1 iteration cap, 2 too many invalid steps, <0 evaluation failed
  int iterations = 0;
  int evaluations = 0;
  double cost = 0;
};

namespace detail {

inline bool chol6_solve(const double* A, const double* b, double* y) {
  double L[36] = {0};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return false;
        L[6 * i + i] = std::sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * z[k];
    z[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * y[k];
    y[i] = s / L[6 * i + i];
  }
  return true;
}

inline void unpack28(const double* o, double* H, double* g, double* cost) {
  int k = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) { H[6 * a + b] = o[k]; H[6 * b + a] = o[k]; ++k; }
  for (int a = 0; a < 6; ++a) g[a] = o[21 + a];
  *cost = o[27];
}

// Ceres: ||x - Plus(x, -g)||_inf  (ambient coordinates)
inline double gradient_max_norm(const double* x, const double* g) {
  double ng[6], xp[7], m = 0;
  for (int i = 0; i < 6; ++i) ng[i] = -g[i];
  se3::plus(x, ng, xp);
  for (int i = 0; i < 7; ++i) m = std::fmax(m, std::fabs(x[i] - xp[i]));
  return m;
}

}  // namespace detail

// eval(qt, out28) -> 0 on success
template <class Eval>
LmResult lm_solve(const LmOptions& opt, Eval&& eval, const double* init_qt, double* out_qt) {
  using namespace detail;
  LmResult res;
  double x[7], o[28], H[36], g[6], cost;
  std::memcpy(x, init_qt, sizeof x);
  if (eval(x, o) != 0) { res.status = -1; std::memcpy(out_qt, x, sizeof x); return res; }
  res.evaluations++;
  unpack28(o, H, g, &cost);
  double x_norm = se3::norm7(x);
  double scale[6];
  for (int j = 0; j < 6; ++j) scale[j] = opt.jacobi_scaling ? 1.0 / (1.0 + std::sqrt(H[6 * j + j])) : 1.0;
  double radius = opt.initial_radius, decrease_factor = 2.0, diag[6] = {0, 0, 0, 0, 0, 0};
  bool reuse_diagonal = false;
  int invalid = 0;
  for (;;) {
    if (res.iterations >= opt.max_iterations) { res.status = 1; break; }
    if (gradient_max_norm(x, g) <= opt.gradient_tolerance) break;
    if (radius <= opt.min_radius) break;
    res.iterations++;
    double Hs[36], gs[6];
    for (int a = 0; a < 6; ++a) {
      gs[a] = g[a] * scale[a];
      for (int b = 0; b < 6; ++b) Hs[6 * a + b] = H[6 * a + b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal)
      for (int j = 0; j < 6; ++j) diag[j] = std::fmin(std::fmax(Hs[6 * j + j], opt.min_lm_diagonal), opt.max_lm_diagonal);
    double A[36], y[6], step[6];
    std::memcpy(A, Hs, sizeof A);
    for (int j = 0; j < 6; ++j) {
      const double lm = std::sqrt(diag[j] / radius);
      A[6 * j + j] += lm * lm;
    }
    reuse_diagonal = true;
    const bool ok = chol6_solve(A, gs, y);
    double model_change = 0;
    if (ok) {
      double sg = 0, sHs = 0;
      for (int a = 0; a < 6; ++a) step[a] = -y[a];
      for (int a = 0; a < 6; ++a) {
        sg += step[a] * gs[a];
        double r = 0;
        for (int b = 0; b < 6; ++b) r += Hs[6 * a + b] * step[b];
        sHs += step[a] * r;
      }
      model_change = -(sg + 0.5 * sHs);
    }
    if (!ok || !(model_change > 0.0)) {
      if (++invalid >= opt.max_consecutive_invalid_steps) { res.status = 2; break; }
      radius *= 0.5;
      continue;
    }
    invalid = 0;
    double delta[6], cand[7], oc[28];
    for (int j = 0; j < 6; ++j) delta[j] = step[j] * scale[j];
    se3::plus(x, delta, cand);
    if (eval(cand, oc) != 0) { res.status = -1; break; }
    res.evaluations++;
    const double cand_cost = oc[27];
    double diff[7];
    for (int i = 0; i < 7; ++i) diff[i] = x[i] - cand[i];
    if (se3::norm7(diff) <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) break;
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= opt.function_tolerance * cost) break;
    const double rel = cost_change / model_change;
    if (rel > opt.min_relative_decrease) {
      std::memcpy(x, cand, sizeof x);
      x_norm = se3::norm7(x);
      unpack28(oc, H, g, &cost);
      const double t = 2.0 * rel - 1.0;
      radius = std::fmin(opt.max_radius, radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  std::memcpy(out_qt, x, sizeof x);
  res.cost = cost;
  return res;
}

}  // namespace sicp
#endif
