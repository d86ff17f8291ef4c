// device_geometry.hpp -- the closed-form GICP residual shared by the weight and the solve kernels
// (device code; included inside the kernel files).
#ifndef SICP_DEVICE_GEOMETRY_HPP_
#define SICP_DEVICE_GEOMETRY_HPP_
#include "kernels.h"

namespace sicp {

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
  // pure float64 algebra with tolerance-level parity (1e-9).  NOT contracted: this function is inlined into several
  // kernels (the weight kernels, the label kernel, the search's weight epilogue) and whether a multiply-add pair fuses
  // would be each context's own choice -- the as-double Probability() branch (quirk Q1 off) then differs in the last
  // bit from kernel to kernel.  Separate multiplies and adds are the same everywhere.
#pragma clang fp contract(off)
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

// GICPCostFunction::Probability as the reference uses it (gicp_cost_function.h:75-87 through em_icp.hpp:108):
// the density det(2 pi A)^-1/2 exp(-r/2), converted to bool (quirk Q1: 1 unless the product underflows to
// exactly 0; NaN -> 1).  pow() and exp() cost ~300 instructions to answer a question that is decided long
// before the last digit: with 1e-60 < det A < 1e30 (det of A = C_t + R C_s R^T is at most 8 for real inputs) the
// power lies in (6e-17, 6e28), so the product cannot be 0 for r < 1300 (exp(-650) = 5e-283: the product is above
// 3e-299, a normal number) and is exactly 0 for r > 1600 (exp(-800) is 0, below the smallest denormal, times a
// finite power).  The band between them -- and every determinant outside that range, NaN, Inf, det <= 0 -- takes
// the literal formula, so the decision is the reference's for every input.
__device__ __forceinline__ double geometric_gate(const Corr& c, int bool_probability) {
  const double two_pi = 6.283185307179586;
  if (bool_probability) {
    const bool sane = c.detA > 1e-60 && c.detA < 1e30;
    if (sane && c.r < 1300.0) return 1.0;
    if (sane && c.r > 1600.0) return 0.0;
    const double probability = pow(two_pi * two_pi * two_pi * c.detA, -0.5) * exp(-0.5 * c.r);
    return (probability != 0.0) ? 1.0 : 0.0;  // quirk Q1: double -> bool (NaN -> true)
  }
  return pow(two_pi * two_pi * two_pi * c.detA, -0.5) * exp(-0.5 * c.r);
}

}  // namespace sicp
#endif
