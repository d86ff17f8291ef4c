// se3.hpp -- host-side SE(3) for the engine (product code; the reference uses Sophus::SE3d,
// which is not vendored and not installed here, so its published formulas are restated).
//
// Storage: qt[7] = [qx qy qz qw tx ty tz] (Sophus order, gicp_cost_function.h:64-70).
// Tangent: [upsilon(3); omega(3)], updates are T * exp(delta)
// (local_parameterization_se3.h:17-25).
#ifndef SICP_SE3_HPP_
#define SICP_SE3_HPP_

#include <math.h>
#include <string.h>

// empty on the host; the kernel files define it as __host__ __device__ so that the device-resident
// solve (lm.hpp) runs the very same SE(3) code
#ifndef SICP_HD
#define SICP_HD
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SICP_SE3_UNROLL _Pragma("unroll")
#else
#define SICP_SE3_UNROLL
#endif

namespace sicp {
namespace se3 {

#define SICP_SE3_EPS 1e-10  /* Sophus::Constants<double>::epsilon() */
#define SICP_SE3_PI 3.14159265358979323846

// sin and cos of one angle; on the GPU one call shares the argument reduction (the one-lane LM
// step is instruction-count bound), the values are those of sin() and cos()
SICP_HD inline void sincos_pair(double a, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  ::sincos(a, s, c);
#else
  *s = sin(a);
  *c = cos(a);
#endif
}

// WAVE variants (GPU only): the caller is a whole 64-lane wavefront whose lanes all hold the SAME operands -- running the one-lane
// LM step on every lane costs the same instructions -- so independent pieces with the same instruction sequence go to
// different lanes and come back with v_readlane: the same operations on the same values, the same bits.
#if defined(__HIP_DEVICE_COMPILE__)
namespace wave {
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ double bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
}  // namespace wave
#endif

// Eigen Quaternion::toRotationMatrix (no normalisation), row-major
SICP_HD inline void rotation(const double* qt, double* R) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

SICP_HD inline void matvec3(const double* A, const double* v, double* o) {
  const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  const double b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  const double c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}

// rows 0..2 of the 4x4 matrix, row-major 3x4 (what pcl::transformPointCloud consumes)
SICP_HD inline void matrix34(const double* qt, double* M) {
  double R[9];
  rotation(qt, R);
  SICP_SE3_UNROLL
  for (int i = 0; i < 3; ++i) {
    M[4 * i + 0] = R[3 * i + 0]; M[4 * i + 1] = R[3 * i + 1]; M[4 * i + 2] = R[3 * i + 2];
    M[4 * i + 3] = qt[4 + i];
  }
}

// Sophus SE3::exp.  WAVE: the two sincos -- of theta / 2 for the quaternion, of theta for V -- are one call, lane 1 taking
// theta and every other lane theta / 2 (~180 of the ~470 instructions of a Plus).
template <bool WAVE = false>
SICP_HD inline void exp(const double* a, double* qt) {
  const double* w = a + 3;
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double theta = sqrt(theta_sq);
  double imag, real;
  double st_w = 0, ct_w = 0;
  (void)st_w; (void)ct_w;
  if (theta < SICP_SE3_EPS) {
    const double t4 = theta_sq * theta_sq;
    imag = 0.5 - theta_sq / 48.0 + t4 / 3840.0;
    real = 1.0 - theta_sq / 8.0 + t4 / 384.0;
  } else {
    double sh, ch;
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE) {
      double s2, c2;
      sincos_pair(wave::lane_id() == 1 ? theta : 0.5 * theta, &s2, &c2);
      sh = wave::bcast(s2, 0); ch = wave::bcast(c2, 0);
      st_w = wave::bcast(s2, 1); ct_w = wave::bcast(c2, 1);
    } else
#endif
    sincos_pair(0.5 * theta, &sh, &ch);
    imag = sh / theta;
    real = ch;
  }
  qt[0] = imag * w[0]; qt[1] = imag * w[1]; qt[2] = imag * w[2]; qt[3] = real;
  // V = I + (1-cos)/th^2 * W + (th - sin)/th^3 * W^2 ; small angle: V = R
  double V[9];
  if (theta < SICP_SE3_EPS) {
    rotation(qt, V);
  } else {
    double st, ct;
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE) { st = st_w; ct = ct_w; } else
#endif
    sincos_pair(theta, &st, &ct);
    const double c1 = (1 - ct) / theta_sq, c2 = (theta - st) / (theta_sq * theta);
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    SICP_SE3_UNROLL
    for (int i = 0; i < 3; ++i)
      SICP_SE3_UNROLL
      for (int j = 0; j < 3; ++j)
        W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    SICP_SE3_UNROLL
    for (int i = 0; i < 9; ++i) V[i] = c1 * W[i] + c2 * W2[i];
    V[0] += 1; V[4] += 1; V[8] += 1;
  }
  matvec3(V, a, qt + 4);
}

// Sophus SE3::log
SICP_HD inline void log(const double* qt, double* a) {
  const double sqn = qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2];
  const double n = sqrt(sqn), qw = qt[3];
  double f;
  if (n < SICP_SE3_EPS) {
    f = 2.0 / qw - (2.0 / 3.0) * sqn / (qw * qw * qw);
  } else if (fabs(qw) < SICP_SE3_EPS) {
    f = (qw > 0 ? SICP_SE3_PI : -SICP_SE3_PI) / n;
  } else {
    f = 2.0 * atan(n / qw) / n;
  }
  const double theta = f * n;
  const double w[3] = {f * qt[0], f * qt[1], f * qt[2]};
  double c;
  if (fabs(theta) < SICP_SE3_EPS) {
    c = 1.0 / 12.0;
  } else {
    const double h = 0.5 * theta;
    c = (1.0 - theta * cos(h) / (2.0 * sin(h))) / (theta * theta);
  }
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double Vi[9];
  SICP_SE3_UNROLL
  for (int i = 0; i < 3; ++i)
    SICP_SE3_UNROLL
    for (int j = 0; j < 3; ++j) {
      const double w2 = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
      Vi[3 * i + j] = -0.5 * W[3 * i + j] + c * w2 + (i == j ? 1.0 : 0.0);
    }
  matvec3(Vi, qt + 4, a);
  a[3] = w[0]; a[4] = w[1]; a[5] = w[2];
}

// group product (Sophus: quaternion product + first-order renormalisation)
SICP_HD inline void mul(const double* a, const double* b, double* out) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  double q[4];
  q[3] = aw * bw - ax * bx - ay * by - az * bz;
  q[0] = aw * bx + ax * bw + ay * bz - az * by;
  q[1] = aw * by + ay * bw + az * bx - ax * bz;
  q[2] = aw * bz + az * bw + ax * by - ay * bx;
  const double sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sq != 1.0) {
    const double s = 2.0 / (1.0 + sq);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  double R[9], t[3];
  rotation(a, R);
  matvec3(R, b + 4, t);
  const double o[7] = {q[0], q[1], q[2], q[3], a[4] + t[0], a[5] + t[1], a[6] + t[2]};
  SICP_SE3_UNROLL
  for (int i = 0; i < 7; ++i) out[i] = o[i];
}

SICP_HD inline void inverse(const double* a, double* out) {
  const double c[7] = {-a[0], -a[1], -a[2], a[3], 0, 0, 0};
  double R[9], t[3];
  rotation(c, R);
  matvec3(R, a + 4, t);
  const double o[7] = {c[0], c[1], c[2], c[3], -t[0], -t[1], -t[2]};
  SICP_SE3_UNROLL
  for (int i = 0; i < 7; ++i) out[i] = o[i];
}

// LocalParameterizationSE3::Plus
template <bool WAVE = false>
SICP_HD inline void plus(const double* qt, const double* delta, double* out) {
  double e[7];
  exp<WAVE>(delta, e);
  mul(qt, e, out);
}

SICP_HD inline double norm7(const double* a) {
  double s = 0;
  SICP_SE3_UNROLL
  for (int i = 0; i < 7; ++i) s += a[i] * a[i];
  return sqrt(s);
}

}  // namespace se3
}  // namespace sicp
#endif
