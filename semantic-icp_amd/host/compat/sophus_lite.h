// sophus_lite.h -- the part of Sophus::SE3d the reference's public surface uses
// (ctor from Matrix4d, matrix(), inverse(), operator*, log(), exp(), data()), over the engine's
// own SE(3) (csrc/se3.hpp).  Used only when the real Sophus is not installed.
#ifndef SICP_COMPAT_SOPHUS_LITE_H_
#define SICP_COMPAT_SOPHUS_LITE_H_
#include <cmath>
#include <cstring>

#include "../../csrc/se3.hpp"
#include "eigen_lite.h"

namespace Sophus {

template <class T> struct Constants { static T epsilon() { return T(1e-10); } };

class SE3d {
 public:
  static const int num_parameters = 7;
  static const int DoF = 6;
  typedef Eigen::Matrix<double, 6, 1> Tangent;
  typedef Eigen::Matrix<double, 4, 4> Transformation;
  typedef Eigen::Vector3d Point;

  SE3d() { d_[0] = d_[1] = d_[2] = 0; d_[3] = 1; d_[4] = d_[5] = d_[6] = 0; }
  // Sophus SE3(Matrix4) -> Eigen::Quaternion(Matrix3) (Shepperd's branches, as Eigen does)
  explicit SE3d(const Eigen::Matrix4d& T) {
    const double m00 = T(0, 0), m11 = T(1, 1), m22 = T(2, 2);
    double t = m00 + m11 + m22, x, y, z, w;
    if (t > 0) {
      t = std::sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t;
      x = (T(2, 1) - T(1, 2)) * t; y = (T(0, 2) - T(2, 0)) * t; z = (T(1, 0) - T(0, 1)) * t;
    } else {
      int i = 0;
      if (m11 > m00) i = 1;
      if (m22 > T(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0);
      double q[3];
      q[i] = 0.5 * t; t = 0.5 / t;
      w = (T(k, j) - T(j, k)) * t; q[j] = (T(j, i) + T(i, j)) * t; q[k] = (T(k, i) + T(i, k)) * t;
      x = q[0]; y = q[1]; z = q[2];
    }
    d_[0] = x; d_[1] = y; d_[2] = z; d_[3] = w;
    d_[4] = T(0, 3); d_[5] = T(1, 3); d_[6] = T(2, 3);
  }
  static SE3d fromData(const double* qt) { SE3d s; std::memcpy(s.d_, qt, sizeof s.d_); return s; }

  // Sophus::SE3::fitToSE3 (used by exec/kitti_metrics.h:24): closest rotation to the 3x3 block.
  // Sophus takes U V^T of an SVD; the Newton polar iteration R <- (R + R^-T)/2 converges to the same
  // orthogonal factor for the nearly orthonormal matrices pose files hold.
  static SE3d fitToSE3(const Eigen::Matrix4d& T) {
    double R[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = T(i, j);
    for (int it = 0; it < 30; ++it) {
      const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
      const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
      if (!(std::fabs(det) > 0)) break;
      const double inv = 1.0 / det;
      // inverse transpose = cofactor matrix / det
      const double C[9] = {c00, c01, c02,
                           R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                           R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
      double delta = 0;
      for (int i = 0; i < 9; ++i) { const double n = 0.5 * (R[i] + C[i] * inv); delta += std::fabs(n - R[i]); R[i] = n; }
      if (delta < 1e-15) break;
    }
    Eigen::Matrix4d M = T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) = R[3 * i + j];
    return SE3d(M);
  }
  // rotation part of log(): what `so3().log()` yields in exec/kitti_metrics.h:35
  Eigen::Vector3d rotationLog() const { Tangent t = log(); return Eigen::Vector3d(t(3), t(4), t(5)); }

  double* data() { return d_; }              // [qx qy qz qw tx ty tz], Sophus storage order
  const double* data() const { return d_; }

  Eigen::Matrix3d rotationMatrix() const {
    double R[9]; sicp::se3::rotation(d_, R);
    Eigen::Matrix3d m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j];
    return m;
  }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(d_[4], d_[5], d_[6]); }
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d m = Eigen::Matrix4d::Identity();
    double R[9]; sicp::se3::rotation(d_, R);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j]; m(i, 3) = d_[4 + i]; }
    return m;
  }
  SE3d inverse() const { SE3d o; sicp::se3::inverse(d_, o.d_); return o; }
  SE3d operator*(const SE3d& b) const { SE3d o; sicp::se3::mul(d_, b.d_, o.d_); return o; }
  Eigen::Vector3d operator*(const Eigen::Vector3d& p) const {
    double R[9], v[3] = {p(0), p(1), p(2)}, o[3];
    sicp::se3::rotation(d_, R); sicp::se3::matvec3(R, v, o);
    return Eigen::Vector3d(o[0] + d_[4], o[1] + d_[5], o[2] + d_[6]);
  }
  Tangent log() const { Tangent t; sicp::se3::log(d_, t.data()); return t; }
  static SE3d exp(const Tangent& a) { SE3d o; sicp::se3::exp(a.data(), o.d_); return o; }

 private:
  double d_[7];
};

}  // namespace Sophus
#endif
