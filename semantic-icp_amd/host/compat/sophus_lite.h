// sophus_lite.h -- the part of Sophus the reference's public surface and drivers use: SE3d (ctor from
// Matrix4d, fitToSE3, matrix(), inverse(), operator*, log(), exp(), data(), so3(), translation()) and SO3d
// (log(), matrix()), over the engine's own SE(3) (csrc/se3.hpp).  Used only when the real Sophus is not
// installed.
#ifndef SICP_COMPAT_SOPHUS_LITE_H_
#define SICP_COMPAT_SOPHUS_LITE_H_
#include <cmath>
#include <cstring>

#include "../../csrc/se3.hpp"
#include "eigen_lite.h"

namespace Sophus {

template <class T> struct Constants { static T epsilon() { return T(1e-10); } };

// Sophus::SO3d as far as exec/kitti_metrics.h:35 / scenenet_metrics.h:38 go: `diff.so3().log().squaredNorm()`
class SO3d {
 public:
  typedef Eigen::Vector3d Tangent;
  SO3d() { q_[0] = q_[1] = q_[2] = 0; q_[3] = 1; }
  static SO3d fromQuaternion(const double* xyzw) { SO3d s; std::memcpy(s.q_, xyzw, sizeof s.q_); return s; }
  const double* data() const { return q_; }  // [qx qy qz qw]
  // Sophus SO3::log: the rotation vector (the omega half of SE3::log)
  Tangent log() const {
    const double qt[7] = {q_[0], q_[1], q_[2], q_[3], 0, 0, 0};
    double a[6];
    sicp::se3::log(qt, a);
    return Tangent(a[3], a[4], a[5]);
  }
  Eigen::Matrix3d matrix() const {
    const double qt[7] = {q_[0], q_[1], q_[2], q_[3], 0, 0, 0};
    double R[9]; sicp::se3::rotation(qt, R);
    Eigen::Matrix3d m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j];
    return m;
  }
  SO3d inverse() const { SO3d s; s.q_[0] = -q_[0]; s.q_[1] = -q_[1]; s.q_[2] = -q_[2]; s.q_[3] = q_[3]; return s; }
 private:
  double q_[4];
};

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

  // Sophus::SE3::fitToSE3 (exec/kitti_metrics.h:24, scenenet_metrics.h:26, kitti_eval.cc:244): the 3x3 block goes
  // through SO3::fitToSO3 = makeRotationMatrix: R = U diag(1, 1, det(U) det(V)) V^T of the block's SVD -- the
  // nearest rotation.  The SVD is taken here from the Jacobi eigen-decomposition of B^T B (V, sigma^2), U = B V / sigma.
  static SE3d fitToSE3(const Eigen::Matrix4d& T) {
    double B[3][3], A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i][j] = T(i, j);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = B[0][i] * B[0][j] + B[1][i] * B[1][j] + B[2][i] * B[2][j];
    for (int sweep = 0; sweep < 60; ++sweep) {  // cyclic Jacobi on the symmetric A
      const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
      if (off < 1e-300) break;
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          if (A[p][q] == 0.0) continue;
          const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
          for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
          for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
        }
    }
    int order[3] = {0, 1, 2};  // singular values descending
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (A[order[b]][order[b]] > A[order[a]][order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    double U[3][3], Vs[3][3];
    for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) Vs[k][c] = V[k][order[c]];
    for (int c = 0; c < 3; ++c) {
      double u[3], n = 0;
      for (int k = 0; k < 3; ++k) { u[k] = B[k][0] * Vs[0][c] + B[k][1] * Vs[1][c] + B[k][2] * Vs[2][c]; n += u[k] * u[k]; }
      n = std::sqrt(n);
      if (c == 2 || !(n > 0)) {  // the last (possibly null) direction: completed from the other two
        if (c == 2) { u[0] = U[1][0] * U[2][1] - U[2][0] * U[1][1]; u[1] = U[2][0] * U[0][1] - U[0][0] * U[2][1]; u[2] = U[0][0] * U[1][1] - U[1][0] * U[0][1]; n = 1; }
        else { u[0] = c == 0; u[1] = c == 1; u[2] = 0; n = 1; }
      }
      for (int k = 0; k < 3; ++k) U[k][c] = u[k] / n;
    }
    // U's third column was completed as u0 x u1, i.e. det(U) = +1; fold det(V) into the last column of V
    const double detV = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) - Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                        Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
    if (detV < 0) for (int k = 0; k < 3; ++k) Vs[k][2] = -Vs[k][2];
    Eigen::Matrix4d M = T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) = U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + U[i][2] * Vs[j][2];
    for (int j = 0; j < 4; ++j) M(3, j) = j == 3;
    return SE3d(M);
  }

  double* data() { return d_; }              // [qx qy qz qw tx ty tz], Sophus storage order
  const double* data() const { return d_; }

  Eigen::Matrix3d rotationMatrix() const {
    double R[9]; sicp::se3::rotation(d_, R);
    Eigen::Matrix3d m; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j];
    return m;
  }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(d_[4], d_[5], d_[6]); }
  SO3d so3() const { return SO3d::fromQuaternion(d_); }
  Eigen::Matrix<double, 3, 4> matrix3x4() const {
    Eigen::Matrix<double, 3, 4> m;
    double R[9]; sicp::se3::rotation(d_, R);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) m(i, j) = R[3 * i + j]; m(i, 3) = d_[4 + i]; }
    return m;
  }
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
