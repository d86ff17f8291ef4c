// eigen_lite.h -- the sliver of Eigen's fixed-size dense API that the reference's class
// surface and drivers touch (Matrix4d::Identity(), operator()(i,j), cast<float>(), <<, ...).
// Used only when the real Eigen is not installed (it is not, in this image); with Eigen present
// include <Eigen/Core> first and define SICP_HAVE_EIGEN.
#ifndef SICP_COMPAT_EIGEN_LITE_H_
#define SICP_COMPAT_EIGEN_LITE_H_
#include <cmath>
#include <cstddef>
#include <memory>
#include <ostream>

namespace Eigen {

template <typename T, int R, int C>
class Matrix {
 public:
  typedef T Scalar;
  Matrix() { for (int i = 0; i < R * C; ++i) d_[i] = T(0); }
  Matrix(T x, T y, T z) { static_assert(R * C == 3, "3-vector ctor"); d_[0] = x; d_[1] = y; d_[2] = z; }
  static Matrix Zero() { return Matrix(); }
  static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
  void setZero() { *this = Matrix(); }
  void setIdentity() { *this = Identity(); }
  T& operator()(int r, int c) { return d_[r * C + c]; }
  const T& operator()(int r, int c) const { return d_[r * C + c]; }
  T& operator()(int i) { return d_[i]; }
  const T& operator()(int i) const { return d_[i]; }
  T& operator[](int i) { return d_[i]; }
  const T& operator[](int i) const { return d_[i]; }
  static constexpr int rows() { return R; }
  static constexpr int cols() { return C; }
  T* data() { return d_; }  // NOTE: row-major (real Eigen defaults to column-major)
  const T* data() const { return d_; }
  template <typename U> Matrix<U, R, C> cast() const { Matrix<U, R, C> o; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) o(r, c) = static_cast<U>((*this)(r, c)); return o; }
  Matrix<T, C, R> transpose() const { Matrix<T, C, R> o; for (int r = 0; r < R; ++r) for (int c = 0; c < C; ++c) o(c, r) = (*this)(r, c); return o; }
  template <int C2> Matrix<T, R, C2> operator*(const Matrix<T, C, C2>& b) const {
    Matrix<T, R, C2> o;
    for (int r = 0; r < R; ++r) for (int c = 0; c < C2; ++c) { T s = T(0); for (int k = 0; k < C; ++k) s += (*this)(r, k) * b(k, c); o(r, c) = s; }
    return o;
  }
  Matrix operator+(const Matrix& b) const { Matrix o; for (int i = 0; i < R * C; ++i) o.d_[i] = d_[i] + b.d_[i]; return o; }
  Matrix operator-(const Matrix& b) const { Matrix o; for (int i = 0; i < R * C; ++i) o.d_[i] = d_[i] - b.d_[i]; return o; }
  Matrix operator*(T s) const { Matrix o; for (int i = 0; i < R * C; ++i) o.d_[i] = d_[i] * s; return o; }
  T squaredNorm() const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d_[i] * d_[i]; return s; }
  T norm() const { return std::sqrt(squaredNorm()); }
  T trace() const { T s = T(0); for (int i = 0; i < (R < C ? R : C); ++i) s += (*this)(i, i); return s; }

 private:
  T d_[R * C];
};

template <typename T, int R, int C>
std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
  for (int r = 0; r < R; ++r) {
    for (int c = 0; c < C; ++c) os << (c ? " " : "") << m(r, c);
    if (r + 1 < R) os << "\n";
  }
  return os;
}

typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 1> Vector3f;

template <class T> using aligned_allocator = std::allocator<T>;

}  // namespace Eigen
#endif
