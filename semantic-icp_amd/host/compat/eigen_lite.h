// eigen_lite.h -- the slice of Eigen's dense API that the reference's class surface AND its drivers
// (exec/kitti_eval.cc, nyu_eval.cc, scenenet_eval.cc, roc_eval.cc, test_icp.cc and the *_metrics.h /
// read_confusion_matrix.h they include) touch, so that those files compile and link unchanged without
// Eigen installed (it is not, in this image):
//   Matrix<T,R,C,Options> fixed or Dynamic, ColMajor (default, as Eigen) or RowMajor storage behind data();
//   Identity / Zero / Constant, (r,c) / (i) / [i], block<R,C>(i,j) as an assignable view, Map<Matrix<..>>,
//   cast<U>(), transpose(), + - * (eager: no expression templates), squaredNorm / norm / trace / dot / cross,
//   `m << a, b, c;` and `os << m` in Eigen's default format (columns right-aligned to the widest coefficient).
// Coefficients are zero-initialised (Eigen leaves them uninitialised).  With the real Eigen present, put its
// include directory first and define SICP_HAVE_REAL_DEPS: nothing here is used then.
#ifndef SICP_COMPAT_EIGEN_LITE_H_
#define SICP_COMPAT_EIGEN_LITE_H_
#include <cassert>
#include <cmath>
#include <cstddef>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace Eigen {

const int Dynamic = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 0x1, AutoAlign = 0, DontAlign = 0x2 };
typedef std::ptrdiff_t Index;

template <typename T, int R, int C, int Options = ((R == 1 && C != 1) ? RowMajor : ColMajor), int MaxR = R, int MaxC = C>
class Matrix;
template <typename Xpr, int BR, int BC> class Block;
template <typename Derived> class CommaInitializer;

namespace internal {
template <typename D> struct traits;
template <typename T, int R, int C, int O, int MR, int MC>
struct traits<Matrix<T, R, C, O, MR, MC>> {
  typedef T Scalar;
  enum { Rows = R, Cols = C };
};
template <typename X, int BR, int BC>
struct traits<Block<X, BR, BC>> {
  typedef typename traits<X>::Scalar Scalar;
  enum { Rows = BR, Cols = BC };
};
// the size of a product / sum when one side is Dynamic
constexpr int pick(int a, int b) { return a == Dynamic ? b : a; }
}  // namespace internal

// Everything a matrix-like object (Matrix, Map, Block) can be read as.  Derived supplies rows(), cols(),
// coeff(r,c) and -- when writable -- coeffRef(r,c).
template <typename Derived>
class DenseBase {
 public:
  typedef typename internal::traits<Derived>::Scalar Scalar;
  enum { RowsAtCompileTime = internal::traits<Derived>::Rows, ColsAtCompileTime = internal::traits<Derived>::Cols };
  typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;

  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  Derived& derived() { return *static_cast<Derived*>(this); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }

  Scalar operator()(Index r, Index c) const { return derived().coeff(r, c); }
  Scalar& operator()(Index r, Index c) { return derived().coeffRef(r, c); }
  // linear access: vectors only (as in Eigen)
  Scalar operator()(Index i) const { return cols() == 1 ? derived().coeff(i, 0) : derived().coeff(0, i); }
  Scalar& operator()(Index i) { return cols() == 1 ? derived().coeffRef(i, 0) : derived().coeffRef(0, i); }
  Scalar operator[](Index i) const { return (*this)(i); }
  Scalar& operator[](Index i) { return (*this)(i); }
  Scalar x() const { return (*this)(0); }
  Scalar y() const { return (*this)(1); }
  Scalar z() const { return (*this)(2); }
  Scalar w() const { return (*this)(3); }

  template <typename Other>
  Derived& assign(const DenseBase<Other>& o) {
    derived().resize_like(o.rows(), o.cols());
    assert(rows() == o.rows() && cols() == o.cols());
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) derived().coeffRef(r, c) = o(r, c);
    return derived();
  }

  PlainObject eval() const { return PlainObject(*this); }

  template <typename U>
  Matrix<U, RowsAtCompileTime, ColsAtCompileTime> cast() const {
    Matrix<U, RowsAtCompileTime, ColsAtCompileTime> o;
    o.resize_like(rows(), cols());
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) o(r, c) = static_cast<U>((*this)(r, c));
    return o;
  }
  Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const {
    Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> o;
    o.resize_like(cols(), rows());
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) o(c, r) = (*this)(r, c);
    return o;
  }

  template <typename Other>
  Matrix<Scalar, RowsAtCompileTime, internal::traits<Other>::Cols> operator*(const DenseBase<Other>& b) const {
    assert(cols() == b.rows());
    Matrix<Scalar, RowsAtCompileTime, internal::traits<Other>::Cols> o;
    o.resize_like(rows(), b.cols());
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < b.cols(); ++c) {
        Scalar s = Scalar(0);
        for (Index k = 0; k < cols(); ++k) s += (*this)(r, k) * b(k, c);
        o(r, c) = s;
      }
    return o;
  }
  template <typename Other> PlainObject operator+(const DenseBase<Other>& b) const { return zip(b, +1); }
  template <typename Other> PlainObject operator-(const DenseBase<Other>& b) const { return zip(b, -1); }
  PlainObject operator-() const { return (*this) * Scalar(-1); }
  PlainObject operator*(Scalar s) const {
    PlainObject o(*this);
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) o(r, c) *= s;
    return o;
  }
  PlainObject operator/(Scalar s) const {
    PlainObject o(*this);
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) o(r, c) /= s;
    return o;
  }
  friend PlainObject operator*(Scalar s, const DenseBase& m) { return m * s; }
  template <typename Other> Derived& operator+=(const DenseBase<Other>& b) { return assign((*this) + b); }
  template <typename Other> Derived& operator-=(const DenseBase<Other>& b) { return assign((*this) - b); }
  Derived& operator*=(Scalar s) { return assign((*this) * s); }
  Derived& operator/=(Scalar s) { return assign((*this) / s); }

  Scalar squaredNorm() const {
    Scalar s = Scalar(0);
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) s += (*this)(r, c) * (*this)(r, c);
    return s;
  }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  PlainObject normalized() const { return (*this) / norm(); }
  Scalar sum() const {
    Scalar s = Scalar(0);
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) s += (*this)(r, c);
    return s;
  }
  Scalar trace() const {
    Scalar s = Scalar(0);
    for (Index i = 0; i < (rows() < cols() ? rows() : cols()); ++i) s += (*this)(i, i);
    return s;
  }
  template <typename Other> Scalar dot(const DenseBase<Other>& b) const {
    assert(size() == b.size());
    Scalar s = Scalar(0);
    for (Index i = 0; i < size(); ++i) s += (*this)(i) * b(i);
    return s;
  }
  template <typename Other> PlainObject cross(const DenseBase<Other>& b) const {
    assert(size() == 3 && b.size() == 3);
    PlainObject o;
    const DenseBase& a = *this;
    o(0) = a(1) * b(2) - a(2) * b(1);
    o(1) = a(2) * b(0) - a(0) * b(2);
    o(2) = a(0) * b(1) - a(1) * b(0);
    return o;
  }
  Scalar determinant() const {  // 2x2 and 3x3, what pose code needs
    assert(rows() == cols() && rows() <= 3);
    const DenseBase& m = *this;
    if (rows() == 1) return m(0, 0);
    if (rows() == 2) return m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0);
    return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
           m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
  }
  bool isApprox(const DenseBase& b, Scalar prec = Scalar(1e-12)) const {
    return ((*this) - b).squaredNorm() <= prec * prec * std::min(squaredNorm(), b.squaredNorm());
  }

  void setZero() { setConstant(Scalar(0)); }
  void setConstant(Scalar v) {
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) derived().coeffRef(r, c) = v;
  }
  void setIdentity() {
    setZero();
    for (Index i = 0; i < (rows() < cols() ? rows() : cols()); ++i) derived().coeffRef(i, i) = Scalar(1);
  }

  // views (writable on a non-const object; a const object hands out a copy)
  template <int BR, int BC> Block<Derived, BR, BC> block(Index i, Index j) { return Block<Derived, BR, BC>(derived(), i, j, BR, BC); }
  template <int BR, int BC> Matrix<Scalar, BR, BC> block(Index i, Index j) const {
    Matrix<Scalar, BR, BC> o;
    for (Index r = 0; r < BR; ++r)
      for (Index c = 0; c < BC; ++c) o(r, c) = (*this)(i + r, j + c);
    return o;
  }
  Block<Derived, Dynamic, Dynamic> block(Index i, Index j, Index nr, Index nc) { return Block<Derived, Dynamic, Dynamic>(derived(), i, j, nr, nc); }
  template <int BR, int BC> Block<Derived, BR, BC> topLeftCorner() { return block<BR, BC>(0, 0); }
  template <int BR, int BC> Matrix<Scalar, BR, BC> topLeftCorner() const { return block<BR, BC>(0, 0); }
  template <int BR, int BC> Block<Derived, BR, BC> topRightCorner() { return block<BR, BC>(0, cols() - BC); }
  template <int BR, int BC> Matrix<Scalar, BR, BC> topRightCorner() const { return block<BR, BC>(0, cols() - BC); }
  Block<Derived, RowsAtCompileTime, 1> col(Index j) { return Block<Derived, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
  Block<Derived, 1, ColsAtCompileTime> row(Index i) { return Block<Derived, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
  Matrix<Scalar, RowsAtCompileTime, 1> col(Index j) const {
    Matrix<Scalar, RowsAtCompileTime, 1> o;
    o.resize_like(rows(), 1);
    for (Index r = 0; r < rows(); ++r) o(r, 0) = (*this)(r, j);
    return o;
  }
  Matrix<Scalar, 1, ColsAtCompileTime> row(Index i) const {
    Matrix<Scalar, 1, ColsAtCompileTime> o;
    o.resize_like(1, cols());
    for (Index c = 0; c < cols(); ++c) o(0, c) = (*this)(i, c);
    return o;
  }

  // `m << a, b, c, ...;` fills row by row
  CommaInitializer<Derived> operator<<(const Scalar& s) { return CommaInitializer<Derived>(derived(), s); }

 protected:
  template <typename Other> PlainObject zip(const DenseBase<Other>& b, int sign) const {
    assert(rows() == b.rows() && cols() == b.cols());
    PlainObject o(*this);
    for (Index r = 0; r < rows(); ++r)
      for (Index c = 0; c < cols(); ++c) o(r, c) = sign > 0 ? (*this)(r, c) + b(r, c) : (*this)(r, c) - b(r, c);
    return o;
  }
};

template <typename Derived>
class CommaInitializer {
 public:
  typedef typename DenseBase<Derived>::Scalar Scalar;
  CommaInitializer(Derived& m, const Scalar& s) : m_(m), i_(0) { put(s); }
  CommaInitializer& operator,(const Scalar& s) { put(s); return *this; }
 private:
  void put(const Scalar& s) {
    assert(i_ < m_.size());
    m_.coeffRef(i_ / m_.cols(), i_ % m_.cols()) = s;
    ++i_;
  }
  Derived& m_;
  Index i_;
};

namespace internal {
// coefficient storage: an array for fixed sizes, a vector + run-time shape when either extent is Dynamic
template <typename T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)>
struct Storage {
  T d[R * C > 0 ? R * C : 1];
  Storage() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
  static constexpr Index rows() { return R; }
  static constexpr Index cols() { return C; }
  void resize(Index r, Index c) { assert(r == R && c == C); (void)r; (void)c; }
  T* data() { return d; }
  const T* data() const { return d; }
};
template <typename T, int R, int C>
struct Storage<T, R, C, true> {
  std::vector<T> d;
  Index r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) {
    assert((R == Dynamic || r == R) && (C == Dynamic || c == C));
    if (r != r_ || c != c_) { r_ = r; c_ = c; d.assign((size_t)(r * c), T(0)); }
  }
  T* data() { return d.data(); }
  const T* data() const { return d.data(); }
};
}  // namespace internal

template <typename T, int R, int C, int Options, int MaxR, int MaxC>
class Matrix : public DenseBase<Matrix<T, R, C, Options, MaxR, MaxC>> {
  typedef DenseBase<Matrix> Base;
 public:
  typedef T Scalar;
  enum { IsRowMajor = (Options & RowMajor) ? 1 : 0 };

  Matrix() {}
  Matrix(const Matrix&) = default;
  Matrix& operator=(const Matrix&) = default;
  // any matrix-like object of the same shape, whatever its storage order (Map, Block, RowMajor <-> ColMajor)
  template <typename Other> Matrix(const DenseBase<Other>& o) { Base::assign(o); }
  template <typename Other> Matrix& operator=(const DenseBase<Other>& o) { return Base::assign(o); }
  // Matrix(rows, cols) when Dynamic, the two coefficients of a fixed 2-vector otherwise (Eigen's own dispatch)
  template <typename A, typename B, typename = typename std::enable_if<std::is_arithmetic<A>::value && std::is_arithmetic<B>::value>::type>
  Matrix(const A& a, const B& b) {
    if (R == Dynamic || C == Dynamic) s_.resize((Index)a, (Index)b);
    else { assert(R * C == 2); s_.data()[0] = (T)a; s_.data()[1] = (T)b; }
  }
  explicit Matrix(Index n) {  // dynamic vector of n coefficients
    if (R == Dynamic && C != Dynamic) s_.resize(n, C);
    else if (C == Dynamic && R != Dynamic) s_.resize(R, n);
    else assert(R * C == n);
  }
  Matrix(T x, T y, T z) { static_assert(R * C == 3, "3-vector constructor"); T* d = s_.data(); d[0] = x; d[1] = y; d[2] = z; }
  Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, "4-vector constructor"); T* d = s_.data(); d[0] = x; d[1] = y; d[2] = z; d[3] = w; }

  static Matrix Zero() { return Matrix(); }
  static Matrix Zero(Index r, Index c) { Matrix m; m.s_.resize(r, c); m.setZero(); return m; }
  static Matrix Constant(T v) { Matrix m; m.setConstant(v); return m; }
  static Matrix Constant(Index r, Index c, T v) { Matrix m; m.s_.resize(r, c); m.setConstant(v); return m; }
  static Matrix Ones() { return Constant(T(1)); }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  static Matrix Identity(Index r, Index c) { Matrix m; m.s_.resize(r, c); m.setIdentity(); return m; }

  void resize(Index r, Index c) { s_.resize(r, c); }
  T* data() { return s_.data(); }  // ColMajor unless Options says RowMajor, as in Eigen
  const T* data() const { return s_.data(); }

  // DenseBase's hooks
  Index rows_() const { return s_.rows(); }
  Index cols_() const { return s_.cols(); }
  T coeff(Index r, Index c) const { return s_.data()[offset(r, c)]; }
  T& coeffRef(Index r, Index c) { return s_.data()[offset(r, c)]; }
  void resize_like(Index r, Index c) { s_.resize(r, c); }

 private:
  Index offset(Index r, Index c) const {
    assert(r >= 0 && r < rows_() && c >= 0 && c < cols_());
    return IsRowMajor ? r * cols_() + c : c * rows_() + r;
  }
  internal::Storage<T, R, C> s_;
};

// Map<Matrix<..>>: a caller's array read (and written) in the plain type's storage order
template <typename Plain>
class Map : public DenseBase<Map<Plain>> {
 public:
  typedef typename Plain::Scalar Scalar;
  enum { R = internal::traits<Plain>::Rows, C = internal::traits<Plain>::Cols };
  explicit Map(Scalar* p) : p_(p), r_(R), c_(C) { static_assert(R != Dynamic && C != Dynamic, "fixed-size Map"); }
  Map(Scalar* p, Index r, Index c) : p_(p), r_(r), c_(c) {}
  template <typename Other> Map& operator=(const DenseBase<Other>& o) { return DenseBase<Map>::assign(o); }
  Map& operator=(const Map& o) { return DenseBase<Map>::assign(o); }
  Scalar* data() { return p_; }
  const Scalar* data() const { return p_; }
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  Scalar coeff(Index r, Index c) const { return p_[Plain::IsRowMajor ? r * c_ + c : c * r_ + r]; }
  Scalar& coeffRef(Index r, Index c) { return p_[Plain::IsRowMajor ? r * c_ + c : c * r_ + r]; }
  void resize_like(Index, Index) {}
 private:
  Scalar* p_;
  Index r_, c_;
};
namespace internal {
template <typename Plain>
struct traits<Map<Plain>> {
  typedef typename Plain::Scalar Scalar;
  enum { Rows = traits<Plain>::Rows, Cols = traits<Plain>::Cols };
};
}  // namespace internal

// block<BR,BC>(i,j) of a writable object: assignments go through to the parent
template <typename Xpr, int BR, int BC>
class Block : public DenseBase<Block<Xpr, BR, BC>> {
 public:
  typedef typename internal::traits<Xpr>::Scalar Scalar;
  Block(Xpr& x, Index i, Index j, Index nr, Index nc) : x_(x), i_(i), j_(j), r_(nr), c_(nc) {
    assert(i >= 0 && j >= 0 && i + nr <= x.rows() && j + nc <= x.cols());
  }
  template <typename Other> Block& operator=(const DenseBase<Other>& o) { return DenseBase<Block>::assign(o); }
  Block& operator=(const Block& o) { return DenseBase<Block>::assign(o); }
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  Scalar coeff(Index r, Index c) const { return static_cast<const Xpr&>(x_).coeff(i_ + r, j_ + c); }
  Scalar& coeffRef(Index r, Index c) { return x_.coeffRef(i_ + r, j_ + c); }
  void resize_like(Index, Index) {}
 private:
  Xpr& x_;
  Index i_, j_, r_, c_;
};

// Eigen's default IOFormat: every coefficient printed with the stream's own precision and flags, columns
// right-aligned to the widest coefficient, " " between columns, "\n" between rows
template <typename Derived>
std::ostream& operator<<(std::ostream& os, const DenseBase<Derived>& m) {
  if (m.size() == 0) return os;
  std::streamsize width = 0;
  for (Index c = 0; c < m.cols(); ++c)
    for (Index r = 0; r < m.rows(); ++r) {
      std::stringstream ss;
      ss.copyfmt(os);
      ss << m(r, c);
      width = std::max<std::streamsize>(width, (std::streamsize)ss.str().length());
    }
  for (Index r = 0; r < m.rows(); ++r) {
    if (r) os << "\n";
    for (Index c = 0; c < m.cols(); ++c) {
      if (c) os << " ";
      os.width(width);
      os << m(r, c);
    }
  }
  return os;
}

typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<float, Dynamic, Dynamic> MatrixXf;
typedef Matrix<int, Dynamic, Dynamic> MatrixXi;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<float, Dynamic, 1> VectorXf;
typedef Matrix<int, Dynamic, 1> VectorXi;

template <class T> using aligned_allocator = std::allocator<T>;

}  // namespace Eigen
#endif
