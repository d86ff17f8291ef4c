/*
 * sicp_oracle.c -- CPU restatement ("oracle") of the kxhit/semantic-icp hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sicp_oracle.h).  PARITY UNPINNED: the reference
 * cannot be compiled here and ships no golden vectors; this restatement is
 * pinned by tests/golden/ (independent numpy/scipy statement) instead.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference) or, for the un-vendored third-party pieces, the library
 * and routine whose published algorithm it restates.
 */
#include "sicp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Sophus::Constants<double>::epsilon() (sophus/common.hpp) */
#define SOPHUS_EPS 1e-10

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* small dense helpers, row-major 3x3                                         */
/* ------------------------------------------------------------------------- */
static void m3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      T[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof T);
}
static void m3_t(const double A[9], double T[9]) {
  double R[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = A[3 * j + i];
  memcpy(T, R, sizeof R);
}
static void m3_add(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 9; i++) C[i] = A[i] + B[i];
}
static void m3_vec(const double A[9], const double v[3], double o[3]) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
/* row vector times matrix: o = v^T A */
static void vec_m3(const double v[3], const double A[9], double o[3]) {
  double t[3];
  for (int j = 0; j < 3; j++) t[j] = v[0] * A[j] + v[1] * A[3 + j] + v[2] * A[6 + j];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static double m3_det(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}
/* Eigen Matrix3d::inverse(): cofactor / determinant, no pivoting (Eigen
 * LU/InverseImpl.h compute_inverse<.,3>) */
static void m3_inv(const double A[9], double I[9]) {
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c10 = A[5] * A[6] - A[3] * A[8];
  double c20 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c10 + A[2] * c20;
  double inv = 1.0 / det;
  double R[9];
  R[0] = c00 * inv;
  R[1] = (A[2] * A[7] - A[1] * A[8]) * inv;
  R[2] = (A[1] * A[5] - A[2] * A[4]) * inv;
  R[3] = c10 * inv;
  R[4] = (A[0] * A[8] - A[2] * A[6]) * inv;
  R[5] = (A[2] * A[3] - A[0] * A[5]) * inv;
  R[6] = c20 * inv;
  R[7] = (A[1] * A[6] - A[0] * A[7]) * inv;
  R[8] = (A[0] * A[4] - A[1] * A[3]) * inv;
  memcpy(I, R, sizeof R);
}

/* ------------------------------------------------------------------------- */
/* Sophus SE3d (third-party, not in tree; Sophus 1.0 so3.hpp / se3.hpp)        */
/* ------------------------------------------------------------------------- */

/* Eigen Quaternion::toRotationMatrix (formula quoted at
 * gicp_cost_function.h:110-120) */
void orc_se3_rotation(const double qt[7], double R[9]) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void orc_se3_matrix(const double qt[7], double M[16]) {
  double R[9];
  orc_se3_rotation(qt, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) M[4 * i + j] = R[3 * i + j];
    M[4 * i + 3] = qt[4 + i];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
}

/* SO3::expAndTheta */
static void so3_exp(const double w[3], double q[4], double* theta_out) {
  double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double theta = sqrt(theta_sq);
  double half = 0.5 * theta, imag, real;
  if (theta < SOPHUS_EPS) {
    double t4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
    real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
  } else {
    imag = sin(half) / theta;
    real = cos(half);
  }
  q[0] = imag * w[0]; q[1] = imag * w[1]; q[2] = imag * w[2]; q[3] = real;
  *theta_out = theta;
}

static void hat(const double w[3], double O[9]) {
  O[0] = 0;     O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2];  O[4] = 0;     O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0];  O[8] = 0;
}

/* SE3::exp, tangent = [upsilon; omega] */
void orc_se3_exp(const double a[6], double qt[7]) {
  double theta, q[4], O[9], O2[9], V[9];
  so3_exp(a + 3, q, &theta);
  hat(a + 3, O);
  m3_mul(O, O, O2);
  if (theta < SOPHUS_EPS) {
    double tmp[7] = {q[0], q[1], q[2], q[3], 0, 0, 0};
    orc_se3_rotation(tmp, V);
  } else {
    double tsq = theta * theta;
    double c1 = (1 - cos(theta)) / tsq, c2 = (theta - sin(theta)) / (tsq * theta);
    for (int i = 0; i < 9; i++) V[i] = c1 * O[i] + c2 * O2[i];
    V[0] += 1; V[4] += 1; V[8] += 1;
  }
  qt[0] = q[0]; qt[1] = q[1]; qt[2] = q[2]; qt[3] = q[3];
  m3_vec(V, a, qt + 4);
}

/* SO3::logAndTheta */
static void so3_log(const double q[4], double w[3], double* theta_out) {
  double sqn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  double n = sqrt(sqn), qw = q[3], f;
  if (n < SOPHUS_EPS) {
    double sw = qw * qw;
    f = 2.0 / qw - (2.0 / 3.0) * sqn / (qw * sw);
  } else {
    if (fabs(qw) < SOPHUS_EPS) {
      f = (qw > 0 ? M_PI : -M_PI) / n;
    } else {
      f = 2.0 * atan(n / qw) / n;
    }
  }
  *theta_out = f * n;
  w[0] = f * q[0]; w[1] = f * q[1]; w[2] = f * q[2];
}

/* SE3::log */
void orc_se3_log(const double qt[7], double a[6]) {
  double theta, w[3], O[9], O2[9], Vi[9];
  so3_log(qt, w, &theta);
  hat(w, O);
  m3_mul(O, O, O2);
  double c;
  if (fabs(theta) < SOPHUS_EPS) {
    c = 1.0 / 12.0;
  } else {
    double half = 0.5 * theta;
    c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
  }
  for (int i = 0; i < 9; i++) Vi[i] = -0.5 * O[i] + c * O2[i];
  Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
  m3_vec(Vi, qt + 4, a);
  a[3] = w[0]; a[4] = w[1]; a[5] = w[2];
}

/* SE3 group product: quaternion product with Sophus' first-order
 * renormalisation, t = t1 + R1 t2 */
void orc_se3_mul(const double a[7], const double b[7], double out[7]) {
  double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  double q[4];
  q[3] = aw * bw - ax * bx - ay * by - az * bz;
  q[0] = aw * bx + ax * bw + ay * bz - az * by;
  q[1] = aw * by + ay * bw + az * bx - ax * bz;
  q[2] = aw * bz + az * bw + ax * by - ay * bx;
  double sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sq != 1.0) {
    double s = 2.0 / (1.0 + sq);
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
  }
  double R[9], t[3];
  orc_se3_rotation(a, R);
  m3_vec(R, b + 4, t);
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
  out[4] = a[4] + t[0]; out[5] = a[5] + t[1]; out[6] = a[6] + t[2];
}

void orc_se3_inv(const double a[7], double out[7]) {
  double c[7] = {-a[0], -a[1], -a[2], a[3], 0, 0, 0};
  double R[9], t[3];
  orc_se3_rotation(c, R);
  m3_vec(R, a + 4, t);
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  out[4] = -t[0]; out[5] = -t[1]; out[6] = -t[2];
}

/* local_parameterization_se3.h:17-25  T * exp(delta) */
void orc_se3_plus(const double qt[7], const double delta[6], double out[7]) {
  double e[7];
  orc_se3_exp(delta, e);
  orc_se3_mul(qt, e, out);
}

/* Sophus SE3::Dx_this_mul_exp_x_at_0, used at local_parameterization_se3.h:30-36 */
void orc_se3_dx_this_mul_exp_x_at_0(const double qt[7], double J[42]) {
  for (int i = 0; i < 42; i++) J[i] = 0;
  const double c0 = 0.5 * qt[3], c1 = 0.5 * qt[2], c2 = -c1, c3 = 0.5 * qt[1],
               c4 = 0.5 * qt[0], c5 = -c4, c6 = -c3;
  J[0 * 6 + 3] = c0; J[0 * 6 + 4] = c2; J[0 * 6 + 5] = c3;
  J[1 * 6 + 3] = c1; J[1 * 6 + 4] = c0; J[1 * 6 + 5] = c5;
  J[2 * 6 + 3] = c6; J[2 * 6 + 4] = c4; J[2 * 6 + 5] = c0;
  J[3 * 6 + 3] = c5; J[3 * 6 + 4] = c6; J[3 * 6 + 5] = c2;
  double R[9];
  orc_se3_rotation(qt, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) J[(4 + i) * 6 + j] = R[3 * i + j];
}

/* ------------------------------------------------------------------------- */
/* pcl::transformPointCloud<PointT,double>: per point, rows evaluated in       */
/* double, left to right, then cast to float (em_icp.hpp:46-50)                */
/* ------------------------------------------------------------------------- */
void orc_transform_points(const double M[16], int n, const float* x, const float* y,
                          const float* z, float* ox, float* oy, float* oz) {
  for (int i = 0; i < n; i++) {
    double px = x[i], py = y[i], pz = z[i];
    ox[i] = (float)(M[0] * px + M[1] * py + M[2] * pz + M[3]);
    oy[i] = (float)(M[4] * px + M[5] * py + M[6] * pz + M[7]);
    oz[i] = (float)(M[8] * px + M[9] * py + M[10] * pz + M[11]);
  }
}

/* ------------------------------------------------------------------------- */
/* exact kNN, FLANN L2_Simple<float>: ((dx*dx)+dy*dy)+dz*dz in float32.        */
/* Results sorted ascending; ties -> lowest index (defined by this build; the  */
/* reference's tie order is traversal dependent).                              */
/* ------------------------------------------------------------------------- */
static inline float l2_simple(float ax, float ay, float az, float bx, float by, float bz) {
  /* volatile-free but contraction-free: compiled with -ffp-contract=off */
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* insert (d,i) into ascending list of length k (lexicographic on (d, idx)) */
static inline void topk_insert(float* bd, int* bi, int k, float d, int i) {
  if (!(d < bd[k - 1] || (d == bd[k - 1] && i < bi[k - 1]))) return;
  int j = k - 1;
  while (j > 0 && (d < bd[j - 1] || (d == bd[j - 1] && i < bi[j - 1]))) {
    bd[j] = bd[j - 1];
    bi[j] = bi[j - 1];
    j--;
  }
  bd[j] = d;
  bi[j] = i;
}

void orc_knn_brute(int nq, const float* qx, const float* qy, const float* qz, int nt,
                   const float* tx, const float* ty, const float* tz, int k, int* idx,
                   float* d2) {
#pragma omp parallel for schedule(static)
  for (int q = 0; q < nq; q++) {
    float* bd = d2 + (size_t)q * k;
    int* bi = idx + (size_t)q * k;
    for (int j = 0; j < k; j++) { bd[j] = INFINITY; bi[j] = -1; }
    float ax = qx[q], ay = qy[q], az = qz[q];
    for (int t = 0; t < nt; t++) {
      float d = l2_simple(ax, ay, az, tx[t], ty[t], tz[t]);
      /* candidates arrive in ascending index order: strict < keeps lowest idx */
      if (d < bd[k - 1]) topk_insert(bd, bi, k, d, t);
    }
  }
}

/* --- kd-tree (stands in for FLANN KDTreeSingleIndex: exact search) -------- */
typedef struct {
  int left, right;   /* children (node ids) or -1 for leaf */
  int begin, end;    /* point range in perm[] for leaves */
  int dim;
  float split;
} kd_node;

typedef struct {
  int n;
  int* perm;
  float *px, *py, *pz; /* points reordered by perm for cache locality */
  kd_node* nodes;
  int n_nodes, cap_nodes;
} kd_tree;

#define KD_LEAF 16

static float kd_coord(const float* x, const float* y, const float* z, int i, int d) {
  return d == 0 ? x[i] : (d == 1 ? y[i] : z[i]);
}

static void kd_select(int* perm, int lo, int hi, int kth, const float* x, const float* y,
                      const float* z, int d) {
  /* quickselect on perm[lo..hi) so that perm[kth] is in sorted position */
  while (hi - lo > 1) {
    int mid = lo + (hi - lo) / 2;
    float a = kd_coord(x, y, z, perm[lo], d), b = kd_coord(x, y, z, perm[mid], d),
          c = kd_coord(x, y, z, perm[hi - 1], d);
    float pv = (a < b) ? ((b < c) ? b : (a < c ? c : a)) : ((a < c) ? a : (b < c ? c : b));
    int i = lo, j = hi - 1;
    while (i <= j) {
      while (kd_coord(x, y, z, perm[i], d) < pv) i++;
      while (kd_coord(x, y, z, perm[j], d) > pv) j--;
      if (i <= j) {
        int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        i++; j--;
      }
    }
    if (kth <= j) hi = j + 1;
    else if (kth >= i) lo = i;
    else return;
  }
}

static int kd_build_rec(kd_tree* t, const float* x, const float* y, const float* z, int lo,
                        int hi) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes *= 2;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  int id = t->n_nodes++;
  kd_node nd;
  nd.left = nd.right = -1;
  nd.begin = lo; nd.end = hi; nd.dim = 0; nd.split = 0;
  if (hi - lo > KD_LEAF) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; i++)
      for (int d = 0; d < 3; d++) {
        float v = kd_coord(x, y, z, t->perm[i], d);
        if (v < mn[d]) mn[d] = v;
        if (v > mx[d]) mx[d] = v;
      }
    int d = 0;
    if (mx[1] - mn[1] > mx[d] - mn[d]) d = 1;
    if (mx[2] - mn[2] > mx[d] - mn[d]) d = 2;
    if (mx[d] > mn[d]) {
      int mid = lo + (hi - lo) / 2;
      kd_select(t->perm, lo, hi, mid, x, y, z, d);
      nd.dim = d;
      nd.split = kd_coord(x, y, z, t->perm[mid], d);
      t->nodes[id] = nd;
      int l = kd_build_rec(t, x, y, z, lo, mid);
      int r = kd_build_rec(t, x, y, z, mid, hi);
      t->nodes[id].left = l;
      t->nodes[id].right = r;
      return id;
    }
  }
  t->nodes[id] = nd;
  return id;
}

static kd_tree* kd_build(int n, const float* x, const float* y, const float* z) {
  kd_tree* t = (kd_tree*)calloc(1, sizeof(kd_tree));
  t->n = n;
  t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) t->perm[i] = i;
  t->cap_nodes = 64;
  t->nodes = (kd_node*)malloc(sizeof(kd_node) * (size_t)t->cap_nodes);
  if (n > 0) kd_build_rec(t, x, y, z, 0, n);
  t->px = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  t->py = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  t->pz = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    t->px[i] = x[t->perm[i]]; t->py[i] = y[t->perm[i]]; t->pz[i] = z[t->perm[i]];
  }
  return t;
}

static void kd_free(kd_tree* t) {
  if (!t) return;
  free(t->perm); free(t->px); free(t->py); free(t->pz); free(t->nodes); free(t);
}

/* Exactness argument: a point in the far half-space has |dx_dim| >= |q-split|,
 * and float rounding is monotone, so its L2_Simple distance is >= fl(diff*diff).
 * Pruning only when that bound is strictly greater than the current worst keeps
 * every (dist, idx)-lexicographic candidate. */
static void kd_search(const kd_tree* t, int node, float qx, float qy, float qz, int k,
                      float* bd, int* bi) {
  const kd_node* nd = &t->nodes[node];
  if (nd->left < 0) {
    for (int i = nd->begin; i < nd->end; i++) {
      float d = l2_simple(qx, qy, qz, t->px[i], t->py[i], t->pz[i]);
      topk_insert(bd, bi, k, d, t->perm[i]);
    }
    return;
  }
  float qv = nd->dim == 0 ? qx : (nd->dim == 1 ? qy : qz);
  float diff = qv - nd->split;
  int near = diff < 0 ? nd->left : nd->right;
  int far = diff < 0 ? nd->right : nd->left;
  kd_search(t, near, qx, qy, qz, k, bd, bi);
  float bound = diff * diff;
  if (!(bound > bd[k - 1])) kd_search(t, far, qx, qy, qz, k, bd, bi);
}

static void kd_knn(const kd_tree* t, float qx, float qy, float qz, int k, int* bi, float* bd) {
  for (int j = 0; j < k; j++) { bd[j] = INFINITY; bi[j] = -1; }
  if (t->n > 0) kd_search(t, 0, qx, qy, qz, k, bd, bi);
}

void orc_knn_kdtree(int nq, const float* qx, const float* qy, const float* qz, int nt,
                    const float* tx, const float* ty, const float* tz, int k, int* idx,
                    float* d2) {
  kd_tree* t = kd_build(nt, tx, ty, tz);
  /* single thread: the reference's correspondence loop is serial
   * (em_icp.hpp:57-156) */
  for (int q = 0; q < nq; q++)
    kd_knn(t, qx[q], qy[q], qz[q], k, idx + (size_t)q * k, d2 + (size_t)q * k);
  kd_free(t);
}

/* ------------------------------------------------------------------------- */
/* covariances                                                                */
/* ------------------------------------------------------------------------- */

/* Stand-in for Eigen::JacobiSVD<Matrix3d>(cov, ComputeFullU).matrixU() on a
 * symmetric input (em_icp.hpp:327): cyclic Jacobi eigen-decomposition; singular
 * values are |eigenvalues| sorted descending, U's columns the matching
 * eigenvectors (sign is irrelevant: only u u^T is used, em_icp.hpp:331-338). */
void orc_sym3_svd_u(const double Ain[9], double U[9], double sv[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(A, Ain, sizeof A);
  /* symmetrise from the lower triangle, which is what the reference fills
   * (cov(l,k) = cov(k,l), em_icp.hpp:322) */
  A[1] = A[3]; A[2] = A[6]; A[5] = A[7];
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int r = 0; r < 3; r++) {
      int p = P[r], q = Q[r];
      double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      double app = A[3 * p + p], aqq = A[3 * q + q];
      double tau = (aqq - app) / (2.0 * apq);
      double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
      /* A <- G^T A G, V <- V G, G = rotation in (p,q) */
      for (int k = 0; k < 3; k++) {
        double akp = A[3 * k + p], akq = A[3 * k + q];
        A[3 * k + p] = c * akp - s * akq;
        A[3 * k + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {
        double apk = A[3 * p + k], aqk = A[3 * q + k];
        A[3 * p + k] = c * apk - s * aqk;
        A[3 * q + k] = s * apk + c * aqk;
      }
      for (int k = 0; k < 3; k++) {
        double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = c * vkp - s * vkq;
        V[3 * k + q] = s * vkp + c * vkq;
      }
    }
  }
  double ev[3] = {A[0], A[4], A[8]};
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++)
      if (fabs(ev[ord[j]]) > fabs(ev[ord[i]])) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int c = 0; c < 3; c++) {
    sv[c] = fabs(ev[ord[c]]);
    for (int r = 0; r < 3; r++) U[3 * r + c] = V[3 * r + ord[c]];
  }
}

/* em_icp.hpp:298-338 (= gicp.hpp:198-235 = semantic_point_cloud.hpp:43-80) */
void orc_cov_from_neighbors(const float* x, const float* y, const float* z, const int* nn,
                            int nn_count, int k, double eps, double cov9[9],
                            double normal[3]) {
  double mean[3] = {0, 0, 0};
  double c00 = 0, c10 = 0, c11 = 0, c20 = 0, c21 = 0, c22 = 0;
  for (int j = 0; j < nn_count; j++) {
    int i = nn[j];
    if (i < 0) continue;
    float px = x[i], py = y[i], pz = z[i];
    mean[0] += px; mean[1] += py; mean[2] += pz;
    /* quirk Q2: float*float products (em_icp.hpp:307-314), summed in double */
    c00 += (float)(px * px);
    c10 += (float)(py * px);
    c11 += (float)(py * py);
    c20 += (float)(pz * px);
    c21 += (float)(pz * py);
    c22 += (float)(pz * pz);
  }
  /* quirk Q3: divide by k regardless of neighbours found (em_icp.hpp:317,320) */
  double kk = (double)k;
  mean[0] /= kk; mean[1] /= kk; mean[2] /= kk;
  double C[9];
  double low[3][3] = {{c00, 0, 0}, {c10, c11, 0}, {c20, c21, c22}};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b <= a; b++) {
      double v = low[a][b];
      v /= kk;
      v -= mean[a] * mean[b];
      C[3 * a + b] = v;
      C[3 * b + a] = v;
    }
  double U[9], sv[3];
  orc_sym3_svd_u(C, U, sv);
  for (int i = 0; i < 9; i++) cov9[i] = 0;
  for (int c = 0; c < 3; c++) {
    double v = (c == 2) ? eps : 1.0;
    double col[3] = {U[c], U[3 + c], U[6 + c]};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov9[3 * a + b] += v * col[a] * col[b];
  }
  if (normal) { normal[0] = U[2]; normal[1] = U[5]; normal[2] = U[8]; }
}

void orc_covariances(int n, const float* x, const float* y, const float* z,
                     const uint32_t* labels, int k, double eps, int C, int use_kdtree,
                     double* cov9, double* normal3, double* hist) {
  int* nn = (int*)malloc(sizeof(int) * (size_t)k * (size_t)(n > 0 ? n : 1));
  float* d2 = (float*)malloc(sizeof(float) * (size_t)k * (size_t)(n > 0 ? n : 1));
  if (use_kdtree) orc_knn_kdtree(n, x, y, z, n, x, y, z, k, nn, d2);
  else orc_knn_brute(n, x, y, z, n, x, y, z, k, nn, d2);
  double increment = 1.0 / (double)k; /* em_icp.hpp:279 */
  for (int i = 0; i < n; i++) {
    const int* row = nn + (size_t)i * k;
    int cnt = 0;
    while (cnt < k && row[cnt] >= 0) cnt++;
    double nrm[3];
    orc_cov_from_neighbors(x, y, z, row, cnt, k, eps, cov9 + (size_t)9 * i, nrm);
    if (normal3) { normal3[3 * (size_t)i] = nrm[0]; normal3[3 * (size_t)i + 1] = nrm[1]; normal3[3 * (size_t)i + 2] = nrm[2]; }
    if (hist && labels) {
      double* h = hist + (size_t)C * i;
      for (int c = 0; c < C; c++) h[c] = 0;
      for (int j = 0; j < cnt; j++) {
        uint32_t l = labels[row[j]];
        /* quirk Q4: labels are 1-based (em_icp.hpp:301); out-of-range labels
         * are undefined behaviour in the reference -- ignored here */
        if (l >= 1 && l <= (uint32_t)C) h[l - 1] += increment;
      }
    }
  }
  free(nn);
  free(d2);
}

/* ------------------------------------------------------------------------- */
/* GICPCostFunction                                                           */
/* ------------------------------------------------------------------------- */

/* gicp_cost_function.h:99-176 */
static void dRtodq(const double dR[9], const double q[4] /*x y z w*/, double out[4] /*x y z w*/) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2], tw = 2 * q[3];
  const double mfx = -2 * tx, mfy = -2 * ty, mfz = -2 * tz, mtw = -1 * tw;
  const double dRdw[9] = {0, -tz, ty, tz, 0, -tx, -ty, tx, 0};
  const double dRdx[9] = {0, ty, tz, ty, mfx, mtw, tz, tw, mfx};
  const double dRdy[9] = {mfy, tx, tw, tx, 0, tz, mtw, tz, mfy};
  const double dRdz[9] = {mfz, mtw, tx, tw, mfz, ty, tx, ty, 0};
  double sw = 0, sx = 0, sy = 0, sz = 0;
  /* (dR^T * dRd?).trace() == sum_ij dR_ij dRd?_ij */
  for (int i = 0; i < 9; i++) {
    sw += dR[i] * dRdw[i];
    sx += dR[i] * dRdx[i];
    sy += dR[i] * dRdy[i];
    sz += dR[i] * dRdz[i];
  }
  out[0] = sx; out[1] = sy; out[2] = sz; out[3] = sw;
}

/* gicp_cost_function.h:27-73 */
void orc_gicp_evaluate(const double qt[7], const double ps[3], const double pt[3],
                       const double Cs[9], const double Ct[9], double* residual,
                       double jac7[7]) {
  double R[9], Rt[9], RC[9], RCRt[9], S[9], M[9];
  orc_se3_rotation(qt, R);                 /* :31 */
  m3_t(R, Rt);
  m3_mul(R, Cs, RC);
  m3_mul(RC, Rt, RCRt);
  m3_add(Ct, RCRt, S);
  m3_inv(S, M);                            /* :32 */
  double tp[3], res[3], dT[3];
  m3_vec(R, ps, tp);
  tp[0] += qt[4]; tp[1] += qt[5]; tp[2] += qt[6];   /* :34 */
  res[0] = pt[0] - tp[0]; res[1] = pt[1] - tp[1]; res[2] = pt[2] - tp[2]; /* :35 */
  m3_vec(M, res, dT);                      /* :36 */
  *residual = res[0] * dT[0] + res[1] * dT[1] + res[2] * dT[2]; /* :37 */
  if (!jac7) return;
  /* :44-55 */
  double Ctt[9], Cst[9], RCst[9], RCstRt[9], S2[9], Ta[9];
  m3_t(Ct, Ctt);
  m3_t(Cs, Cst);
  m3_mul(R, Cst, RCst);
  m3_mul(RCst, Rt, RCstRt);
  m3_add(Ctt, RCstRt, S2);
  m3_inv(S2, Ta);
  double tb[3], tc[3];
  m3_vec(M, res, tb);
  m3_vec(Ta, res, tc);
  /* row vectors res^T*Ta*R*Cs^T and res^T*M*R*Cs */
  double r1[3], r2[3], tmp[3];
  vec_m3(res, Ta, tmp); vec_m3(tmp, R, tmp); vec_m3(tmp, Cst, r1);
  vec_m3(res, M, tmp);  vec_m3(tmp, R, tmp); vec_m3(tmp, Cs, r2);
  double dR[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      dR[3 * i + j] = -(tb[i] * ps[j] + tc[i] * r1[j] + tb[i] * r2[j] + tc[i] * ps[j]);
  dT[0] *= -2.0; dT[1] *= -2.0; dT[2] *= -2.0;   /* :62 */
  double dq[4];
  dRtodq(dR, qt, dq);                              /* :63 */
  jac7[4] = dT[0]; jac7[5] = dT[1]; jac7[6] = dT[2];   /* :64-66 */
  jac7[3] = dq[3]; jac7[0] = dq[0]; jac7[1] = dq[1]; jac7[2] = dq[2]; /* :67-70 */
}

/* Ceres ResidualBlock::Evaluate: J_local = J_ambient(1x7) * PlusJacobian(7x6)
 * with the plus-Jacobian of local_parameterization_se3.h:30-36 */
void orc_gicp_evaluate_local(const double qt[7], const double ps[3], const double pt[3],
                             const double Cs[9], const double Ct[9], double* residual,
                             double jac6[6]) {
  double j7[7], P[42];
  orc_gicp_evaluate(qt, ps, pt, Cs, Ct, residual, jac6 ? j7 : NULL);
  if (!jac6) return;
  orc_se3_dx_this_mul_exp_x_at_0(qt, P);
  for (int c = 0; c < 6; c++) {
    double s = 0;
    for (int r = 0; r < 7; r++) s += j7[r] * P[r * 6 + c];
    jac6[c] = s;
  }
}

/* gicp_cost_function.h:75-87; the return type is bool (quirk Q1) */
int orc_gicp_probability(const double qt[7], const double ps[3], const double pt[3],
                         const double Cs[9], const double Ct[9], double* value) {
  double R[9], Rt[9], RC[9], RCRt[9], cov[9], M[9];
  orc_se3_rotation(qt, R);
  m3_t(R, Rt);
  m3_mul(R, Cs, RC);
  m3_mul(RC, Rt, RCRt);
  m3_add(Ct, RCRt, cov);
  m3_inv(cov, M);
  double tp[3], res[3], dT[3];
  m3_vec(R, ps, tp);
  tp[0] += qt[4]; tp[1] += qt[5]; tp[2] += qt[6];
  res[0] = pt[0] - tp[0]; res[1] = pt[1] - tp[1]; res[2] = pt[2] - tp[2];
  m3_vec(M, res, dT);
  double mahal = -1.0 / 2.0 * (res[0] * dT[0] + res[1] * dT[1] + res[2] * dT[2]);
  double c2[9];
  for (int i = 0; i < 9; i++) c2[i] = 2 * M_PI * cov[i];
  double probability = pow(m3_det(c2), -1.0 / 2.0) * exp(mahal);
  if (value) *value = probability;
  return probability != 0.0; /* double -> bool; NaN -> true */
}

/* ------------------------------------------------------------------------- */
/* losses                                                                     */
/* ------------------------------------------------------------------------- */

/* ceres::CauchyLoss(a) (Ceres loss_function.cc) */
static void cauchy_loss(double a, double s, double rho[3]) {
  double b = a * a, c = 1.0 / b;
  double sum = 1.0 + s * c, inv = 1.0 / sum;
  rho[0] = b * log(sum);
  rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
  rho[2] = -c * (inv * inv);
}
/* sqloss.h:13-18 */
static void sq_loss(double s, double rho[3]) {
  double v = s + DBL_EPSILON;
  rho[0] = sqrt(v);
  rho[1] = 1.0 / (2.0 * sqrt(v));
  rho[2] = -1.0 / (4.0 * pow(v, 1.5));
}

void orc_loss(const orc_params* p, double s, double w, double rho[3]) {
  if (p->use_sqloss) {
    /* ComposedLoss(f = [Scaled](Cauchy(a)), g = SQLoss): em_icp.hpp:109-117,
     * gicp.hpp:98-104 */
    double g[3], f[3];
    sq_loss(s, g);
    cauchy_loss(p->cauchy_a, g[0], f);
    if (p->mode == ORC_MODE_EM) { f[0] *= w; f[1] *= w; f[2] *= w; } /* ScaledLoss */
    rho[0] = f[0];
    rho[1] = f[1] * g[1];
    rho[2] = f[2] * g[1] * g[1] + f[1] * g[2];
  } else {
    /* plain CauchyLoss on r^2: semantic_icp.hpp:96 */
    cauchy_loss(p->cauchy_a, s, rho);
  }
}

/* em_icp.hpp:84-89 */
double orc_em_prob(int C, const double* cm, const double* t_dist, const double* s_dist) {
  double prob = 0;
  for (int s = 0; s < C; s++) {
    double temp = 0, temp2 = 0;
    for (int r = 0; r < C; r++) temp += t_dist[r] * cm[r * C + s];
    for (int r = 0; r < C; r++) temp2 += s_dist[r] * cm[r * C + s];
    temp *= temp2;
    prob += temp;
  }
  return prob;
}

/* ------------------------------------------------------------------------- */
/* one evaluation sweep (what ceres::Evaluator does over all residual blocks)  */
/* ------------------------------------------------------------------------- */
static void accumulate_range(const orc_params* p, const double qt[7], int i0, int i1,
                             const float* sx, const float* sy, const float* sz,
                             const double* scov9, const float* tx, const float* ty,
                             const float* tz, const double* tcov9, int K, const int* idx,
                             const double* w, int want_jac, double out28[28]) {
  for (int i = 0; i < 28; i++) out28[i] = 0;
  for (int i = i0; i < i1; i++) {
    for (int c = 0; c < K; c++) {
      int j = idx[(size_t)i * K + c];
      if (j < 0) continue;
      /* GICPCostFunction ctor converts float points to double
       * (gicp_cost_function.h:21-22) */
      double ps[3] = {sx[i], sy[i], sz[i]}, pt[3] = {tx[j], ty[j], tz[j]};
      double r, J[6], rho[3];
      orc_gicp_evaluate_local(qt, ps, pt, scov9 + 9 * (size_t)i, tcov9 + 9 * (size_t)j, &r,
                              want_jac ? J : NULL);
      double wt = w ? w[(size_t)i * K + c] : 1.0;
      double sq = r * r;
      orc_loss(p, sq, wt, rho);
      out28[27] += 0.5 * rho[0];
      if (!want_jac) continue;
      /* Ceres Corrector (corrector.cc) */
      double sqrt_rho1 = sqrt(rho[1]);
      double rs, alpha_sq_norm;
      if (sq == 0.0 || rho[2] <= 0.0) {
        rs = sqrt_rho1;
        alpha_sq_norm = 0.0;
      } else {
        double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
        double alpha = 1.0 - sqrt(D);
        rs = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
      }
      double Jc[6];
      if (alpha_sq_norm == 0.0) {
        for (int a = 0; a < 6; a++) Jc[a] = sqrt_rho1 * J[a];
      } else {
        for (int a = 0; a < 6; a++) {
          double rtj = J[a] * r;
          Jc[a] = sqrt_rho1 * (J[a] - alpha_sq_norm * r * rtj);
        }
      }
      double rc = r * rs;
      int o = 0;
      for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) out28[o++] += Jc[a] * Jc[b];
      for (int a = 0; a < 6; a++) out28[21 + a] += Jc[a] * rc;
    }
  }
}

static void accumulate_mt(const orc_params* p, const double qt[7], int n_s, const float* sx,
                          const float* sy, const float* sz, const double* scov9,
                          const float* tx, const float* ty, const float* tz,
                          const double* tcov9, int K, const int* idx, const double* w,
                          int want_jac, double out28[28]) {
  int nt = p->num_threads > 0 ? p->num_threads : 1;
  if (nt > 64) nt = 64;
  if (n_s < 4 * nt) nt = 1;
  double part[64][28];
  int chunk = (n_s + nt - 1) / nt;
#pragma omp parallel for num_threads(nt) schedule(static, 1)
  for (int t = 0; t < nt; t++) {
    int i0 = t * chunk, i1 = i0 + chunk;
    if (i1 > n_s) i1 = n_s;
    if (i0 > n_s) i0 = n_s;
    accumulate_range(p, qt, i0, i1, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, want_jac,
                     part[t]);
  }
  for (int i = 0; i < 28; i++) out28[i] = 0;
  for (int t = 0; t < nt; t++)
    for (int i = 0; i < 28; i++) out28[i] += part[t][i];
}

void orc_accumulate(const orc_params* p, const double qt[7], int n_s, const float* sx,
                    const float* sy, const float* sz, const double* scov9, const float* tx,
                    const float* ty, const float* tz, const double* tcov9, int K,
                    const int* idx, const double* w, double out28[28]) {
  accumulate_mt(p, qt, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, 1, out28);
}

/* ------------------------------------------------------------------------- */
/* Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy on one 6-DoF block  */
/* (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc; options at      */
/* em_icp.hpp:162-172)                                                         */
/* ------------------------------------------------------------------------- */
static double norm7(const double* a) {
  double s = 0;
  for (int i = 0; i < 7; i++) s += a[i] * a[i];
  return sqrt(s);
}

/* solve (A) y = b for symmetric positive definite 6x6 A (full storage) */
static int chol6_solve(const double A[36], const double b[6], double y[6]) {
  double L[36];
  memset(L, 0, sizeof L);
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j <= i; j++) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k];
      if (i == j) {
        if (!(s > 0)) return -1;
        L[6 * i + i] = sqrt(s);
      } else {
        L[6 * i + j] = s / L[6 * j + j];
      }
    }
  }
  double z[6];
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[6 * i + k] * z[k];
    z[i] = s / L[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = z[i];
    for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * y[k];
    y[i] = s / L[6 * i + i];
  }
  return 0;
}

static void unpack28(const double o[28], double H[36], double g[6], double* cost) {
  int k = 0;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++) { H[6 * a + b] = o[k]; H[6 * b + a] = o[k]; k++; }
  for (int a = 0; a < 6; a++) g[a] = o[21 + a];
  *cost = o[27];
}

static double gradient_max_norm(const double x[7], const double g[6]) {
  double ng[6], xp[7], m = 0;
  for (int i = 0; i < 6; i++) ng[i] = -g[i];
  orc_se3_plus(x, ng, xp);
  for (int i = 0; i < 7; i++) { double d = fabs(x[i] - xp[i]); if (d > m) m = d; }
  return m;
}

/* trace of the trust-region loop, one record per step attempt (tests compare it with an independently
 * written loop, tests/lm_ref.py): cost at the accepted iterate before the step, radius used for the
 * step, candidate cost, and whether the step was accepted (1), rejected (0) or invalid (-1) */
typedef struct { int max, n; double *cost, *radius, *cand_cost; int* accepted; } lm_trace;

static int solve_impl(const orc_params* p, int n_s, const float* sx, const float* sy, const float* sz,
              const double* scov9, const float* tx, const float* ty, const float* tz,
              const double* tcov9, int K, const int* idx, const double* w,
              const double init_qt[7], double out_qt[7], int* lm_iters, int* evals,
              double* final_cost, lm_trace* tr) {
  double x[7], o[28], H[36], g[6], cost;
  memcpy(x, init_qt, sizeof x);
  int n_eval = 0, iter = 0, status = 0;
  accumulate_mt(p, x, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, 1, o);
  n_eval++;
  /* Ceres: ResidualBlock::Evaluate -> IsArrayValid; "Initial residual and Jacobian evaluation failed": FAILURE,
   * the parameters stay where they were */
  for (int k = 0; k < 28; k++)
    if (!isfinite(o[k])) {
      memcpy(out_qt, x, sizeof x);
      if (lm_iters) *lm_iters = 0;
      if (evals) *evals = n_eval;
      if (final_cost) *final_cost = o[27];
      return 3;
    }
  unpack28(o, H, g, &cost);
  double x_norm = norm7(x);
  double scale[6];
  for (int j = 0; j < 6; j++) scale[j] = p->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[6 * j + j])) : 1.0;
  double radius = p->initial_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, invalid = 0;
  double diag[6] = {0, 0, 0, 0, 0, 0};
  for (;;) {
    if (iter >= p->max_lm_iterations) { status = 1; break; }
    if (gradient_max_norm(x, g) <= p->gradient_tolerance) break;
    if (radius <= p->min_radius) break;
    iter++;
    double Hs[36], gs[6];
    for (int a = 0; a < 6; a++) {
      gs[a] = g[a] * scale[a];
      for (int b = 0; b < 6; b++) Hs[6 * a + b] = H[6 * a + b] * scale[a] * scale[b];
    }
    if (!reuse_diagonal)
      for (int j = 0; j < 6; j++) {
        double d = Hs[6 * j + j];
        if (d < p->min_lm_diagonal) d = p->min_lm_diagonal;
        if (d > p->max_lm_diagonal) d = p->max_lm_diagonal;
        diag[j] = d;
      }
    double A[36], y[6], step[6];
    memcpy(A, Hs, sizeof A);
    for (int j = 0; j < 6; j++) {
      double lm = sqrt(diag[j] / radius); /* lm_diagonal_ */
      A[6 * j + j] += lm * lm;
    }
    reuse_diagonal = 1;
    int ok = chol6_solve(A, gs, y) == 0;
    double model_change = 0;
    if (ok) {
      for (int j = 0; j < 6; j++) step[j] = -y[j];
      double sg = 0, sHs = 0;
      for (int a = 0; a < 6; a++) {
        sg += step[a] * gs[a];
        double r = 0;
        for (int b = 0; b < 6; b++) r += Hs[6 * a + b] * step[b];
        sHs += step[a] * r;
      }
      model_change = -(sg + 0.5 * sHs);
    }
    if (!ok || !(model_change > 0.0)) {
      if (tr && tr->n < tr->max) { tr->cost[tr->n] = cost; tr->radius[tr->n] = radius; tr->cand_cost[tr->n] = cost; tr->accepted[tr->n] = -1; tr->n++; }
      if (++invalid >= p->max_consecutive_invalid_steps) { status = 2; break; }
      radius *= 0.5;
      reuse_diagonal = 1;
      continue;
    }
    invalid = 0;
    double delta[6], cand[7], oc[28];
    for (int j = 0; j < 6; j++) delta[j] = step[j] * scale[j];
    orc_se3_plus(x, delta, cand);
    accumulate_mt(p, cand, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, 0, oc);
    n_eval++;
    /* trust_region_minimizer.cc: "Step failed to evaluate. Treating it as a step with infinite cost" */
    double cand_cost = isfinite(oc[27]) ? oc[27] : DBL_MAX;
    if (tr && tr->n < tr->max) { tr->cost[tr->n] = cost; tr->radius[tr->n] = radius; tr->cand_cost[tr->n] = cand_cost; tr->accepted[tr->n] = 0; tr->n++; }
    double diff[7];
    for (int i = 0; i < 7; i++) diff[i] = x[i] - cand[i];
    double step_norm = norm7(diff);
    if (step_norm <= p->parameter_tolerance * (x_norm + p->parameter_tolerance)) break;
    double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= p->function_tolerance * cost) break;
    double rel = cost_change / model_change;
    if (rel > p->min_relative_decrease) {
      if (tr && tr->n > 0) tr->accepted[tr->n - 1] = 1;
      memcpy(x, cand, sizeof x);
      x_norm = norm7(x);
      accumulate_mt(p, x, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, 1, o);
      n_eval++;
      unpack28(o, H, g, &cost);
      double t = 2.0 * rel - 1.0;
      double den = 1.0 - t * t * t;
      if (den < 1.0 / 3.0) den = 1.0 / 3.0;
      radius = radius / den;
      if (radius > p->max_radius) radius = p->max_radius;
      decrease_factor = 2.0;
      reuse_diagonal = 0;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
    }
  }
  memcpy(out_qt, x, sizeof x);
  if (lm_iters) *lm_iters = iter;
  if (evals) *evals = n_eval;
  if (final_cost) *final_cost = cost;
  return status;
}

int orc_solve(const orc_params* p, int n_s, const float* sx, const float* sy, const float* sz,
              const double* scov9, const float* tx, const float* ty, const float* tz,
              const double* tcov9, int K, const int* idx, const double* w,
              const double init_qt[7], double out_qt[7], int* lm_iters, int* evals,
              double* final_cost) {
  return solve_impl(p, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, init_qt, out_qt, lm_iters, evals, final_cost, NULL);
}

int orc_solve_trace(const orc_params* p, int n_s, const float* sx, const float* sy, const float* sz,
                    const double* scov9, const float* tx, const float* ty, const float* tz,
                    const double* tcov9, int K, const int* idx, const double* w,
                    const double init_qt[7], double out_qt[7], int max_trace, double* trace_cost,
                    double* trace_radius, double* trace_cand_cost, int* trace_accepted, int* n_trace) {
  lm_trace tr = {max_trace, 0, trace_cost, trace_radius, trace_cand_cost, trace_accepted};
  int it = 0, ev = 0;
  double fc = 0;
  int st = solve_impl(p, n_s, sx, sy, sz, scov9, tx, ty, tz, tcov9, K, idx, w, init_qt, out_qt, &it, &ev, &fc, &tr);
  *n_trace = tr.n;
  return st;
}

/* ------------------------------------------------------------------------- */
/* params                                                                     */
/* ------------------------------------------------------------------------- */
void orc_default_params(int mode, orc_params* p) {
  memset(p, 0, sizeof *p);
  p->mode = mode;
  p->k_cov = 20;            /* em_icp.h:42, gicp.h:34, semantic_point_cloud.h:31 */
  p->epsilon = 0.001;       /* em_icp.h:43 */
  p->gate_sq = 250.0;       /* em_icp.hpp:65, gicp.hpp:70, semantic_icp.hpp:69 */
  p->min_class_pts = 400;   /* semantic_icp.hpp:51 */
  p->num_classes = 0;
  p->gradient_tolerance = 0.1 * SOPHUS_EPS; /* em_icp.hpp:163 */
  p->function_tolerance = 0.1 * SOPHUS_EPS; /* em_icp.hpp:164 */
  p->max_lm_iterations = 400;               /* em_icp.hpp:169 */
  p->parameter_tolerance = 1e-8;
  p->initial_radius = 1e4;
  p->max_radius = 1e16;
  p->min_radius = 1e-32;
  p->min_relative_decrease = 1e-3;
  p->min_lm_diagonal = 1e-6;
  p->max_lm_diagonal = 1e32;
  p->max_consecutive_invalid_steps = 5;
  p->jacobi_scaling = 1;
  p->use_kdtree = 1;
  if (mode == ORC_MODE_EM) {
    p->knn = 4; p->cauchy_a = 3.0; p->use_sqloss = 1;     /* em_icp.hpp:60,111,115 */
    p->outer_tol = 1e-5; p->max_outer = 50;               /* em_icp.hpp:180 */
    p->num_threads = 8;                                   /* em_icp.hpp:166 */
  } else if (mode == ORC_MODE_GICP) {
    p->knn = 1; p->cauchy_a = 3.0; p->use_sqloss = 1;     /* gicp.hpp:69,100,102 */
    p->outer_tol = 1e-5; p->max_outer = 50;               /* gicp.hpp:154 */
    p->num_threads = 8;                                   /* gicp.hpp:142 */
  } else {
    p->knn = 1; p->cauchy_a = 1.5; p->use_sqloss = 0;     /* semantic_icp.hpp:68,96 */
    p->outer_tol = 0.001; p->max_outer = 35;              /* semantic_icp.hpp:152 */
    p->num_threads = 4;                                   /* semantic_icp.hpp:140 */
  }
}

/* ------------------------------------------------------------------------- */
/* the three align() bodies                                                   */
/* ------------------------------------------------------------------------- */

typedef struct {
  int n;
  float *x, *y, *z;
  uint32_t* l;
  int n_seg;        /* label segments (semantic mode) */
  uint32_t* seg_label;
  int* seg_off;     /* n_seg+1 */
  double* cov9;
  double* hist;
} cloud_t;

static void cloud_free(cloud_t* c) {
  free(c->x); free(c->y); free(c->z); free(c->l);
  free(c->seg_label); free(c->seg_off); free(c->cov9); free(c->hist);
  memset(c, 0, sizeof *c);
}

/* copy (flat) or group by label in first-seen order (pcl_2_semantic.h:24-39) */
static void cloud_init(cloud_t* c, int n, const float* x, const float* y, const float* z,
                       const uint32_t* l, int group) {
  memset(c, 0, sizeof *c);
  c->n = n;
  size_t m = (size_t)(n > 0 ? n : 1);
  c->x = (float*)malloc(sizeof(float) * m);
  c->y = (float*)malloc(sizeof(float) * m);
  c->z = (float*)malloc(sizeof(float) * m);
  c->l = (uint32_t*)calloc(m, sizeof(uint32_t));
  if (!group) {
    memcpy(c->x, x, sizeof(float) * (size_t)n);
    memcpy(c->y, y, sizeof(float) * (size_t)n);
    memcpy(c->z, z, sizeof(float) * (size_t)n);
    if (l) memcpy(c->l, l, sizeof(uint32_t) * (size_t)n);
    c->n_seg = 1;
    c->seg_label = (uint32_t*)calloc(1, sizeof(uint32_t));
    c->seg_off = (int*)malloc(2 * sizeof(int));
    c->seg_off[0] = 0; c->seg_off[1] = n;
    return;
  }
  uint32_t* labs = (uint32_t*)malloc(sizeof(uint32_t) * m);
  int* cnt = (int*)calloc(m, sizeof(int));
  int* which = (int*)malloc(sizeof(int) * m);
  int ns = 0;
  for (int i = 0; i < n; i++) {
    int s = -1;
    for (int k = 0; k < ns; k++) if (labs[k] == l[i]) { s = k; break; }
    if (s < 0) { s = ns++; labs[s] = l[i]; }
    which[i] = s;
    cnt[s]++;
  }
  c->n_seg = ns;
  c->seg_label = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(ns > 0 ? ns : 1));
  c->seg_off = (int*)malloc(sizeof(int) * (size_t)(ns + 1));
  c->seg_off[0] = 0;
  for (int k = 0; k < ns; k++) { c->seg_label[k] = labs[k]; c->seg_off[k + 1] = c->seg_off[k] + cnt[k]; }
  int* fill = (int*)malloc(sizeof(int) * (size_t)(ns > 0 ? ns : 1));
  for (int k = 0; k < ns; k++) fill[k] = c->seg_off[k];
  for (int i = 0; i < n; i++) {
    int d = fill[which[i]]++;
    c->x[d] = x[i]; c->y[d] = y[i]; c->z[d] = z[i]; c->l[d] = l[i];
  }
  free(labs); free(cnt); free(which); free(fill);
}

static void cloud_covariances(const orc_params* p, cloud_t* c, int with_hist) {
  size_t m = (size_t)(c->n > 0 ? c->n : 1);
  c->cov9 = (double*)malloc(sizeof(double) * 9 * m);
  if (with_hist) c->hist = (double*)malloc(sizeof(double) * (size_t)p->num_classes * m);
  for (int s = 0; s < c->n_seg; s++) {
    int o = c->seg_off[s], n = c->seg_off[s + 1] - o;
    orc_covariances(n, c->x + o, c->y + o, c->z + o, with_hist ? c->l + o : NULL, p->k_cov,
                    p->epsilon, p->num_classes, p->use_kdtree, c->cov9 + 9 * (size_t)o, NULL,
                    with_hist ? c->hist + (size_t)p->num_classes * o : NULL);
  }
}

static int find_seg(const cloud_t* c, uint32_t label) {
  for (int k = 0; k < c->n_seg; k++) if (c->seg_label[k] == label) return k;
  return -1;
}

int orc_align(const orc_params* p, int n_s, const float* sx, const float* sy,
              const float* sz, const uint32_t* sl, int n_t, const float* tx,
              const float* ty, const float* tz, const uint32_t* tl, const double* cm,
              const double init_qt[7], double out_qt[7], orc_stats* stats) {
  const int K = p->knn;
  const int em = p->mode == ORC_MODE_EM, sem = p->mode == ORC_MODE_SEMANTIC;
  if (K < 1 || K > 32) return -1;
  if ((em || sem) && (!sl || !tl)) return -2;
  if (em && (!cm || p->num_classes < 1)) return -3;
  if (!sem && n_t < K) return -4; /* quirk Q5: reference reads past the result vectors */
  orc_stats st;
  memset(&st, 0, sizeof st);
  double t_begin = now_s();

  cloud_t S, T;
  cloud_init(&S, n_s, sx, sy, sz, sl, sem);
  cloud_init(&T, n_t, tx, ty, tz, tl, sem);
  /* em_icp.hpp:28-29 / gicp.hpp:33-34; for SemanticICP the covariances were
   * computed per label at cloud construction (semantic_point_cloud.hpp:25-84) */
  double t0 = now_s();
  cloud_covariances(p, &S, em);
  cloud_covariances(p, &T, em);
  st.t_cov_s = now_s() - t0;

  size_t ms = (size_t)(n_s > 0 ? n_s : 1);
  float* qx = (float*)malloc(sizeof(float) * ms);
  float* qy = (float*)malloc(sizeof(float) * ms);
  float* qz = (float*)malloc(sizeof(float) * ms);
  int* idx = (int*)malloc(sizeof(int) * ms * (size_t)K);
  float* d2 = (float*)malloc(sizeof(float) * ms * (size_t)K);
  double* w = (double*)malloc(sizeof(double) * ms * (size_t)K);
  int* seg_map = (int*)malloc(sizeof(int) * (size_t)(S.n_seg > 0 ? S.n_seg : 1));
  kd_tree** trees = (kd_tree**)calloc((size_t)(T.n_seg > 0 ? T.n_seg : 1), sizeof(kd_tree*));
  if (p->use_kdtree) {
    /* setTargetCloud builds the tree once (em_icp.h:59-66) */
    for (int s = 0; s < T.n_seg; s++) {
      int o = T.seg_off[s];
      trees[s] = kd_build(T.seg_off[s + 1] - o, T.x + o, T.y + o, T.z + o);
    }
  }
  for (int s = 0; s < S.n_seg; s++) seg_map[s] = sem ? find_seg(&T, S.seg_label[s]) : 0;

  double cur[7], est[7];
  memcpy(cur, init_qt, sizeof cur);
  int converged = 0, outer = 0, count = 0;
  while (!converged) {
    memcpy(est, cur, sizeof est);
    if (sem) count++; /* semantic_icp.hpp:47 */
    double M[16];
    orc_se3_matrix(cur, M);
    /* --- correspondences ------------------------------------------------ */
    t0 = now_s();
    for (size_t i = 0; i < ms * (size_t)K; i++) { idx[i] = -1; w[i] = 0; }
    int64_t n_active = 0;
    for (int s = 0; s < S.n_seg; s++) {
      int so = S.seg_off[s], sn = S.seg_off[s + 1] - so;
      int ts = seg_map[s];
      if (sem) {
        if (ts < 0) continue;                 /* semantic_icp.hpp:50 */
        if (!(sn > p->min_class_pts)) continue; /* semantic_icp.hpp:51 */
      }
      int to = T.seg_off[ts], tn = T.seg_off[ts + 1] - to;
      orc_transform_points(M, sn, S.x + so, S.y + so, S.z + so, qx + so, qy + so, qz + so);
      if (p->use_kdtree) {
        for (int i = 0; i < sn; i++)
          kd_knn(trees[ts], qx[so + i], qy[so + i], qz[so + i], K, idx + (size_t)(so + i) * K,
                 d2 + (size_t)(so + i) * K);
      } else {
        orc_knn_brute(sn, qx + so, qy + so, qz + so, tn, T.x + to, T.y + to, T.z + to, K,
                      idx + (size_t)so * K, d2 + (size_t)so * K);
      }
      st.total_corr += (int64_t)sn * K;
      for (int i = so; i < so + sn; i++)
        for (int c = 0; c < K; c++) {
          size_t e = (size_t)i * K + c;
          if (idx[e] >= 0 && d2[e] < (float)p->gate_sq) {  /* strict <, float compare */
            idx[e] += to;
            n_active++;
          } else {
            idx[e] = -1;
          }
        }
    }
    st.t_nn_s += now_s() - t0;
    /* --- weights (EM only) ----------------------------------------------- */
    t0 = now_s();
    for (int i = 0; i < n_s; i++)
      for (int c = 0; c < K; c++) {
        size_t e = (size_t)i * K + c;
        int j = idx[e];
        if (j < 0) continue;
        if (em) {
          double prob = orc_em_prob(p->num_classes, cm, T.hist + (size_t)p->num_classes * j,
                                    S.hist + (size_t)p->num_classes * i);
          double ps[3] = {S.x[i], S.y[i], S.z[i]}, pt[3] = {T.x[j], T.y[j], T.z[j]};
          /* em_icp.hpp:108: prob *= bool */
          prob *= (double)orc_gicp_probability(est, ps, pt, S.cov9 + 9 * (size_t)i,
                                               T.cov9 + 9 * (size_t)j, NULL);
          w[e] = prob;
        } else {
          w[e] = 1.0;
        }
      }
    st.t_weight_s += now_s() - t0;
    st.total_active += n_active;
    /* --- inner solve ------------------------------------------------------ */
    t0 = now_s();
    if (n_active > 0) {
      int it = 0, ev = 0;
      double fc = 0;
      orc_solve(p, n_s, S.x, S.y, S.z, S.cov9, T.x, T.y, T.z, T.cov9, K, idx, w, est, est, &it,
                &ev, &fc);
      st.total_lm_iters += it;
      st.total_evals += ev;
      st.final_cost = fc;
    }
    st.t_solve_s += now_s() - t0;
    /* --- convergence (em_icp.hpp:179-187 / semantic_icp.hpp:151-158) ------- */
    double inv[7], rel[7], lg[6];
    orc_se3_inv(cur, inv);
    orc_se3_mul(inv, est, rel);
    orc_se3_log(rel, lg);
    double mse = 0;
    for (int i = 0; i < 6; i++) mse += lg[i] * lg[i];
    if (sem) {
      if (mse < p->outer_tol || count > p->max_outer) converged = 1;
      memcpy(cur, est, sizeof cur);
    } else {
      if (mse < p->outer_tol || outer > p->max_outer) converged = 1;
      memcpy(cur, est, sizeof cur);
      outer++;
    }
  }
  memcpy(out_qt, cur, sizeof cur);
  st.outer_iters = sem ? count : outer;
  st.t_total_s = now_s() - t_begin;
  if (stats) *stats = st;
  for (int s = 0; s < T.n_seg; s++) kd_free(trees[s]);
  free(trees); free(seg_map); free(qx); free(qy); free(qz); free(idx); free(d2); free(w);
  cloud_free(&S);
  cloud_free(&T);
  return 0;
}

/* em_icp.hpp:202-268 */
int orc_fused_labels(const orc_params* p, int n_s, const float* sx, const float* sy,
                     const float* sz, const uint32_t* sl, int n_t, const float* tx,
                     const float* ty, const float* tz, const uint32_t* tl, const double* cm,
                     const double qt[7], uint32_t* out_labels) {
  const int K = 4, C = p->num_classes; /* em_icp.hpp:221 */
  if (n_t < K || C < 1 || !cm || !sl || !tl) return -1;
  cloud_t S, T;
  cloud_init(&S, n_s, sx, sy, sz, sl, 0);
  cloud_init(&T, n_t, tx, ty, tz, tl, 0);
  /* getFusedLabels reads the covariances/distributions align() left behind */
  cloud_covariances(p, &S, 1);
  cloud_covariances(p, &T, 1);
  size_t ms = (size_t)(n_s > 0 ? n_s : 1);
  float* qx = (float*)malloc(sizeof(float) * ms);
  float* qy = (float*)malloc(sizeof(float) * ms);
  float* qz = (float*)malloc(sizeof(float) * ms);
  int* idx = (int*)malloc(sizeof(int) * ms * K);
  float* d2 = (float*)malloc(sizeof(float) * ms * K);
  double M[16];
  orc_se3_matrix(qt, M);
  orc_transform_points(M, n_s, S.x, S.y, S.z, qx, qy, qz);
  if (p->use_kdtree) orc_knn_kdtree(n_s, qx, qy, qz, n_t, T.x, T.y, T.z, K, idx, d2);
  else orc_knn_brute(n_s, qx, qy, qz, n_t, T.x, T.y, T.z, K, idx, d2);
  double* sprob = (double*)malloc(sizeof(double) * (size_t)C);
  for (int i = 0; i < n_s; i++) {
    for (int s = 0; s < C; s++) sprob[s] = 0;
    for (int c = 0; c < K; c++) {
      size_t e = (size_t)i * K + c;
      if (!(d2[e] < (float)p->gate_sq)) continue; /* :228 */
      int j = idx[e];
      double ps[3] = {S.x[i], S.y[i], S.z[i]}, pt[3] = {T.x[j], T.y[j], T.z[j]};
      double prob = (double)orc_gicp_probability(qt, ps, pt, S.cov9 + 9 * (size_t)i,
                                                 T.cov9 + 9 * (size_t)j, NULL); /* :248 */
      const double* td = T.hist + (size_t)C * j;
      const double* sd = S.hist + (size_t)C * i;
      for (int s = 0; s < C; s++) { /* :249-253 */
        double temp = 0, temp2 = 0;
        for (int r = 0; r < C; r++) temp += td[r] * cm[r * C + s];
        for (int r = 0; r < C; r++) temp2 += sd[r] * cm[r * C + s];
        temp *= temp2;
        sprob[s] += temp * prob;
      }
    }
    double max_prob = 0;
    int max_s = 0;
    for (int s = 0; s < C; s++)
      if (sprob[s] > max_prob) { max_s = s; max_prob = sprob[s]; } /* :256-263 */
    out_labels[i] = (uint32_t)(max_s + 1);                         /* :265 */
  }
  free(sprob); free(qx); free(qy); free(qz); free(idx); free(d2);
  cloud_free(&S);
  cloud_free(&T);
  return 0;
}
