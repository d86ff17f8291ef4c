/*
 * sicp_oracle.h -- CPU restatement ("oracle") of the kxhit/semantic-icp hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the reference is header-only C++ on PCL/FLANN + Ceres +
 * Sophus + Eigen, none of which exist in this image, and it ships no golden
 * vectors or known-answer tests.  This file restates
 *   - semantic_icp/impl/em_icp.hpp, impl/gicp.hpp, impl/semantic_icp.hpp:28-166,
 *     impl/semantic_point_cloud.hpp:12-86, gicp_cost_function.h,
 *     local_parameterization_se3.h, sqloss.h, pcl_2_semantic.h
 * line by line, and restates the published algorithms of the absent
 * third-party pieces (FLANN exact float kNN, pcl::transformPointCloud,
 * Sophus SE3 exp/log/Dx_this_mul_exp_x_at_0, Eigen JacobiSVD/inverse,
 * Ceres 1.14..2.1 trust-region LM + CauchyLoss/ScaledLoss/ComposedLoss +
 * Corrector).  It is pinned instead by tests/golden/ (independent numpy /
 * scipy restatement, finite differences, scipy.linalg.expm/logm, cKDTree,
 * LAPACK SVD) -- see tests/golden/make_golden.py.
 *
 * Pose exchange format everywhere: Sophus storage order
 *   qt[7] = [qx qy qz qw tx ty tz]      (gicp_cost_function.h:64-70)
 * Tangent order: [upsilon(3); omega(3)], right perturbation T*exp(delta)
 *   (local_parameterization_se3.h:17-25).
 */
#ifndef SICP_ORACLE_H_
#define SICP_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- modes (which reference class is being restated) ------------------- */
enum {
  ORC_MODE_GICP = 0,     /* semanticicp::GICP<PointT>            impl/gicp.hpp        */
  ORC_MODE_EM = 1,       /* EmIterativeClosestPoint<N>           impl/em_icp.hpp      */
  ORC_MODE_SEMANTIC = 2  /* SemanticIterativeClosestPoint        impl/semantic_icp.hpp*/
};

typedef struct {
  int mode;
  int knn;              /* correspondences per source point: 4 (EM, em_icp.hpp:60) or 1 */
  int k_cov;            /* covariance neighbourhood, default 20 (em_icp.h:42)            */
  double epsilon;       /* GICP epsilon, default 1e-3 (em_icp.h:43)                       */
  double gate_sq;       /* 250, strict < (em_icp.hpp:65)                                  */
  double cauchy_a;      /* 3.0 (em_icp.hpp:111, gicp.hpp:100) / 1.5 (semantic_icp.hpp:96) */
  int use_sqloss;       /* 1 for EM/GICP (Composed(.., SQLoss)), 0 for Semantic           */
  double outer_tol;     /* 1e-5 (em_icp.hpp:180) / 1e-3 (semantic_icp.hpp:152)            */
  int max_outer;        /* index threshold: 50 (em/gicp) / 35 (semantic)                  */
  int min_class_pts;    /* 400, strict > (semantic_icp.hpp:51)                            */
  int num_classes;      /* runtime replacement of template parameter N (em_icp.h:16)      */
  /* Ceres options set by the reference (em_icp.hpp:162-172) */
  double gradient_tolerance;   /* 0.1 * Sophus eps = 1e-11 */
  double function_tolerance;   /* 1e-11 */
  int max_lm_iterations;       /* 400 */
  /* Ceres defaults (not in tree; Ceres 1.14..2.1 solver.h) */
  double parameter_tolerance;        /* 1e-8 */
  double initial_radius;             /* 1e4  */
  double max_radius;                 /* 1e16 */
  double min_radius;                 /* 1e-32 */
  double min_relative_decrease;      /* 1e-3 */
  double min_lm_diagonal;            /* 1e-6 */
  double max_lm_diagonal;            /* 1e32 */
  int max_consecutive_invalid_steps; /* 5 */
  int jacobi_scaling;                /* 1 */
  int num_threads;      /* residual evaluation threads (reference: 8 / 8 / 4) */
  int use_kdtree;       /* 1: exact kd-tree kNN (CPU baseline), 0: brute force; same results */
} orc_params;

typedef struct {
  int outer_iters;          /* value the reference stores in outer_iter / count            */
  int total_lm_iters;       /* LM iterations summed over outer passes                      */
  int total_evals;          /* residual(+jacobian) sweeps (E of SURVEY 8d, summed)         */
  int64_t total_corr;       /* sum over outer passes of candidate correspondence slots     */
  int64_t total_active;     /* slots that passed the distance gate                         */
  double final_cost;
  double t_cov_s, t_nn_s, t_weight_s, t_solve_s, t_total_s;
} orc_stats;

void orc_default_params(int mode, orc_params* p);

/* ---- Sophus SE3d restatement (not in tree; Sophus 1.0 se3.hpp / so3.hpp) - */
void orc_se3_exp(const double a[6], double qt[7]);
void orc_se3_log(const double qt[7], double a[6]);
void orc_se3_mul(const double a[7], const double b[7], double out[7]);
void orc_se3_inv(const double a[7], double out[7]);
void orc_se3_rotation(const double qt[7], double R[9]);   /* row-major, Eigen toRotationMatrix */
void orc_se3_matrix(const double qt[7], double M[16]);    /* row-major 4x4 */
void orc_se3_plus(const double qt[7], const double delta[6], double out[7]);
void orc_se3_dx_this_mul_exp_x_at_0(const double qt[7], double J[42]); /* 7x6 row-major */

/* ---- pcl::transformPointCloud (double matrix, float points) ------------- */
void orc_transform_points(const double M[16], int n, const float* x, const float* y,
                          const float* z, float* ox, float* oy, float* oz);

/* ---- exact float32 kNN (FLANN L2_Simple<float>, ascending, ties->lowest index)
 * idx/d2 are [nq*k]; when nt < k the tail is idx=-1, d2=+inf. */
void orc_knn_brute(int nq, const float* qx, const float* qy, const float* qz, int nt,
                   const float* tx, const float* ty, const float* tz, int k, int* idx,
                   float* d2);
void orc_knn_kdtree(int nq, const float* qx, const float* qy, const float* qz, int nt,
                    const float* tx, const float* ty, const float* tz, int k, int* idx,
                    float* d2);

/* ---- per-point covariances + label histograms (em_icp.hpp:270-343) ------
 * cov9: n*9 row-major, normal3: n*3 (U[:,2]), hist: n*C doubles (nullable when
 * labels == NULL). labels are 1-based (quirk Q4). */
void orc_covariances(int n, const float* x, const float* y, const float* z,
                     const uint32_t* labels, int k, double eps, int C, int use_kdtree,
                     double* cov9, double* normal3, double* hist);
/* helper: covariance of one neighbourhood given neighbour index list */
void orc_cov_from_neighbors(const float* x, const float* y, const float* z, const int* nn,
                            int nn_count, int k, double eps, double cov9[9],
                            double normal[3]);
/* symmetric 3x3 -> U (columns sorted by descending |eigenvalue|, as JacobiSVD's U) */
void orc_sym3_svd_u(const double A[9], double U[9], double sv[3]);

/* ---- GICPCostFunction (gicp_cost_function.h) ---------------------------- */
/* literal restatement of Evaluate :27-73 ; jac7 may be NULL */
void orc_gicp_evaluate(const double qt[7], const double ps[3], const double pt[3],
                       const double Cs[9], const double Ct[9], double* residual,
                       double jac7[7]);
/* Evaluate followed by Ceres' multiply with the 7x6 plus-Jacobian */
void orc_gicp_evaluate_local(const double qt[7], const double ps[3], const double pt[3],
                             const double Cs[9], const double Ct[9], double* residual,
                             double jac6[6]);
/* Probability :75-87 -- returns the bool the reference returns (quirk Q1);
 * *value (nullable) receives the double before the bool conversion */
int orc_gicp_probability(const double qt[7], const double ps[3], const double pt[3],
                         const double Cs[9], const double Ct[9], double* value);

/* ---- losses (sqloss.h, Ceres loss_function.cc) --------------------------- */
/* rho[3] for the composition used by `mode`, at s = residual^2, scaled by w */
void orc_loss(const orc_params* p, double s, double w, double rho[3]);

/* ---- EM label weight (em_icp.hpp:84-89) ---------------------------------- */
double orc_em_prob(int C, const double* cm_rowmajor, const double* t_dist,
                   const double* s_dist);

/* ---- one evaluation sweep: out28 = [H upper 21 | g 6 | cost] ---------------
 * H = sum rho1 J J^T, g = sum rho1 r J, cost = 1/2 sum rho0   (Ceres Corrector,
 * rho2 <= 0 branch).  idx[n_s*K] (-1 = dropped), w[n_s*K]. */
void orc_accumulate(const orc_params* p, const double qt[7], int n_s, const float* sx,
                    const float* sy, const float* sz, const double* scov9, const float* tx,
                    const float* ty, const float* tz, const double* tcov9, int K,
                    const int* idx, const double* w, double out28[28]);

/* ---- full registration (the three align() bodies) ------------------------
 * labels: 1..C for EM, arbitrary for SEMANTIC, ignored (may be NULL) for GICP.
 * cm: C*C row-major confusion matrix (EM only).  Returns 0 on success. */
int orc_align(const orc_params* p, int n_s, const float* sx, const float* sy,
              const float* sz, const uint32_t* sl, int n_t, const float* tx,
              const float* ty, const float* tz, const uint32_t* tl, const double* cm,
              const double init_qt[7], double out_qt[7], orc_stats* stats);

/* getFusedLabels (em_icp.hpp:202-268): out_labels[n_s] */
int orc_fused_labels(const orc_params* p, int n_s, const float* sx, const float* sy,
                     const float* sz, const uint32_t* sl, int n_t, const float* tx,
                     const float* ty, const float* tz, const uint32_t* tl, const double* cm,
                     const double qt[7], uint32_t* out_labels);

/* inner solve only, given fixed correspondences (used by tests to compare the
 * LM against an independent minimiser) */
int orc_solve(const orc_params* p, int n_s, const float* sx, const float* sy, const float* sz,
              const double* scov9, const float* tx, const float* ty, const float* tz,
              const double* tcov9, int K, const int* idx, const double* w,
              const double init_qt[7], double out_qt[7], int* lm_iters, int* evals,
              double* final_cost);
/* the same solve, recording every step attempt: cost before the step, trust-region radius of the
 * step, candidate cost, accepted (1) / rejected (0) / invalid (-1) */
int orc_solve_trace(const orc_params* p, int n_s, const float* sx, const float* sy, const float* sz,
                    const double* scov9, const float* tx, const float* ty, const float* tz,
                    const double* tcov9, int K, const int* idx, const double* w,
                    const double init_qt[7], double out_qt[7], int max_trace, double* trace_cost,
                    double* trace_radius, double* trace_cand_cost, int* trace_accepted, int* n_trace);

#ifdef __cplusplus
}
#endif
#endif /* SICP_ORACLE_H_ */
