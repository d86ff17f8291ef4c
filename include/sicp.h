/*
 * sicp.h -- C ABI of the MI355X-native semantic-ICP registration engine.
 *
 * This is the drop-in boundary for the hot path of kxhit/semantic-icp: the body
 * of align() in the reference's three header-only registration classes.  The
 * reference has no FFI of its own (it is header-only C++ that the drivers
 * instantiate directly), so every entry point below cites the reference
 * member(s) it replaces; the C++ class shims in semantic-icp_amd/host/ keep the
 * reference's class/method names on top of this ABI.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, opaque handle, int status
 *     (0 = ok, negative = error).  No exception crosses the ABI: every entry
 *     point and the stream worker thread run inside a barrier that turns
 *     std::bad_alloc into SICP_ERR_OUT_OF_MEMORY and anything else into
 *     SICP_ERR_INTERNAL (csrc/abi_barrier.hpp).  The library never prints --
 *     unless the developer sets SICP_DEBUG in the environment, which unlocks
 *     the stderr logs of SICP_KNN_STATS / SICP_SOLO_LOG / SICP_STREAM_LOG.
 *   - The caller owns every input/output buffer (host memory unless a name
 *     ends in _device); the handle owns all device memory.
 *   - One handle = one HIP device + one stream.  Handles are independent and
 *     may be driven from different host threads or processes (that is how
 *     scan pairs shard across the 8 GPUs of a node); a single handle is not
 *     thread-safe, like the reference classes.
 *   - Pose exchange format: Sophus storage order qt[7] = [qx qy qz qw tx ty tz]
 *     (reference: gicp_cost_function.h:64-70).  Tangent order [upsilon; omega].
 *   - Clouds are SoA float32 xyz (+ uint32 labels): the layout the kernels read.
 *     Labels are 1..C for SICP_MODE_EM (reference quirk: em_icp.hpp:301 indexes
 *     label-1), arbitrary for SICP_MODE_SEMANTIC, ignored for SICP_MODE_GICP.
 */
#ifndef SICP_H_
#define SICP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SICP_VERSION_MAJOR 0
#define SICP_VERSION_MINOR 6
#define SICP_MAX_K_COV 32  /* largest covariance neighbourhood (ctor argument k) */

/* ---- status codes ------------------------------------------------------- */
enum {
  SICP_OK = 0,
  SICP_ERR_INVALID_ARGUMENT = -1,
  SICP_ERR_NO_DEVICE = -2,       /* no HIP device / HIP runtime failure at create */
  SICP_ERR_HIP = -3,             /* a HIP call failed; see sicp_last_error()       */
  SICP_ERR_NOT_READY = -4,       /* clouds / confusion matrix missing for the mode */
  SICP_ERR_TOO_FEW_POINTS = -5,  /* target has fewer than K points (reference: UB,
                                    em_icp.hpp:62-65)                             */
  SICP_ERR_BAD_LABEL = -6,       /* EM label outside 1..C (reference: UB)          */
  SICP_ERR_OUT_OF_MEMORY = -7,   /* host (std::bad_alloc) or device (hipErrorOutOfMemory, or the
                                    limit of sicp_set_memory_limit) memory exhausted     */
  SICP_ERR_INTERNAL = -8         /* a C++ exception was caught at the ABI boundary; see
                                    sicp_last_error() / sicp_stream_last_error()        */
};

/* ---- which reference class the handle behaves as ------------------------ */
enum {
  SICP_MODE_GICP = 0,     /* semanticicp::GICP<PointT>             gicp.h:15-132        */
  SICP_MODE_EM = 1,       /* EmIterativeClosestPoint<N>            em_icp.h:17-122      */
  SICP_MODE_SEMANTIC = 2  /* SemanticIterativeClosestPoint<P,S>    semantic_icp.h:15-83 */
};

enum { SICP_SOURCE = 0, SICP_TARGET = 1 };

/* sicp_params.profile bits.  Each timed kernel adds one event synchronisation. */
enum {
  SICP_PROFILE_NN = 1,      /* correspondence search (nn_partial kernel)          */
  SICP_PROFILE_COV = 2,     /* covariance self-kNN (nn_partial<20> kernel)        */
  SICP_PROFILE_WEIGHT = 4,  /* EM weight kernel                                   */
  SICP_PROFILE_ACC = 8      /* accumulate + finalize kernels, per LM evaluation   */
};

typedef struct sicp_context* sicp_handle;

/* Tunables.  The reference hard-codes most of these as literals; defaults
 * (sicp_default_params) reproduce them exactly. */
typedef struct sicp_params {
  int32_t mode;            /* SICP_MODE_*                                              */
  int32_t knn;             /* correspondences per source point: 4 (em_icp.hpp:60) or 1  */
  int32_t k_cov;           /* covariance neighbourhood (ctor arg k = 20, em_icp.h:42):
                              any 1..SICP_MAX_K_COV                                      */
  int32_t num_classes;     /* runtime C, replaces template parameter N (em_icp.h:16)    */
  double epsilon;          /* ctor arg epsilon = 1e-3 (em_icp.h:43)                     */
  double gate_sq;          /* 250, strict < (em_icp.hpp:65)                             */
  double cauchy_a;         /* 3.0 (em_icp.hpp:111) / 1.5 (semantic_icp.hpp:96)          */
  int32_t use_sqloss;      /* ComposedLoss(.., SQLoss): 1 EM/GICP, 0 Semantic            */
  int32_t max_outer;       /* 50 (em_icp.hpp:180) / 35 (semantic_icp.hpp:152)            */
  double outer_tol;        /* 1e-5 / 1e-3 on ||log(T_cur^-1 T_est)||^2                   */
  int32_t min_class_pts;   /* 400, strict > (semantic_icp.hpp:51)                        */
  int32_t max_lm_iterations;      /* 400 (em_icp.hpp:169)                                */
  double gradient_tolerance;      /* 1e-11 (em_icp.hpp:163)                              */
  double function_tolerance;      /* 1e-11 (em_icp.hpp:164)                              */
  /* Ceres defaults the reference leaves untouched (Ceres 1.14..2.1 solver.h) */
  double parameter_tolerance;     /* 1e-8  */
  double initial_radius;          /* 1e4   */
  double max_radius;              /* 1e16  */
  double min_radius;              /* 1e-32 */
  double min_relative_decrease;   /* 1e-3  */
  double min_lm_diagonal;         /* 1e-6  */
  double max_lm_diagonal;         /* 1e32  */
  int32_t max_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;         /* 1 */
  /* Reference quirks, reproduced by default (SURVEY.md 8a "Quirks") */
  int32_t quirk_bool_probability; /* Q1: GICPCostFunction::Probability returns bool
                                     (gicp_cost_function.h:75); 0 = use the double   */
  int32_t quirk_float_products;   /* Q2: float32 products in the covariance moments
                                     (em_icp.hpp:307-314); 0 = double products       */
  /* engine knobs (no reference counterpart) */
  int32_t nn_method;              /* 0 = LDS-tiled brute force, 1 = Hilbert box-tree, the
                                     16 queries of a wave walk it together (default),
                                     2 = box-tree, every query walks alone; all exact,
                                     bit-identical results                              */
  int32_t profile;                /* SICP_PROFILE_* bit mask: bracket those kernels with
                                     HIP events on the handle's stream (sicp_stats)     */
  int32_t lm_on_device;           /* 0 = host loop (one synchronisation per evaluation);
                                     1 (default; 3 is an alias) = trust-region state lives on
                                     the GPU: ticks of [accumulate kernel, LM-step kernel] x
                                     lm_batch as one graph launch, the host polls once per tick;
                                     the ONLY pair still iterating -- sicp_align, or the tail of
                                     a batch / stream -- continues its solve as persistent
                                     launches (its chunks stay in registers, one workgroup per
                                     chunk + one that steps the solver) when they
                                     fit the chip, falling back to the ticks if that grid cannot
                                     become resident;
                                     2 = as 1 without the persistent launches.
                                     Same machine (csrc/lm.hpp), same iterates, same bits.  */
  int32_t lm_batch;               /* evaluations queued per host poll (lm_on_device)    */
  int32_t reuse_features;         /* 0 (default) = recompute normals / histograms on every
                                     align() like em_icp.hpp:28-29 and gicp.hpp:33-34 do;
                                     1 = keep them while the uploaded cloud, k and C are
                                     unchanged (same values: they only depend on the cloud),
                                     what setSourceCloud(cloud, kdtree, covs) gicp.h:48-56
                                     exists for                                          */
  int32_t reserved_;
} sicp_params;

/* Per-align() counters; times in milliseconds.  *_kernel_ms are HIP-event
 * times on the handle's stream and are only filled when params.profile = 1. */
typedef struct sicp_stats {
  int32_t outer_iters;      /* what getOuterIter() returns (em_icp.h:88-91)          */
  int32_t total_lm_iters;
  int32_t total_evals;      /* accumulate passes ("E" of SURVEY.md 8d), all outer    */
  int32_t weights_in_search; /* correspondence searches of this align() that wrote the EM weights themselves
                               (a handle alone, batches of <= 4 pairs): these have no weight launch of their own,
                               so weight_launches + weights_in_search = searches with weights            */
  int64_t total_corr;       /* sum over outer passes of N_s*K candidate slots        */
  int64_t total_active;     /* slots that passed the distance gate                   */
  double final_cost;
  double t_cov_ms, t_nn_ms, t_weight_ms, t_solve_ms, t_total_ms;  /* host wall clock */
  double cov_kernel_ms;     /* nn_partial<k_cov> (self-kNN) launches                 */
  double nn_kernel_ms;      /* nn_partial<K> (correspondence search) launches        */
  double weight_kernel_ms;  /* EM weight kernel launches                             */
  double acc_kernel_ms;     /* accumulate + finalize pairs                           */
  int32_t cov_launches, nn_launches, weight_launches, acc_launches; /* timed launches */
  /* LM evaluation launches this pair was part of (ticks of lm_batch evaluations, one graph launch
   * each, shared by all pairs of a sicp_align_batch), including the tail of a tick it sat through
   * after its own inner solve had finished: idle slots = lockstep_slots - total_evals */
  int32_t lockstep_slots;
  int32_t graph_builds;     /* hipGraph instantiations during this align (leader handle of a batch) */
} sicp_stats;

/* ---- lifetime ------------------------------------------------------------- */
int sicp_device_count(int* count);
/* replaces the three class constructors (em_icp.h:42-48, gicp.h:34-40,
 * semantic_icp.h:35-39); mode and tunables follow via sicp_set_params */
int sicp_create(int device_id, sicp_handle* out);
int sicp_destroy(sicp_handle h);
/* Uploaded clouds (device buffers, search structures, pinned staging memory) are recycled through a
 * per-device pool when their last handle lets go of them, and all device buffers are carved from a
 * per-device arena that is not returned to the driver by itself; this frees what the pool of
 * `device_id` currently holds and every arena slab no live buffer sits in (it waits for the device
 * first).  Never required. */
int sicp_release_pool(int device_id);
/* Device memory the library may hold on `device_id`, in bytes (0 = no limit, the default): the arena takes no new
 * slab from the driver beyond it, and whatever then cannot be allocated -- a cloud, a handle's buffers -- fails with
 * SICP_ERR_OUT_OF_MEMORY (a status, like every other failure; the handle / stream stays usable for smaller work).
 * Memory already held is not given back by lowering the limit (sicp_release_pool does that).  For a process that
 * shares its GPU.  sicp_memory_reserved: what the arena holds right now. */
int sicp_set_memory_limit(int device_id, int64_t bytes);
int sicp_memory_reserved(int device_id, int64_t* bytes);
const char* sicp_strerror(int status);
const char* sicp_last_error(sicp_handle h); /* detail of the last SICP_ERR_HIP */
const char* sicp_version(void);

/* ---- configuration --------------------------------------------------------- */
int sicp_default_params(int mode, sicp_params* p);
int sicp_set_params(sicp_handle h, const sicp_params* p);
int sicp_get_params(sicp_handle h, sicp_params* p);

/* setSourceCloud / setTargetCloud (em_icp.h:50-66, gicp.h:42-70) and
 * setInputSource / setInputTarget (semantic_icp.h:41-49).  Copies the cloud to
 * HBM (SoA).  label may be NULL for SICP_MODE_GICP.  In SICP_MODE_SEMANTIC the
 * points are grouped by label in order of first appearance
 * (pcl_2_semantic.h:24-39); all outputs stay in the caller's point order. */
int sicp_set_cloud(sicp_handle h, int which, int32_t n, const float* x, const float* y,
                   const float* z, const uint32_t* label);
/* The same from an array of points as the reference holds them (pcl::PointCloud<pcl::PointXYZL>::points,
 * em_icp.h:37 / exec/kitti_eval.cc:132: x y z at bytes 0 4 8 of a 32-byte point, the label at 16; a plain
 * float[n][3] has stride 12): `xyz` = address of the first point's x, `label` = address of the first
 * point's uint32 label or NULL; strides in bytes.  One pass over the caller's memory, no intermediate
 * arrays. */
int sicp_set_cloud_strided(sicp_handle h, int which, int32_t n, const void* xyz, int64_t stride_bytes,
                           const void* label, int64_t label_stride_bytes);
/* Non-finite points (NaN / Inf in any coordinate, e.g. the invalid pixels of an organized RGB-D
 * cloud) are accepted and LEFT OUT of the device cloud, as pcl::KdTreeFLANN::setInputCloud
 * (called by setSourceCloud / setTargetCloud, em_icp.h:50-66) leaves them out of its index: they
 * are never found as neighbours and -- a NaN query keeps no candidate -- never matched, so they
 * contribute no residual.  Every per-point output keeps the caller's size and order; for a
 * dropped point: correspondences idx = -1, d2 = NaN, w = 0; normal / covariance NaN, histogram
 * 0, neighbour list -1; fused label 0; sicp_transform_source transforms it like any other point.
 * n_points = what the caller handed over (the size of per-point outputs), n_indexed = the finite
 * points held on the device.  Either output may be NULL. */
int sicp_cloud_size(sicp_handle h, int which, int32_t* n_points, int32_t* n_indexed);
/* same, from buffers already resident on the handle's device */
int sicp_set_cloud_device(sicp_handle h, int which, int32_t n, const float* x_device,
                          const float* y_device, const float* z_device,
                          const uint32_t* label_device);
/* setSourceCloud(cloud, kdtree, covs) / setTargetCloud(cloud, kdtree, covs) (gicp.h:48-56,
 * 64-70, em_icp.h:50-66 via getTargetKdTree()/getTargetCovariances(), used by
 * exec/kitti_eval.cc:207-226 to hand one scan's search tree and covariances from one
 * registration to the next): slot `which` of `h` refers to the SAME device-resident cloud
 * (points, search structure, normals, histograms) as slot `from_which` of `from`; nothing is
 * copied or rebuilt.  Both handles must be on the same device.  Handles that share a cloud may
 * run in one sicp_align_batch, or one after the other; two host threads must not ALIGN handles
 * that share a cloud at the same time.  One thing is safe across threads, because a sequence
 * driver needs it: while one thread registers handles (sicp_align / sicp_align_batch), another
 * may upload clouds into OTHER handles and sicp_share_cloud from a handle of the running call --
 * a shared cloud is never written by an align, a handle that gets a new cloud lets go of the
 * shared one instead of overwriting it, and the "upload still in flight" flag is atomic. */
int sicp_share_cloud(sicp_handle h, int which, sicp_handle from, int from_which);
/* setConfusionMatrix (em_icp.h:68-71); cm is C*C row-major, cm[r*C+s] */
int sicp_set_confusion(sicp_handle h, int32_t C, const double* cm_rowmajor);

/* ---- the hot path ------------------------------------------------------------ */
/* align(final, init) + getFinalTransFormation() + getOuterIter()
 * (em_icp.hpp:25-200, gicp.hpp:29-175, semantic_icp.hpp:28-166).
 * stats may be NULL. */
int sicp_align(sicp_handle h, const double init_qt[7], double out_qt[7],
               int32_t* outer_iters, sicp_stats* stats);
/* n independent align() calls -- one handle per scan pair, all on one device, same mode /
 * knn / solver knobs -- batched continuously: every launch of the inner solve evaluates the
 * pairs that are inside a solve (up to 256; with more, the others wait with their search done),
 * the searches of the pairs between two solves run between the launches (what
 * exec/kitti_eval.cc:124-249 does pair after pair).  Per pair the result is bit-identical to
 * sicp_align on that handle.  init_qt, out_qt: n*7; outer_iters (nullable): n; stats
 * (nullable): n.  No reference counterpart. */
int sicp_align_batch(sicp_handle* handles, int32_t n, const double* init_qt, double* out_qt,
                     int32_t* outer_iters, sicp_stats* stats);

/* ---- registration streams: an OPEN sequence of scan pairs --------------------------------------
 * The unit the reference iterates over is one align() per loop trip (exec/kitti_eval.cc:124-249:
 * ~4.5K stride-3 pairs of one odometry sequence, every scan the source of one registration and the
 * target of the next).  A stream is the continuous batching of sicp_align_batch without the closed
 * batch: registrations are submitted as their scans arrive and come out as they converge, up to
 * max_in_flight of them share the GPU, and a pair that needs 500 LM evaluations does not hold back the
 * batch it happened to be submitted with.  Per pair the result is bit-identical to sicp_align.
 *
 *   sicp_stream_create(device, params, max_in_flight, &s)   mode, K, tolerances ... of every registration
 *   sicp_stream_set_confusion(s, C, cm)                      SICP_MODE_EM
 *   id = sicp_stream_add_cloud(s, n, x, y, z, label)         setSourceCloud / setTargetCloud
 *        (em_icp.h:50-66): copies the cloud into pinned memory and queues upload + search-tree
 *        build on the stream's own HIP stream, beside the running registrations; returns at once.
 *        A cloud may take part in any number of registrations: it is uploaded and indexed once
 *        and its normals / histograms are computed once (what setSourceCloud(cloud, kdtree, covs),
 *        gicp.h:48-56, exists for).
 *   ticket = sicp_stream_submit(s, source_id, target_id, init_qt)   align(final, init); blocks
 *        while max_in_flight registrations are already waiting (back-pressure)
 *   sicp_stream_release_cloud(s, id)                         the caller is done with it; it is
 *        recycled when its last registration has finished
 *   sicp_stream_poll(s, wait, max, results, &n)              finished registrations, in order of
 *        completion: wait = 0 returns what is there, 1 waits for at least one result (or for the
 *        stream to run dry), 2 waits until everything submitted so far has finished
 *
 * A library-owned worker thread advances the registrations; submit / add_cloud / poll may be called
 * from any thread(s).  Requires the default engine (nn_method 1, lm_on_device 1, profile 0).
 * No reference counterpart. */
typedef struct sicp_stream_ctx* sicp_stream;
typedef struct sicp_stream_result {
  int64_t ticket;       /* what sicp_stream_submit returned */
  int32_t status;       /* SICP_OK, or why this registration could not run */
  int32_t outer_iters;  /* getOuterIter() */
  double qt[7];         /* getFinalTransFormation() */
  sicp_stats stats;     /* as from sicp_align, except total_active (0: not counted in a stream) */
} sicp_stream_result;
int sicp_stream_create(int device_id, const sicp_params* params, int32_t max_in_flight, sicp_stream* out);
int sicp_stream_destroy(sicp_stream s);  /* registrations still in flight are abandoned */
int sicp_stream_set_confusion(sicp_stream s, int32_t C, const double* cm_rowmajor);
int sicp_stream_add_cloud(sicp_stream s, int32_t n, const float* x, const float* y, const float* z,
                          const uint32_t* label, int64_t* cloud_id);
/* sicp_stream_add_cloud from an array of points (see sicp_set_cloud_strided) */
int sicp_stream_add_cloud_strided(sicp_stream s, int32_t n, const void* xyz, int64_t stride_bytes,
                                  const void* label, int64_t label_stride_bytes, int64_t* cloud_id);
int sicp_stream_release_cloud(sicp_stream s, int64_t cloud_id);
int sicp_stream_submit(sicp_stream s, int64_t source_id, int64_t target_id, const double init_qt[7],
                       int64_t* ticket);
/* sicp_stream_submit with options (flags = 0: the same):
 *   SICP_SUBMIT_FUSED_LABELS   SICP_MODE_EM: getFusedLabels(out, final pose) (em_icp.hpp:202-268, what
 *       exec/scenenet_eval.cc:193-198 calls right after align) is computed when the registration retires -- one more
 *       K = 4 search and one label kernel, queued beside the running registrations -- and kept until
 *       sicp_stream_take_labels(ticket) fetches it (once; n = the source cloud's point count, caller order).  The
 *       registration's result is only handed out by sicp_stream_poll when its labels are there.  Taking them is
 *       the caller's side of the contract: a label set (4 bytes per source point) stays in host memory until it is taken or
 *       the stream is destroyed -- a caller may take them long after the poll that returned the registration.
 *   SICP_SUBMIT_FRESH_FEATURES the normals / label histograms of BOTH clouds are recomputed for this registration,
 *       like every align() of the reference does (em_icp.hpp:28-29, gicp.hpp:33-34), instead of being kept with the
 *       cloud (a stream's default: what setSourceCloud(cloud, kdtree, covs) exists for).  Same values either way. */
enum { SICP_SUBMIT_FUSED_LABELS = 1, SICP_SUBMIT_FRESH_FEATURES = 2 };
int sicp_stream_submit_ex(sicp_stream s, int64_t source_id, int64_t target_id, const double init_qt[7],
                          uint32_t flags, int64_t* ticket);
int sicp_stream_take_labels(sicp_stream s, int64_t ticket, int32_t n, uint32_t* out_labels);
int sicp_stream_poll(sicp_stream s, int32_t wait, int32_t max_results, sicp_stream_result* results,
                     int32_t* n_results);
/* counters since creation: registrations submitted / finished, and -- over the finished ones -- their
 * own LM evaluations and the evaluation launches they sat through (busy fraction = the ratio).  Any
 * output may be NULL. */
int sicp_stream_counters(sicp_stream s, int64_t* submitted, int64_t* completed, int64_t* busy_evals,
                         int64_t* slot_evals);
const char* sicp_stream_last_error(sicp_stream s);

/* Caller-supplied per-point covariances, where the reference reads them instead of computing them:
 * SemanticIterativeClosestPoint::align takes whatever sits in the public `labeledCovariances` of its two clouds
 * (impl/semantic_icp.hpp:73,77; semantic_point_cloud.h:36-42: addSemanticCloud(..., computeKd, computeCov = false) leaves
 * them to the caller), and GICP::setSourceCloud(cloud, tree, covs) / setTargetCloud(cloud, tree, covs) (gicp.h:50-55, 65-70)
 * accept arbitrary vectors (GICP::align and EmIterativeClosestPoint::align then overwrite them, impl/gicp.hpp:33-34,
 * impl/em_icp.hpp:28-29 -- which is what this engine does too unless reuse_features is set).
 * cov9: n_points x 9 row-major 3x3 matrices in the caller's point order.
 *   - every matrix of the form the reference's own routine produces, C = I - (1 - epsilon) n n^T (a unit normal n; epsilon =
 *     params.epsilon; to 1e-8): the engine keeps the normals and registers with its product kernels, batches and streams included;
 *   - otherwise, every matrix symmetric and finite (SICP_MODE_GICP / SICP_MODE_SEMANTIC): kept as they are, and the cloud's
 *     registrations evaluate gicp_cost_function.h:27-73 on the full 3x3 matrices (its closed form for symmetric covariances),
 *     with the trust-region loop on the host: correct to the same tolerances, ONE PAIR AT A TIME (sicp_align, sicp_solve,
 *     sicp_accumulate; sicp_align_batch of more than one pair and streams answer SICP_ERR_INVALID_ARGUMENT) and far from the
 *     product path's speed -- the path of an exotic input, e.g. exec/test_gradient.cc:32-50's fixture;
 *   - anything else (a non-symmetric or non-finite matrix; a general matrix in SICP_MODE_EM, whose align() recomputes
 *     covariances and label histograms together) is REFUSED: SICP_ERR_INVALID_ARGUMENT, sicp_last_error names the first
 *     offending point, nothing is changed -- never a silent replacement.
 * Accepted covariances count as the cloud's current features for SICP_MODE_SEMANTIC, and for SICP_MODE_GICP with
 * reuse_features = 1 (without it align() recomputes them, as impl/gicp.hpp:33-34 does).  Entries of non-finite points (which
 * never reach the device) are ignored.  sicp_covariances hands back what is there. */
int sicp_set_covariances(sicp_handle h, int which, const double* cov9);

/* the final_cloud output of align (em_icp.hpp:192-198): source transformed by
 * float(matrix(qt)); ox/oy/oz are host buffers of n_source floats */
int sicp_transform_source(sicp_handle h, const double qt[7], float* ox, float* oy, float* oz);

/* getFusedLabels (em_icp.hpp:202-268): out_labels[n_source] */
int sicp_fused_labels(sicp_handle h, const double qt[7], uint32_t* out_labels);

/* ---- test / bench hooks: the individual stages -------------------------------- */
/* ComputeCovariances (em_icp.hpp:270-343 = gicp.hpp:177-239 =
 * semantic_point_cloud.hpp:25-84) for one cloud.  Any output may be NULL.
 * cov9: n*9 row-major 3x3; normal3: n*3; hist: n*C uint8 neighbour counts
 * (the reference's double histogram is count * (1/k) accumulated, em_icp.hpp:301);
 * nn_idx: n*k_cov neighbour indices. */
int sicp_covariances(sicp_handle h, int which, double* cov9, double* normal3, uint8_t* hist,
                     int32_t* nn_idx);
/* transform + nearestKSearch + gate (+ EM weight) at pose qt, i.e. the
 * correspondence loop em_icp.hpp:46-108.  idx: n_source*K target indices
 * (-1 = gated out), d2: n_source*K float32 squared distances, w: n_source*K
 * weights (prob; 1 for non-EM).  Any output may be NULL; results stay on the
 * device for sicp_accumulate. */
int sicp_correspondences(sicp_handle h, const double qt[7], int32_t* idx, float* d2, double* w);
/* one evaluation sweep of the inner solve at pose qt over the current
 * correspondences: out28 = [H upper-triangular 21 | g 6 | cost], H = sum rho1 J J^T,
 * g = sum rho1 r J, cost = 1/2 sum rho0 (what ceres::Evaluator produces from
 * GICPCostFunction::Evaluate gicp_cost_function.h:27-73 through the losses). */
int sicp_accumulate(sicp_handle h, const double qt[7], double out28[28]);
/* the same sweep for n handles in ONE launch (the kernel sicp_align_batch runs): qt n*7,
 * out28 n*28.  The launch is issued `repeat` (>= 1) times back to back between two HIP
 * events on handles[0]'s stream; kernel_ms (nullable) = event time / repeat. */
int sicp_accumulate_batch(sicp_handle* handles, int32_t n, const double* qt, double* out28,
                          int32_t repeat, double* kernel_ms);
/* the search kernels of n handles as sicp_align_batch launches them (one job per handle and label
 * segment, up to 8 jobs per launch), issued `repeat` times back to back between two HIP events on
 * handles[0]'s stream; kernel_ms (nullable) = event time / repeat, for all n searches together.
 *   what = 0: the correspondence search at poses qt (n*7): transform + K nearest targets + gate
 *             (em_icp.hpp:46-65); use_hint = 0 starts every walk from the curve position like the first
 *             search of an align(), 1 from the handle's previous result like the later ones
 *   what = 1 / 2: the k_cov self-search of the source / target cloud (em_icp.hpp:283-296)
 *   what = 3: what = 0 as an EM-ICP align() launches it after its first flush of features: the search writes the slots' EM
 *             weights in its epilogue (em_icp.hpp:77-89,108; K = 4, at most 16 classes), or a weight kernel follows once
 * The handles end up as after sicp_correspondences / sicp_covariances. */
int sicp_search_batch(sicp_handle* handles, int32_t n, const double* qt, int32_t what, int32_t use_hint,
                      int32_t repeat, double* kernel_ms);
/* the inner ceres::Solve (em_icp.hpp:162-177) on the current correspondences */
int sicp_solve(sicp_handle h, const double init_qt[7], double out_qt[7], int32_t* lm_iters,
               int32_t* evals, double* final_cost);

/* the device build of csrc/se3.hpp (what the device-resident LM step runs in place of Sophus,
 * local_parameterization_se3.h:17-25), one lane per item; op and layouts:
 *   SICP_SE3_EXP  in n*6 tangents [upsilon; omega]      -> out n*7 poses
 *   SICP_SE3_LOG  in n*7 poses                          -> out n*6
 *   SICP_SE3_PLUS in n*13 = pose (7) | delta (6)        -> out n*7 = pose * exp(delta)
 *   SICP_SE3_MUL  in n*14 = pose a (7) | pose b (7)     -> out n*7
 *   SICP_SE3_INV  in n*7                                -> out n*7
 * and the device build of csrc/lm.hpp (the inner ceres::Solve's step control, em_icp.hpp:162-177), one wavefront per item:
 *   SICP_LM_SEQUENCE           in n*679 = start pose (7) | 24 evaluations x 28 sums [H upper 21 | g 6 | cost], fed in turn to
 *                              the trust-region machine (csrc/lm.hpp, default options) AS THE KERNELS RUN IT -- by a whole
 *                              wavefront, the finite test by ballot -- until it stops or the sequence ends
 *                              -> out n*37 = pose 7 | x 7 | diag 6 | scale 6 | radius, cost, model change, decrease factor,
 *                              |x| | status, iterations, evaluations, invalid steps, reuse_diagonal, phase
 *   SICP_LM_SEQUENCE_ONE_LANE  the same in the one-lane form the host loop runs: the two must agree bit for bit on ANY
 *                              sequence (the machine is a pure function of state and evaluation) */
enum { SICP_SE3_EXP = 0, SICP_SE3_LOG = 1, SICP_SE3_PLUS = 2, SICP_SE3_MUL = 3, SICP_SE3_INV = 4, SICP_LM_SEQUENCE = 5, SICP_LM_SEQUENCE_ONE_LANE = 6 };
int sicp_se3_device(sicp_handle h, int op, int32_t n, const double* in, double* out);

/* counters accumulated since the last sicp_align() began (the hooks above add to them) */
int sicp_get_stats(sicp_handle h, sicp_stats* stats);

int sicp_synchronize(sicp_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SICP_H_ */
