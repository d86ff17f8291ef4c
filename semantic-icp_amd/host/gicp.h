// gicp.h -- semanticicp::GICP<PointT> with the reference's public surface
// (reference: semantic_icp/gicp.h:15-132, impl/gicp.hpp).  align() runs on the MI355X engine.
#ifndef GICP_H_
#define GICP_H_
#include <memory>
#include <vector>

#include "semantic_point_cloud.h"
#include "sicp_engine.hpp"

namespace semanticicp {

template <typename PointT>
class GICP {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d>> MatricesVector;
  typedef std::vector<Eigen::Matrix<double, 6, 6>, Eigen::aligned_allocator<Eigen::Matrix<double, 6, 6>>> CovarianceVector;
  typedef std::shared_ptr<MatricesVector> MatricesVectorPtr;
  typedef std::shared_ptr<const MatricesVector> MatricesVectorConstPtr;
  typedef pcl::KdTreeFLANN<PointT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef Eigen::Matrix<double, 6, 1> Vector6d;

  GICP(int k = 20, double epsilon = 0.001) : kCorrespondences_(k), epsilon_(epsilon), outer_iter(0) {}

  // reference: gicp.h:42-70.  The kd-tree / covariance arguments of the 3-argument overloads are
  // accepted and handed back by the getters, but align() recomputes the covariances from the
  // cloud, as the reference does (impl/gicp.hpp:33-34).  (exec/kitti_eval.cc:213 passes the tree
  // of a *different* scan there -- SURVEY.md quirk Q7 -- which this engine therefore ignores.)
  // The covariance vectors align() fills (impl/gicp.hpp:33-34) are fetched from the GPU on the
  // first getSourceCovariances() / getTargetCovariances() call after an align(), not inside it.
  // Every setter also forgets the "covariances of the last align() are still on the GPU" state: until
  // the next align() the getters hand back exactly what the reference would (a fresh empty vector, or
  // the caller's own), never the previous cloud's values.
  inline void setSourceCloud(const PointCloudPtr& cloud) {
    source_cov_supplied_ = false;
    sourceCloud_ = cloud;
    sourceKdTree_ = KdTreePtr(new KdTree());
    sourceKdTree_->setInputCloud(sourceCloud_);
    sourceCovariances_ = MatricesVectorPtr(new MatricesVector());
    source_cov_stale_ = false;
  }
  inline void setSourceCloud(const PointCloudPtr& cloud, const KdTreePtr& tree, const MatricesVectorPtr& covs) {
    sourceCloud_ = cloud; sourceKdTree_ = tree; sourceCovariances_ = covs;
    source_cov_stale_ = false;
    source_cov_supplied_ = covs && cloud && !covs->empty() && covs->size() == cloud->size();
  }
  inline void setTargetCloud(const PointCloudPtr& cloud) {
    target_cov_supplied_ = false;
    shared_target_from_ = nullptr;
    targetCloud_ = cloud;
    targetKdTree_ = KdTreePtr(new KdTree());
    targetKdTree_->setInputCloud(targetCloud_);
    targetCovariances_ = MatricesVectorPtr(new MatricesVector());
    target_cov_stale_ = false;
  }
  inline void setTargetCloud(const PointCloudPtr& cloud, const KdTreePtr& tree, const MatricesVectorPtr& covs) {
    shared_target_from_ = nullptr;
    targetCloud_ = cloud; targetKdTree_ = tree; targetCovariances_ = covs;
    target_cov_stale_ = false;
    target_cov_supplied_ = covs && cloud && !covs->empty() && covs->size() == cloud->size();
  }
  // Engine extensions for scan sequences: the target of this registration is the source cloud of
  // `other` as it lives on the GPU after other's align() (one upload, tree and covariance set per scan;
  // what the 3-argument overloads above exist for), and keepFeatures(true) keeps the covariances of a
  // cloud across align() calls instead of recomputing them (impl/gicp.hpp:33-34; same values).
  // Call it BEFORE other.setSourceCloud(next scan) when `other` is this very object's slot of the
  // previous batch: the host-side target pointer is taken from other's current source here.
  inline void setTargetCloudSharedWithSourceOf(GICP& other) {
    targetCloud_ = other.sourceCloud_;
    shared_target_from_ = &other;
    targetCovariances_ = MatricesVectorPtr(new MatricesVector());
    target_cov_stale_ = false;
  }
  inline void keepFeatures(bool on) { reuse_features_ = on; }
  inline KdTreePtr getSourceKdTree() { return sourceKdTree_; }
  inline MatricesVectorPtr getSourceCovariances() { fetch_covariances(SICP_SOURCE, sourceCovariances_, source_cov_stale_); return sourceCovariances_; }
  inline KdTreePtr getTargetKdTree() { return targetKdTree_; }
  inline MatricesVectorPtr getTargetCovariances() { fetch_covariances(SICP_TARGET, targetCovariances_, target_cov_stale_); return targetCovariances_; }

  void align(PointCloudPtr finalCloud) {  // reference: impl/gicp.hpp:21-27
    Sophus::SE3d init;
    align(finalCloud, init);
  }

  void align(PointCloudPtr finalCloud, Sophus::SE3d& initTransform) {  // reference: impl/gicp.hpp:29-175
    sicp_handle h = engine_.get();
    sicp_params p;
    detail::check(sicp_default_params(SICP_MODE_GICP, &p), h, "sicp_default_params");
    p.k_cov = kCorrespondences_;
    p.epsilon = epsilon_;
    p.reuse_features = reuse_features_ ? 1 : 0;
    detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
    upload_clouds(h);
    double out[7];
    int32_t iters = 0;
    detail::check(sicp_align(h, initTransform.data(), out, &iters, nullptr), h, "sicp_align");
    finalTransformation_ = detail::to_se3(out);
    outer_iter = iters;
    source_cov_stale_ = target_cov_stale_ = true;  // filled on demand by the getters
    if (finalCloud != nullptr) {  // impl/gicp.hpp:166-172
      Eigen::Matrix4f mat = (finalTransformation_.matrix()).template cast<float>();
      pcl::transformPointCloud(*sourceCloud_, *finalCloud, mat);
    }
  }

  Sophus::SE3d getFinalTransFormation() { return finalTransformation_; }
  int getOuterIter() { return outer_iter; }

  // Engine extension (no reference counterpart): align() of several objects, one per scan pair with
  // its clouds set, advanced in lock step by sicp_align_batch; every object ends up as after its own
  // align().  finalClouds may hold nullptrs.
  static void alignBatch(const std::vector<GICP*>& objs, const std::vector<PointCloudPtr>& finalClouds,
                         const std::vector<Sophus::SE3d>& initTransforms) {
    const size_t n = objs.size();
    if (n == 0) return;
    if (finalClouds.size() != n || initTransforms.size() != n) throw std::runtime_error("alignBatch: argument sizes differ");
    std::vector<sicp_handle> hs(n);
    std::vector<double> init(7 * n), out(7 * n);
    std::vector<int32_t> iters(n, 0);
    for (size_t q = 0; q < n; ++q) {
      GICP& o = *objs[q];
      sicp_handle h = hs[q] = o.engine_.get();
      sicp_params p;
      detail::check(sicp_default_params(SICP_MODE_GICP, &p), h, "sicp_default_params");
      p.k_cov = o.kCorrespondences_;
      p.epsilon = o.epsilon_;
      p.reuse_features = o.reuse_features_ ? 1 : 0;
      detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
      o.upload_clouds(h);
      for (int i = 0; i < 7; ++i) init[7 * q + i] = initTransforms[q].data()[i];
    }
    detail::check(sicp_align_batch(hs.data(), (int32_t)n, init.data(), out.data(), iters.data(), nullptr), hs[0], "sicp_align_batch");
    for (size_t q = 0; q < n; ++q) {
      GICP& o = *objs[q];
      o.finalTransformation_ = detail::to_se3(&out[7 * q]);
      o.outer_iter = iters[q];
      o.source_cov_stale_ = o.target_cov_stale_ = true;
      if (finalClouds[q] != nullptr) {
        Eigen::Matrix4f mat = (o.finalTransformation_.matrix()).template cast<float>();
        pcl::transformPointCloud(*o.sourceCloud_, *finalClouds[q], mat);
      }
    }
  }

 protected:
  // source: uploaded; target: uploaded, or the device-resident source cloud of another object
  void upload_clouds(sicp_handle h) {
    // a shared target refers to the OTHER object's current source: bind it before this object's source
    // slot is replaced (the other object may be this batch slot's predecessor from the previous batch)
    if (shared_target_from_)
      detail::check(sicp_share_cloud(h, SICP_TARGET, shared_target_from_->engine_.get(), SICP_SOURCE), h, "sicp_share_cloud");
    detail::check(detail::set_cloud(h, SICP_SOURCE, *sourceCloud_, false), h, "sicp_set_cloud_strided");
    if (!shared_target_from_) detail::check(detail::set_cloud(h, SICP_TARGET, *targetCloud_, false), h, "sicp_set_cloud_strided");
    // The covariances of the 3-argument setters.  The reference's align() overwrites them (impl/gicp.hpp:33-34), and so does
    // this engine -- unless keepFeatures(true) asked for a cloud's covariances to be kept across align() calls: then the
    // caller's are the ones kept: of the engine's form on the product kernels, any other symmetric matrix on the full-matrix
    // path (sicp_set_covariances); what is no covariance at all (not symmetric, not finite) is refused loudly.
    if (reuse_features_ && source_cov_supplied_) push_covariances(h, SICP_SOURCE, *sourceCovariances_);
    if (reuse_features_ && target_cov_supplied_ && !shared_target_from_) push_covariances(h, SICP_TARGET, *targetCovariances_);
  }
  static void push_covariances(sicp_handle h, int which, const MatricesVector& v) {
    std::vector<double> c9(v.size() * 9);
    for (size_t i = 0; i < v.size(); ++i)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) c9[i * 9 + 3 * a + b] = v[i](a, b);
    detail::check(sicp_set_covariances(h, which, c9.data()), h, "sicp_set_covariances");
  }

  // what align() computed on the GPU (normals -> C = I - (1-eps) n n^T), copied out once per align
  void fetch_covariances(int which, MatricesVectorPtr& out, bool& stale) {
    if (!stale) return;
    stale = false;
    sicp_handle h = engine_.get();
    // sized by the cloud that IS on the device (what sicp_covariances writes), not by a host pointer
    // that may have been replaced since the align()
    int32_t n = 0;
    detail::check(sicp_cloud_size(h, which, &n, nullptr), h, "sicp_cloud_size");
    if (!out) out = MatricesVectorPtr(new MatricesVector());
    out->resize(n);
    if (n == 0) return;
    detail::check(sicp_covariances(h, which, detail::matrix3d_storage(out->data()), nullptr, nullptr, nullptr), h, "sicp_covariances");
  }

  int kCorrespondences_;
  double epsilon_;
  int outer_iter;
  bool source_cov_stale_ = false, target_cov_stale_ = false;
  bool source_cov_supplied_ = false, target_cov_supplied_ = false;  // the 3-argument setters handed over one matrix per point
  bool reuse_features_ = false;
  GICP* shared_target_from_ = nullptr;
  Sophus::SE3d finalTransformation_;
  PointCloudPtr sourceCloud_;
  KdTreePtr sourceKdTree_;
  MatricesVectorPtr sourceCovariances_;
  PointCloudPtr targetCloud_;
  KdTreePtr targetKdTree_;
  MatricesVectorPtr targetCovariances_;
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // GICP_H_
