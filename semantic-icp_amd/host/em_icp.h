// em_icp.h -- semanticicp::EmIterativeClosestPoint<N> with the reference's public surface
// (reference: semantic_icp/em_icp.h:17-122, impl/em_icp.hpp).  align() / getFusedLabels() run on
// the MI355X engine; the class count N stays a template parameter here (the C ABI takes it at
// run time).
#ifndef SEMANTIC_ICP_EM_ICP_H_
#define SEMANTIC_ICP_EM_ICP_H_
#include <cstddef>
#include <memory>
#include <vector>

#include "sicp_engine.hpp"

namespace semanticicp {

template <size_t N>
class EmIterativeClosestPoint {
 public:
  typedef pcl::PointXYZL PointT;
  typedef typename pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  // reference: em_icp.h:22-35
  typedef std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d>> MatricesVector;
  typedef std::vector<Eigen::Matrix<double, 6, 6>, Eigen::aligned_allocator<Eigen::Matrix<double, 6, 6>>> CovarianceVector;
  typedef std::vector<Eigen::Matrix<double, (int)N, 1>, Eigen::aligned_allocator<Eigen::Matrix<double, (int)N, 1>>> DistVector;
  typedef std::shared_ptr<MatricesVector> MatricesVectorPtr;
  typedef std::shared_ptr<const MatricesVector> MatricesVectorConstPtr;
  typedef std::shared_ptr<DistVector> DistVectorPtr;
  typedef typename pcl::KdTreeFLANN<PointT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef Eigen::Matrix<double, 6, 1> Vector6d;

  EmIterativeClosestPoint(int k = 20, double epsilon = 0.001) : kCorrespondences_(k), kEpsilon_(epsilon), outer_iter(0), cm_set_(false) {}

  // reference: em_icp.h:50-66 (the reference builds its kd-tree here; the engine uploads the
  // cloud to HBM here)
  inline void setSourceCloud(const PointCloudPtr& cloud) { source_cloud_ = cloud; upload(SICP_SOURCE, cloud); }
  inline void setTargetCloud(const PointCloudPtr& cloud) { target_cloud_ = cloud; upload(SICP_TARGET, cloud); }

  // Engine extensions for scan sequences (no reference counterpart in this class; GICP has
  // setSourceCloud(cloud, kdtree, covs), gicp.h:48-56, for the same purpose): the target of this
  // registration IS the source cloud of `other` -- one upload, one search tree, one set of normals and
  // histograms on the GPU (sicp_share_cloud) -- and keepFeatures(true) computes them once per upload
  // instead of once per align() (impl/em_icp.hpp:28-29 recomputes; the values are the same).
  inline void setTargetCloudSharedWithSourceOf(EmIterativeClosestPoint& other) {
    target_cloud_ = other.source_cloud_;
    sicp_handle h = engine_.get();
    configure(h);
    detail::check(sicp_share_cloud(h, SICP_TARGET, other.engine_.get(), SICP_SOURCE), h, "sicp_share_cloud");
  }
  inline void keepFeatures(bool on) { reuse_features_ = on; }

  // reference: em_icp.h:68-71
  inline void setConfusionMatrix(const Eigen::Matrix<double, (int)N, (int)N>& in) {
    double cm[N * N];
    for (size_t r = 0; r < N; ++r)
      for (size_t c = 0; c < N; ++c) cm[r * N + c] = in((int)r, (int)c);
    sicp_handle h = engine_.get();
    detail::check(sicp_set_confusion(h, (int32_t)N, cm), h, "sicp_set_confusion");
    cm_set_ = true;
  }

  // reference: em_icp.h:73-74 declares this overload (and never defines it): identity start
  void align(PointCloudPtr finalCloud) { align(finalCloud, Sophus::SE3d()); }

  // reference: impl/em_icp.hpp:25-200 (finalCloud may be nullptr, :194)
  void align(PointCloudPtr finalCloud, const Sophus::SE3d& initTransform) {
    sicp_handle h = engine_.get();
    configure(h);
    double out[7];
    int32_t iters = 0;
    detail::check(sicp_align(h, initTransform.data(), out, &iters, nullptr), h, "sicp_align");
    final_transformation_ = detail::to_se3(out);
    outer_iter = iters;
    if (finalCloud != nullptr) {
      Eigen::Matrix4f mat = (final_transformation_.matrix()).template cast<float>();
      pcl::transformPointCloud(*source_cloud_, *finalCloud, mat);
    }
  }

  // reference: impl/em_icp.hpp:202-268 (appends one point per source point)
  void getFusedLabels(PointCloudPtr labeledCloud, const Sophus::SE3d& transformation) {
    sicp_handle h = engine_.get();
    configure(h);
    std::vector<uint32_t> lab(source_cloud_->size());
    detail::check(sicp_fused_labels(h, transformation.data(), lab.data()), h, "sicp_fused_labels");
    for (size_t i = 0; i < source_cloud_->size(); ++i) {
      PointT p = source_cloud_->points[i];
      p.label = lab[i];
      labeledCloud->push_back(p);
    }
  }

  Sophus::SE3d getFinalTransFormation() { return final_transformation_; }
  int getOuterIter() { return outer_iter; }

  // Engine extension (no reference counterpart): align() of several objects -- one per scan pair,
  // clouds and confusion matrices already set -- advanced in lock step by sicp_align_batch.  Every
  // object ends up exactly as if its own align() had been called.  finalClouds may hold nullptrs.
  static void alignBatch(const std::vector<EmIterativeClosestPoint*>& objs, const std::vector<PointCloudPtr>& finalClouds,
                         const std::vector<Sophus::SE3d>& initTransforms) {
    const size_t n = objs.size();
    if (n == 0) return;
    if (finalClouds.size() != n || initTransforms.size() != n) throw std::runtime_error("alignBatch: argument sizes differ");
    std::vector<sicp_handle> hs(n);
    std::vector<double> init(7 * n), out(7 * n);
    std::vector<int32_t> iters(n, 0);
    for (size_t p = 0; p < n; ++p) {
      hs[p] = objs[p]->engine_.get();
      objs[p]->configure(hs[p]);
      for (int i = 0; i < 7; ++i) init[7 * p + i] = initTransforms[p].data()[i];
    }
    detail::check(sicp_align_batch(hs.data(), (int32_t)n, init.data(), out.data(), iters.data(), nullptr), hs[0], "sicp_align_batch");
    for (size_t p = 0; p < n; ++p) {
      objs[p]->final_transformation_ = detail::to_se3(&out[7 * p]);
      objs[p]->outer_iter = iters[p];
      if (finalClouds[p] != nullptr) {
        Eigen::Matrix4f mat = (objs[p]->final_transformation_.matrix()).template cast<float>();
        pcl::transformPointCloud(*objs[p]->source_cloud_, *finalClouds[p], mat);
      }
    }
  }

 protected:
  void configure(sicp_handle h) {
    sicp_params p;
    detail::check(sicp_default_params(SICP_MODE_EM, &p), h, "sicp_default_params");
    p.k_cov = kCorrespondences_;
    p.epsilon = kEpsilon_;
    p.num_classes = (int32_t)N;
    p.reuse_features = reuse_features_ ? 1 : 0;
    detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
  }
  void upload(int which, const PointCloudPtr& cloud) {
    sicp_handle h = engine_.get();
    configure(h);
    detail::check(detail::set_cloud(h, which, *cloud), h, "sicp_set_cloud_strided");
  }

  int kCorrespondences_;
  double kEpsilon_;
  int outer_iter;
  bool cm_set_;
  bool reuse_features_ = false;
  Sophus::SE3d final_transformation_;
  PointCloudPtr source_cloud_, target_cloud_;
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // SEMANTIC_ICP_EM_ICP_H_
