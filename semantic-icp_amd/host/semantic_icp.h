// semantic_icp.h -- semanticicp::SemanticIterativeClosestPoint<PointT, SemanticT> with the
// reference's public surface (reference: semantic_icp/semantic_icp.h:15-83,
// impl/semantic_icp.hpp:20-166).  align() runs on the MI355X engine.  The unreachable pose-fusion
// members of the reference (impl/semantic_icp.hpp:170-265) are not reproduced.
#ifndef SEMANTIC_ICP_H_
#define SEMANTIC_ICP_H_
#include <memory>
#include <vector>

#include "semantic_point_cloud.h"
#include "sicp_engine.hpp"

namespace semanticicp {

template <typename PointT, typename SemanticT>
class SemanticIterativeClosestPoint {
 public:
  typedef SemanticPointCloud<PointT, SemanticT> SemanticCloud;
  typedef typename std::shared_ptr<SemanticCloud> SemanticCloudPtr;
  typedef typename std::shared_ptr<const SemanticCloud> SemanticCloudConstPtr;
  typedef std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d>> MatricesVector;
  typedef std::shared_ptr<MatricesVector> MatricesVectorPtr;
  typedef pcl::KdTreeFLANN<PointT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef Eigen::Matrix<double, 6, 1> Vector6d;

  SemanticIterativeClosestPoint() {}

  inline void setInputSource(const SemanticCloudPtr& cloud) { sourceCloud_ = cloud; }  // semantic_icp.h:41-44
  inline void setInputTarget(const SemanticCloudPtr& cloud) { targetCloud_ = cloud; }  // semantic_icp.h:46-49

  void align(SemanticCloudPtr finalCloud) {  // impl/semantic_icp.hpp:20-25
    Sophus::SE3d init;
    align(finalCloud, init);
  }

  // impl/semantic_icp.hpp:27-166.  The two clouds are registered where they already are: each SemanticPointCloud keeps its
  // label clouds on the GPU (uploaded at its first use) and this handle refers to them (sicp_share_cloud) -- nothing is
  // flattened, uploaded or indexed again per align(), and the per-label covariances are the ones computed once per cloud
  // (reuse_features; impl/semantic_icp.hpp:73,77 read them out of the cloud objects the same way).
  void align(SemanticCloudPtr finalCloud, Sophus::SE3d& initTransform) {
    sicp_handle h = engine_.get();
    sicp_params p;
    detail::check(sicp_default_params(SICP_MODE_SEMANTIC, &p), h, "sicp_default_params");
    p.k_cov = sourceCloud_->getK();
    p.epsilon = sourceCloud_->getEpsilon();
    p.reuse_features = 1;
    detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
    detail::check(sicp_share_cloud(h, SICP_SOURCE, sourceCloud_->device(), SICP_SOURCE), h, "sicp_share_cloud");
    detail::check(sicp_share_cloud(h, SICP_TARGET, targetCloud_->device(), SICP_SOURCE), h, "sicp_share_cloud");
    double out[7];
    detail::check(sicp_align(h, initTransform.data(), out, nullptr, nullptr), h, "sicp_align");
    finalTransformation_ = detail::to_se3(out);
    Eigen::Matrix4f mat = (finalTransformation_.matrix()).template cast<float>();  // :163-165
    finalCloud->transform(mat);
  }

  Sophus::SE3d getFinalTransFormation() { return finalTransformation_; }  // (sic) semantic_icp.h:58-62

 protected:
  Sophus::SE3d finalTransformation_;
  SemanticCloudPtr sourceCloud_, targetCloud_;
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // SEMANTIC_ICP_H_
