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

  void align(SemanticCloudPtr finalCloud, Sophus::SE3d& initTransform) {  // impl/semantic_icp.hpp:27-166
    sicp_handle h = engine_.get();
    sicp_params p;
    detail::check(sicp_default_params(SICP_MODE_SEMANTIC, &p), h, "sicp_default_params");
    p.k_cov = sourceCloud_->getK();
    p.epsilon = sourceCloud_->getEpsilon();
    detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
    detail::FlatCloud s = flatten(*sourceCloud_), t = flatten(*targetCloud_);
    detail::check(sicp_set_cloud(h, SICP_SOURCE, s.size(), s.x.data(), s.y.data(), s.z.data(), s.label.data()), h, "sicp_set_cloud");
    detail::check(sicp_set_cloud(h, SICP_TARGET, t.size(), t.x.data(), t.y.data(), t.z.data(), t.label.data()), h, "sicp_set_cloud");
    double out[7];
    detail::check(sicp_align(h, initTransform.data(), out, nullptr, nullptr), h, "sicp_align");
    finalTransformation_ = detail::to_se3(out);
    Eigen::Matrix4f mat = (finalTransformation_.matrix()).template cast<float>();  // :163-165
    finalCloud->transform(mat);
  }

  Sophus::SE3d getFinalTransFormation() { return finalTransformation_; }  // (sic) semantic_icp.h:58-62

 protected:
  // label clouds concatenated in semanticLabels order == the order the reference iterates them
  static detail::FlatCloud flatten(SemanticCloud& c) {
    detail::FlatCloud f;
    for (SemanticT s : c.semanticLabels)
      for (const PointT& p : *(c.labeledPointClouds[s])) f.push(p.x, p.y, p.z, (uint32_t)s);
    return f;
  }

  Sophus::SE3d finalTransformation_;
  SemanticCloudPtr sourceCloud_, targetCloud_;
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // SEMANTIC_ICP_H_
