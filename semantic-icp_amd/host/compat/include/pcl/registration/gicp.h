// <pcl/registration/gicp.h>: pcl::GeneralizedIterativeClosestPoint is PCL's own (third-party) GICP, which
// the reference's drivers run beside semanticicp::GICP as a comparison (exec/test_icp.cc:107-113,
// exec/kitti_eval.cc:229-247).  It is NOT part of the path this engine replaces (SURVEY.md section 2:
// out of scope): the declaration below only lets those drivers compile; calling align() throws.
#ifndef SICP_COMPAT_INCLUDE_PCL_REGISTRATION_GICP_H_
#define SICP_COMPAT_INCLUDE_PCL_REGISTRATION_GICP_H_
#include <stdexcept>

#include "pcl/point_types.h"

namespace pcl {
template <typename PointSource, typename PointTarget>
class GeneralizedIterativeClosestPoint {
 public:
  typedef typename PointCloud<PointSource>::Ptr PointCloudSourcePtr;
  typedef typename PointCloud<PointTarget>::Ptr PointCloudTargetPtr;
  void setInputCloud(const PointCloudSourcePtr&) {}
  void setInputSource(const PointCloudSourcePtr&) {}
  void setInputTarget(const PointCloudTargetPtr&) {}
  void setMaxCorrespondenceDistance(double) {}
  void setMaximumIterations(int) {}
  void align(PointCloud<PointSource>&) { out_of_scope(); }
  void align(PointCloud<PointSource>&, const Eigen::Matrix4f&) { out_of_scope(); }
  Eigen::Matrix4f getFinalTransformation() const { return Eigen::Matrix4f::Identity(); }
 private:
  static void out_of_scope() {
    throw std::runtime_error("pcl::GeneralizedIterativeClosestPoint is third-party PCL code outside the MI355X engine's scope; build against real PCL to run it");
  }
};
}  // namespace pcl
#endif
