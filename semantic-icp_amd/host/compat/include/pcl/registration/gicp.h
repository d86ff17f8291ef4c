// <pcl/registration/gicp.h>: pcl::GeneralizedIterativeClosestPoint is PCL's own (third-party) GICP, which
// the reference's drivers run beside semanticicp::GICP as a comparison column (exec/test_icp.cc:107-113,
// exec/kitti_eval.cc:229-247, nyu_eval.cc:193-210, scenenet_eval.cc:229-247, roc_eval.cc:160-176).  It is NOT
// part of the path this engine replaces (out of scope).  So that those drivers still RUN
// end to end without PCL, the stand-in does no registration at all -- and what it reports cannot be mistaken for one:
// getFinalTransformation() is all NaN (so every number a driver derives from it, the error columns of its PCL-GICP
// result file included, is NaN), hasConverged() is false, and align() says so once on stderr.  The output cloud is the
// source moved by the initial guess.  (SICP_PCL_GICP_RETURNS_GUESS=1 in the environment: the guess instead of NaN, the
// behaviour of round 5, for scripts that need a finite matrix there.)  Build against real PCL to get PCL's numbers.
#ifndef SICP_COMPAT_INCLUDE_PCL_REGISTRATION_GICP_H_
#define SICP_COMPAT_INCLUDE_PCL_REGISTRATION_GICP_H_
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "pcl/point_types.h"

namespace pcl {
template <typename PointSource, typename PointTarget>
class GeneralizedIterativeClosestPoint {
 public:
  typedef typename PointCloud<PointSource>::Ptr PointCloudSourcePtr;
  typedef typename PointCloud<PointTarget>::Ptr PointCloudTargetPtr;
  void setInputCloud(const PointCloudSourcePtr& c) { source_ = c; }
  void setInputSource(const PointCloudSourcePtr& c) { source_ = c; }
  void setInputTarget(const PointCloudTargetPtr&) {}
  void setMaxCorrespondenceDistance(double) {}
  void setMaximumIterations(int) {}
  void align(PointCloud<PointSource>& out) { align(out, Eigen::Matrix4f::Identity()); }
  void align(PointCloud<PointSource>& out, const Eigen::Matrix4f& guess) {
    static bool told = false;
    if (!told) {
      told = true;
      std::fprintf(stderr, "[sicp compat] pcl::GeneralizedIterativeClosestPoint is third-party PCL code outside the MI355X engine: "
                           "the stand-in registers nothing and reports an all-NaN transformation (build against real PCL for PCL's own GICP column)\n");
    }
    if (std::getenv("SICP_PCL_GICP_RETURNS_GUESS")) {
      final_ = guess;
    } else {
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) final_(r, c) = std::numeric_limits<float>::quiet_NaN();
    }
    if (source_) transformPointCloud(*source_, out, guess);
  }
  bool hasConverged() const { return false; }
  Eigen::Matrix4f getFinalTransformation() const { return final_; }
 private:
  PointCloudSourcePtr source_;
  Eigen::Matrix4f final_ = Eigen::Matrix4f::Identity();
};
}  // namespace pcl
#endif
