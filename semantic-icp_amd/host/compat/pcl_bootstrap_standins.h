// pcl_bootstrap_standins.h -- declarations of the PCL feature / RANSAC types that exec/bootstrap.h names
// (class Bootstrap: VoxelGrid down-sampling, normals, FPFH features, SAC-IA initial alignment;
// exec/bootstrap.h:20-134).  Every driver includes that header, and every call site of Bootstrap is commented
// out in the reference (exec/kitti_eval.cc:166-167, nyu_eval.cc:127-128, scenenet_eval.cc:161-162): the initial
// guess is the identity.  The feature stack is third-party PCL code outside the registration path (SURVEY.md
// section 2), so these types exist only to let the drivers compile and link unchanged
// without PCL; using one of them at run time throws.
#ifndef SICP_COMPAT_PCL_BOOTSTRAP_STANDINS_H_
#define SICP_COMPAT_PCL_BOOTSTRAP_STANDINS_H_
#include <memory>
#include <stdexcept>
#include <vector>

#include "pcl_lite.h"

namespace pcl {
namespace detail {
[[noreturn]] inline void bootstrap_out_of_scope(const char* what) {
  throw std::runtime_error(std::string(what) + " is third-party PCL code outside the MI355X engine's scope (exec/bootstrap.h is never "
                                               "called by the reference's drivers); build against real PCL to run it");
}
}  // namespace detail

struct Normal { float normal_x = 0, normal_y = 0, normal_z = 0, curvature = 0; };
struct FPFHSignature33 { float histogram[33] = {0}; };
struct Correspondence { int index_query = 0, index_match = -1; float distance = 0; };
typedef std::vector<Correspondence> Correspondences;
typedef std::shared_ptr<Correspondences> CorrespondencesPtr;

namespace search {
template <typename PointT>
class KdTree {
 public:
  typedef std::shared_ptr<KdTree<PointT>> Ptr;
  KdTree(bool = true) {}
};
}  // namespace search

template <typename PointT>
class VoxelGrid {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr&) {}
  void setLeafSize(float, float, float) {}
  void filter(PointCloud<PointT>&) { detail::bootstrap_out_of_scope("pcl::VoxelGrid"); }
};

template <typename PointInT, typename PointOutT>
class NormalEstimation {
 public:
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr&) {}
  void setSearchMethod(const typename search::KdTree<PointInT>::Ptr&) {}
  void setRadiusSearch(double) {}
  void setKSearch(int) {}
  void compute(PointCloud<PointOutT>&) { detail::bootstrap_out_of_scope("pcl::NormalEstimation"); }
};

template <typename PointInT, typename PointNT, typename PointOutT = FPFHSignature33>
class FPFHEstimation {
 public:
  typedef std::shared_ptr<FPFHEstimation<PointInT, PointNT, PointOutT>> Ptr;
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr&) {}
  void setInputNormals(const typename PointCloud<PointNT>::ConstPtr&) {}
  void setSearchMethod(const typename search::KdTree<PointInT>::Ptr&) {}
  void setRadiusSearch(double) {}
  void compute(PointCloud<PointOutT>&) { detail::bootstrap_out_of_scope("pcl::FPFHEstimation"); }
};

template <typename PointSource, typename PointTarget, typename FeatureT>
class SampleConsensusInitialAlignment {
 public:
  void setInputSource(const typename PointCloud<PointSource>::ConstPtr&) {}
  void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr&) {}
  void setSourceFeatures(const typename PointCloud<FeatureT>::ConstPtr&) {}
  void setTargetFeatures(const typename PointCloud<FeatureT>::ConstPtr&) {}
  void setMinSampleDistance(float) {}
  void setMaxCorrespondenceDistance(double) {}
  void setMaximumIterations(int) {}
  void align(PointCloud<PointSource>&) { detail::bootstrap_out_of_scope("pcl::SampleConsensusInitialAlignment"); }
  Eigen::Matrix4f getFinalTransformation() const { return Eigen::Matrix4f::Identity(); }
};

template <typename PointSource, typename FeatureT>
class MultiscaleFeaturePersistence {
 public:
  MultiscaleFeaturePersistence() { detail::bootstrap_out_of_scope("pcl::MultiscaleFeaturePersistence"); }
};

namespace registration {
template <typename PointSource, typename PointTarget, typename Scalar = float>
class CorrespondenceEstimation {
 public:
  void determineCorrespondences(Correspondences&, double = 0) { detail::bootstrap_out_of_scope("pcl::registration::CorrespondenceEstimation"); }
};
template <typename PointT>
class CorrespondenceRejectorSampleConsensus {
 public:
  CorrespondenceRejectorSampleConsensus() { detail::bootstrap_out_of_scope("pcl::registration::CorrespondenceRejectorSampleConsensus"); }
};
}  // namespace registration
}  // namespace pcl
#endif
