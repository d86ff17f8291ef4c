// semantic_point_cloud.h -- semanticicp::SemanticPointCloud with the reference's public surface
// (reference: semantic_icp/semantic_point_cloud.h:15-63, impl/semantic_point_cloud.hpp).
// The per-point covariances of addSemanticCloud (impl :25-84) are computed by the GPU engine.
#ifndef SEMANTIC_POINT_CLOUD_H_
#define SEMANTIC_POINT_CLOUD_H_
#include <algorithm>
#include <map>
#include <memory>
#include <vector>

#include "sicp_engine.hpp"

namespace semanticicp {

template <typename PointT, typename SemanticT>
class SemanticPointCloud {
 public:
  typedef std::shared_ptr<SemanticPointCloud<PointT, SemanticT>> Ptr;
  typedef std::shared_ptr<const SemanticPointCloud<PointT, SemanticT>> ConstPtr;
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef pcl::KdTreeFLANN<PointT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d>> MatricesVector;
  typedef std::shared_ptr<MatricesVector> MatricesVectorPtr;

  SemanticPointCloud(int k = 20, double epsilon = 0.001) : k_correspondences_(k), epsilon_(epsilon) {}

  std::vector<SemanticT> semanticLabels;
  std::map<SemanticT, PointCloudPtr> labeledPointClouds;
  std::map<SemanticT, MatricesVectorPtr> labeledCovariances;
  std::map<SemanticT, KdTreePtr> labeledKdTrees;

  // reference: impl/semantic_point_cloud.hpp:12-86
  void addSemanticCloud(SemanticT label, PointCloudPtr cloud_ptr, bool computeKd = true, bool computeCov = true) {
    semanticLabels.push_back(label);
    labeledPointClouds[label] = cloud_ptr;
    if (!computeKd) return;
    KdTreePtr tree(new KdTree());
    tree->setInputCloud(cloud_ptr);
    labeledKdTrees[label] = tree;
    if (!computeCov) return;
    MatricesVectorPtr covs(new MatricesVector(cloud_ptr->size()));
    if (cloud_ptr->size() > 0) {
      sicp_handle h = engine_.get();
      sicp_params p;
      detail::check(sicp_default_params(SICP_MODE_GICP, &p), h, "sicp_default_params");
      p.k_cov = k_correspondences_;
      p.epsilon = epsilon_;
      detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
      detail::FlatCloud f = detail::flatten(*cloud_ptr);
      detail::check(sicp_set_cloud(h, SICP_SOURCE, f.size(), f.x.data(), f.y.data(), f.z.data(), nullptr), h, "sicp_set_cloud");
      std::vector<double> c9((size_t)f.size() * 9);
      detail::check(sicp_covariances(h, SICP_SOURCE, c9.data(), nullptr, nullptr, nullptr), h, "sicp_covariances");
      for (int i = 0; i < f.size(); ++i)
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) (*covs)[i](a, b) = c9[(size_t)i * 9 + 3 * a + b];
    }
    labeledCovariances[label] = covs;
  }

  // reference: semantic_point_cloud.h:44-52
  void removeSemanticClass(SemanticT label) {
    auto it = std::find(semanticLabels.begin(), semanticLabels.end(), label);
    if (it != semanticLabels.end()) {
      semanticLabels.erase(it);
      labeledPointClouds.erase(label);
      labeledCovariances.erase(label);
      labeledKdTrees.erase(label);
    }
  }

  // reference: impl/semantic_point_cloud.hpp:88-103
  typename pcl::PointCloud<pcl::PointXYZL>::Ptr getpclPointCloud() {
    typename pcl::PointCloud<pcl::PointXYZL>::Ptr out(new pcl::PointCloud<pcl::PointXYZL>());
    for (SemanticT s : semanticLabels)
      for (const PointT& p : *(labeledPointClouds[s])) {
        pcl::PointXYZL t;
        t.x = p.x; t.y = p.y; t.z = p.z; t.label = uint32_t(s);
        out->push_back(t);
      }
    return out;
  }

  // reference: impl/semantic_point_cloud.hpp:105-111
  void transform(Eigen::Matrix4f trans) {
    for (SemanticT s : semanticLabels) pcl::transformPointCloud(*(labeledPointClouds[s]), *(labeledPointClouds[s]), trans);
  }

  int getK() const { return k_correspondences_; }
  double getEpsilon() const { return epsilon_; }

 private:
  int k_correspondences_;
  double epsilon_;
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // SEMANTIC_POINT_CLOUD_H_
