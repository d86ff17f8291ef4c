// pcl_2_semantic.h -- split an XYZL cloud by label, first-seen label order
// (reference: semantic_icp/pcl_2_semantic.h:14-42; `inline` added: the reference defines a
// non-inline function in a header).
#ifndef PCL_2_SEMANTIC_H_
#define PCL_2_SEMANTIC_H_
#include <map>
#include <memory>
#include <vector>

#include "semantic_point_cloud.h"

namespace semanticicp {

inline void pcl_2_semantic(const pcl::PointCloud<pcl::PointXYZL>::Ptr pclCloud,
                           std::shared_ptr<SemanticPointCloud<pcl::PointXYZ, uint32_t>> semanticCloud) {
  typedef pcl::PointCloud<pcl::PointXYZ> PointCloud;
  typedef PointCloud::Ptr PointCloudPtr;
  std::vector<uint32_t> labels;
  std::map<uint32_t, PointCloudPtr> map;
  for (const pcl::PointXYZL& p : pclCloud->points) {
    auto it = map.find(p.label);
    if (it == map.end()) {
      PointCloudPtr cloud(new PointCloud());
      cloud->push_back(pcl::PointXYZ(p.x, p.y, p.z));
      map[p.label] = cloud;
      labels.push_back(p.label);
    } else {
      it->second->push_back(pcl::PointXYZ(p.x, p.y, p.z));
    }
  }
  for (uint32_t l : labels) semanticCloud->addSemanticCloud(l, map[l]);
}

}  // namespace semanticicp
#endif  // PCL_2_SEMANTIC_H_
