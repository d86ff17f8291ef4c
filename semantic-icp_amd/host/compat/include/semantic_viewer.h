// semantic_viewer.h: the reference's PCLVisualizer wrapper (semantic_icp/semantic_viewer.h) is a GUI and
// out of scope for the registration engine.  Declared so that exec/test_icp.cc
// compiles; constructing one throws.
#ifndef SICP_COMPAT_INCLUDE_SEMANTIC_VIEWER_H_
#define SICP_COMPAT_INCLUDE_SEMANTIC_VIEWER_H_
#include <memory>
#include <stdexcept>
#include <string>

#include "semantic_point_cloud.h"

namespace semanticicp {
template <typename PointT, typename SemanticT>
class SemanticViewer {
 public:
  typedef std::shared_ptr<SemanticPointCloud<PointT, SemanticT>> SemanticCloudPtr;
  SemanticViewer() { throw std::runtime_error("semanticicp::SemanticViewer (PCLVisualizer GUI) is outside the MI355X engine's scope"); }
  void addSemanticPointCloud(const SemanticCloudPtr&, const std::string& = "") {}
  void addSemanticPointCloudSingleColor(const SemanticCloudPtr&, int, int, int, const std::string& = "") {}
  bool wasStopped() const { return true; }
};
}  // namespace semanticicp
#endif
