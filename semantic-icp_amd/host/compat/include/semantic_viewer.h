// semantic_viewer.h: the reference's PCLVisualizer wrapper (semantic_icp/semantic_viewer.h) is a GUI and
// out of scope for the registration engine.  Declared so that exec/test_icp.cc
// compiles, links and runs to its end without a display: nothing is drawn and wasStopped() is true at once.
#ifndef SICP_COMPAT_INCLUDE_SEMANTIC_VIEWER_H_
#define SICP_COMPAT_INCLUDE_SEMANTIC_VIEWER_H_
#include <cstdio>
#include <memory>
#include <string>

#include "semantic_point_cloud.h"

namespace semanticicp {
template <typename PointT, typename SemanticT>
class SemanticViewer {
 public:
  typedef std::shared_ptr<SemanticPointCloud<PointT, SemanticT>> SemanticCloudPtr;
  SemanticViewer() { std::fprintf(stderr, "[sicp compat] semanticicp::SemanticViewer (PCLVisualizer GUI) is outside the MI355X engine's scope: nothing is shown\n"); }
  void addSemanticPointCloud(const SemanticCloudPtr&, const std::string& = "") {}
  void addSemanticPointCloudSingleColor(const SemanticCloudPtr&, int, int, int, const std::string& = "") {}
  bool wasStopped() const { return true; }
};
}  // namespace semanticicp
#endif
