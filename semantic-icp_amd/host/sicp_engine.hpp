// sicp_engine.hpp -- RAII holder of one C-ABI handle (include/sicp.h) for the class shims.
#ifndef SICP_HOST_ENGINE_HPP_
#define SICP_HOST_ENGINE_HPP_
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "sicp.h"

#if defined(SICP_HAVE_REAL_DEPS)
#include <Eigen/Core>
#include <pcl/kdtree/kdtree_flann.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <sophus/se3.hpp>
#else
#include "compat/eigen_lite.h"
#include "compat/pcl_lite.h"
#include "compat/sophus_lite.h"
#endif

namespace semanticicp {
namespace detail {

// The reference's methods return void and never throw; a registration that cannot run on the
// GPU must not look like a success, so failures surface as std::runtime_error.
inline void check(int status, sicp_handle h, const char* where) {
  if (status == SICP_OK) return;
  std::string msg = std::string(where) + ": " + sicp_strerror(status);
  if (status == SICP_ERR_HIP && h) msg += std::string(" -- ") + sicp_last_error(h);
  throw std::runtime_error(msg);
}

class Engine {
 public:
  Engine() = default;
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  ~Engine() { if (h_) sicp_destroy(h_); }
  sicp_handle get() {
    if (!h_) {
      const char* dev = std::getenv("SICP_DEVICE");  // one process / thread per GPU picks its device here
      check(sicp_create(dev ? std::atoi(dev) : 0, &h_), nullptr, "sicp_create");
    }
    return h_;
  }
 private:
  sicp_handle h_ = nullptr;
};

struct FlatCloud {
  std::vector<float> x, y, z;
  std::vector<uint32_t> label;
  void push(float px, float py, float pz, uint32_t l) { x.push_back(px); y.push_back(py); z.push_back(pz); label.push_back(l); }
  int size() const { return (int)x.size(); }
};

inline uint32_t label_of(const pcl::PointXYZ&) { return 0; }
inline uint32_t label_of(const pcl::PointXYZL& p) { return p.label; }

template <typename PointT>
FlatCloud flatten(const pcl::PointCloud<PointT>& c) {
  FlatCloud f;
  f.x.reserve(c.size()); f.y.reserve(c.size()); f.z.reserve(c.size()); f.label.reserve(c.size());
  for (const auto& p : c.points) f.push(p.x, p.y, p.z, label_of(p));
  return f;
}

// A PCL cloud goes to the engine as it lies in memory (sicp_set_cloud_strided: one pass, no SoA copies): x y z are
// the first three floats of every PCL point type used here, the label of a PointXYZL is a uint32 member.
inline const void* label_base(const pcl::PointCloud<pcl::PointXYZ>&) { return nullptr; }
inline const void* label_base(const pcl::PointCloud<pcl::PointXYZL>& c) { return c.points.empty() ? nullptr : (const void*)&c.points[0].label; }
template <typename PointT>
inline int set_cloud(sicp_handle h, int which, const pcl::PointCloud<PointT>& c, bool with_labels = true) {
  const void* xyz = c.points.empty() ? nullptr : (const void*)&c.points[0].x;
  return sicp_set_cloud_strided(h, which, (int32_t)c.points.size(), xyz, (int64_t)sizeof(PointT), with_labels ? label_base(c) : nullptr,
                                (int64_t)sizeof(PointT));
}
template <typename PointT>
inline int stream_add_cloud(sicp_stream s, const pcl::PointCloud<PointT>& c, bool with_labels, int64_t* id) {
  const void* xyz = c.points.empty() ? nullptr : (const void*)&c.points[0].x;
  return sicp_stream_add_cloud_strided(s, (int32_t)c.points.size(), xyz, (int64_t)sizeof(PointT), with_labels ? label_base(c) : nullptr,
                                       (int64_t)sizeof(PointT), id);
}

// sicp_covariances writes n row-major 3x3 matrices; an Eigen::Matrix3d is nine contiguous doubles, column-major -- and the
// engine's matrices are BIT-symmetric (upper triangle mirrored), so its rows are Eigen's columns: the storage of a
// std::vector<Eigen::Matrix3d> is the output array, no staging vector and no conversion pass
inline double* matrix3d_storage(Eigen::Matrix3d* m) {
  static_assert(sizeof(Eigen::Matrix3d) == 9 * sizeof(double), "Eigen::Matrix3d is nine doubles");
  return reinterpret_cast<double*>(m);
}

inline Sophus::SE3d to_se3(const double* qt) {
#if defined(SICP_HAVE_REAL_DEPS)
  Sophus::SE3d s;
  for (int i = 0; i < 7; ++i) s.data()[i] = qt[i];
  return s;
#else
  return Sophus::SE3d::fromData(qt);
#endif
}

}  // namespace detail
}  // namespace semanticicp
#endif
