// pcl_lite.h -- the part of PCL's point / cloud / PCD API the reference's class surface and
// drivers use: pcl::PointXYZ, pcl::PointXYZL, pcl::PointCloud<T>, KdTreeFLANN<T> (as a token: the
// spatial search lives on the GPU), transformPointCloud, and ASCII / binary PCD I/O for
// `x y z [label]` clouds (exec/kitti_eval.cc:132, exec/scenenet_eval.cc:198).  Used only when the
// real PCL is not installed.
#ifndef SICP_COMPAT_PCL_LITE_H_
#define SICP_COMPAT_PCL_LITE_H_
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "eigen_lite.h"

namespace pcl {

struct PointXYZ {
  float x = 0, y = 0, z = 0;
  PointXYZ() = default;
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct PointXYZL {
  float x = 0, y = 0, z = 0;
  uint32_t label = 0;
};
inline std::ostream& operator<<(std::ostream& os, const PointXYZ& p) { return os << "(" << p.x << "," << p.y << "," << p.z << ")"; }
inline std::ostream& operator<<(std::ostream& os, const PointXYZL& p) { return os << "(" << p.x << "," << p.y << "," << p.z << " - " << p.label << ")"; }

template <typename PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  typedef typename std::vector<PointT>::iterator iterator;
  typedef typename std::vector<PointT>::const_iterator const_iterator;
  std::vector<PointT> points;
  uint32_t width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  void clear() { points.clear(); width = 0; height = 1; }
  void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  iterator erase(iterator it) { auto r = points.erase(it); width = (uint32_t)points.size(); height = 1; return r; }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  PointT& at(size_t i) { return points.at(i); }
  const PointT& at(size_t i) const { return points.at(i); }
};

// Token standing in for pcl::KdTreeFLANN<PointT>: the engine builds its own search structure on
// the GPU from the cloud; drivers only pass these pointers around (exec/kitti_eval.cc:120-121,213).
template <typename PointT>
class KdTreeFLANN {
 public:
  typedef std::shared_ptr<KdTreeFLANN<PointT>> Ptr;
  void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { cloud_ = c; }
  typename PointCloud<PointT>::Ptr getInputCloud() const { return cloud_; }
 private:
  typename PointCloud<PointT>::Ptr cloud_;
};

namespace detail {
inline void copy_extra(const PointXYZ&, PointXYZ&) {}
inline void copy_extra(const PointXYZL& a, PointXYZL& b) { b.label = a.label; }
}  // namespace detail

// pcl::transformPointCloud (PCL 1.8/1.9 form): rows evaluated left to right in Scalar, cast to float
template <typename PointT, typename Scalar>
void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix<Scalar, 4, 4>& M) {
  PointCloud<PointT> tmp;
  tmp.points.resize(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    const Scalar x = in[i].x, y = in[i].y, z = in[i].z;
    PointT p = in[i];
    p.x = static_cast<float>(M(0, 0) * x + M(0, 1) * y + M(0, 2) * z + M(0, 3));
    p.y = static_cast<float>(M(1, 0) * x + M(1, 1) * y + M(1, 2) * z + M(1, 3));
    p.z = static_cast<float>(M(2, 0) * x + M(2, 1) * y + M(2, 2) * z + M(2, 3));
    tmp.points[i] = p;
  }
  tmp.width = (uint32_t)tmp.points.size();
  tmp.height = 1;
  out = tmp;
}

namespace io {
namespace detail {
inline void set_label(PointXYZ&, uint32_t) {}
inline void set_label(PointXYZL& p, uint32_t l) { p.label = l; }
inline uint32_t get_label(const PointXYZ&) { return 0; }
inline uint32_t get_label(const PointXYZL& p) { return p.label; }
template <class P> struct has_label { static const bool value = false; };
template <> struct has_label<PointXYZL> { static const bool value = true; };
}  // namespace detail

// Reads FIELDS containing x y z (F 4) and optionally label (U 4); DATA ascii | binary.
// Returns 0 on success, -1 on failure (like pcl::io::loadPCDFile).
template <typename PointT>
int loadPCDFile(const std::string& path, PointCloud<PointT>& cloud) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return -1;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t npoints = 0;
  std::string line, data_kind;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS") { std::string s; while (ss >> s) fields.push_back(s); }
    else if (key == "SIZE") { int v; while (ss >> v) sizes.push_back(v); }
    else if (key == "TYPE") { char c; while (ss >> c) types.push_back(c); }
    else if (key == "COUNT") { int v; while (ss >> v) counts.push_back(v); }
    else if (key == "POINTS") { ss >> npoints; }
    else if (key == "DATA") { ss >> data_kind; break; }
  }
  const size_t nf = fields.size();
  if (nf == 0 || sizes.size() != nf || types.size() != nf) return -1;
  if (counts.size() != nf) counts.assign(nf, 1);
  int ix = -1, iy = -1, iz = -1, il = -1;
  for (size_t k = 0; k < nf; ++k) {
    if (fields[k] == "x") ix = (int)k; else if (fields[k] == "y") iy = (int)k;
    else if (fields[k] == "z") iz = (int)k; else if (fields[k] == "label") il = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) return -1;
  cloud.clear();
  cloud.points.reserve(npoints);
  if (data_kind == "ascii") {
    for (size_t n = 0; n < npoints && std::getline(f, line); ++n) {
      std::istringstream ss(line);
      PointT p;
      for (size_t k = 0; k < nf; ++k)
        for (int c = 0; c < counts[k]; ++c) {
          double v; ss >> v;
          if ((int)k == ix) p.x = (float)v; else if ((int)k == iy) p.y = (float)v;
          else if ((int)k == iz) p.z = (float)v; else if ((int)k == il) detail::set_label(p, (uint32_t)v);
        }
      cloud.points.push_back(p);
    }
  } else if (data_kind == "binary") {
    size_t stride = 0;
    std::vector<size_t> off(nf);
    for (size_t k = 0; k < nf; ++k) { off[k] = stride; stride += (size_t)sizes[k] * counts[k]; }
    std::vector<char> buf(stride);
    for (size_t n = 0; n < npoints && f.read(buf.data(), stride); ++n) {
      PointT p;
      std::memcpy(&p.x, &buf[off[ix]], 4); std::memcpy(&p.y, &buf[off[iy]], 4); std::memcpy(&p.z, &buf[off[iz]], 4);
      if (il >= 0) { uint32_t l; std::memcpy(&l, &buf[off[il]], 4); detail::set_label(p, l); }
      cloud.points.push_back(p);
    }
  } else {
    return -1;  // binary_compressed is not supported
  }
  cloud.width = (uint32_t)cloud.points.size();
  cloud.height = 1;
  return cloud.points.size() == npoints ? 0 : -1;
}

template <typename PointT>
int savePCDFileASCII(const std::string& path, const PointCloud<PointT>& cloud) {
  std::ofstream f(path);
  if (!f) return -1;
  const bool L = detail::has_label<PointT>::value;
  f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n";
  f << (L ? "FIELDS x y z label\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n" : "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n");
  f << "WIDTH " << cloud.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.size() << "\nDATA ascii\n";
  f.precision(9);
  for (const auto& p : cloud.points) {
    f << p.x << " " << p.y << " " << p.z;
    if (L) f << " " << detail::get_label(p);
    f << "\n";
  }
  return f ? 0 : -1;
}
}  // namespace io
}  // namespace pcl
#endif
