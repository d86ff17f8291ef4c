// pcl_lite.h -- the part of PCL's point / cloud / search / PCD API the reference's class surface and
// drivers use: pcl::PointXYZ, pcl::PointXYZL, pcl::PointCloud<T>, pcl::KdTreeFLANN<T> (exact host
// nearestKSearch with FLANN's float arithmetic, built lazily: the engine's own spatial search lives on the
// GPU and never needs it, exec/nyu_metrics.h:56 and exec/roc_metrics.h:32 do), transformPointCloud, and
// PCD I/O for `x y z [label]` clouds: ASCII / binary / binary_compressed in, ASCII out
// (exec/kitti_eval.cc:132, exec/scenenet_eval.cc:198).  Used only when the real PCL is not installed.
#ifndef SICP_COMPAT_PCL_LITE_H_
#define SICP_COMPAT_PCL_LITE_H_
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
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
  typedef PointT PointType;
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
  void reserve(size_t n) { points.reserve(n); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
  iterator erase(iterator it) { auto r = points.erase(it); width = (uint32_t)points.size(); height = 1; return r; }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  PointT& at(size_t i) { return points.at(i); }
  const PointT& at(size_t i) const { return points.at(i); }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};

// pcl::KdTreeFLANN<PointT>: what the reference's classes pass around (exec/kitti_eval.cc:120-121,213) and what
// its label metrics search on the host (exec/nyu_metrics.h:47-56, exec/roc_metrics.h:27-32).  The search is
// exact with FLANN's L2_Simple<float> arithmetic -- d2 = ((dx*dx) + dy*dy) + dz*dz in float -- over the finite
// points of the cloud (PCL leaves non-finite points out of the index), results ascending by (d2, index): the
// order the engine's GPU search produces too.  The tree is built at the first
// nearestKSearch, not in setInputCloud: the registration classes hold these objects without ever searching them.
template <typename PointT>
class KdTreeFLANN {
 public:
  typedef std::shared_ptr<KdTreeFLANN<PointT>> Ptr;
  typedef std::shared_ptr<const KdTreeFLANN<PointT>> ConstPtr;
  typedef typename PointCloud<PointT>::ConstPtr PointCloudConstPtr;

  KdTreeFLANN(bool /*sorted*/ = true) {}
  void setInputCloud(const PointCloudConstPtr& c) {
    std::lock_guard<std::mutex> g(m_);
    cloud_ = c;
    built_ = false;
  }
  PointCloudConstPtr getInputCloud() const { return cloud_; }

  // neighbours of an arbitrary point; k is clamped to the number of indexed points (PCL does the same);
  // returns the number of neighbours written
  int nearestKSearch(const PointT& p, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    k_indices.clear();
    k_sqr_distances.clear();
    if (!cloud_ || k <= 0 || !std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) return 0;
    build();
    if (k > (int)order_.size()) k = (int)order_.size();
    if (k == 0) return 0;
    std::vector<uint64_t> best((size_t)k, ~uint64_t(0));  // ascending keys = float_bits(d2) << 32 | index
    const float q[3] = {p.x, p.y, p.z};
    search(0, q, best);
    k_indices.resize((size_t)k);
    k_sqr_distances.resize((size_t)k);
    for (int i = 0; i < k; ++i) {
      const uint32_t bits = (uint32_t)(best[(size_t)i] >> 32);
      float d;
      std::memcpy(&d, &bits, 4);
      k_indices[(size_t)i] = (int)(uint32_t)best[(size_t)i];
      k_sqr_distances[(size_t)i] = d;
    }
    return k;
  }
  int nearestKSearch(int index, int k, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances) const {
    return nearestKSearch(cloud_->points.at((size_t)index), k, k_indices, k_sqr_distances);
  }

 private:
  struct Node {
    int32_t lo, hi;       // leaf: points order_[lo..hi)
    int32_t left, right;  // inner: children; leaf: -1
    int32_t dim;
    float lo_max, hi_min;  // largest coordinate on the left, smallest on the right, along dim
  };
  static float coord(const PointT& p, int d) { return d == 0 ? p.x : d == 1 ? p.y : p.z; }

  void build() const {
    std::lock_guard<std::mutex> g(m_);
    if (built_) return;
    order_.clear();
    nodes_.clear();
    const auto& pts = cloud_->points;
    for (size_t i = 0; i < pts.size(); ++i)
      if (std::isfinite(pts[i].x) && std::isfinite(pts[i].y) && std::isfinite(pts[i].z)) order_.push_back((int32_t)i);
    if (!order_.empty()) split(0, (int32_t)order_.size());
    built_ = true;
  }
  int32_t split(int32_t lo, int32_t hi) const {
    const int32_t id = (int32_t)nodes_.size();
    nodes_.push_back(Node{lo, hi, -1, -1, 0, 0.f, 0.f});
    if (hi - lo <= 12) return id;
    const auto& pts = cloud_->points;
    float mn[3], mx[3];
    for (int d = 0; d < 3; ++d) mn[d] = mx[d] = coord(pts[(size_t)order_[(size_t)lo]], d);
    for (int32_t i = lo + 1; i < hi; ++i)
      for (int d = 0; d < 3; ++d) {
        const float v = coord(pts[(size_t)order_[(size_t)i]], d);
        mn[d] = std::min(mn[d], v);
        mx[d] = std::max(mx[d], v);
      }
    int dim = 0;
    for (int d = 1; d < 3; ++d)
      if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    if (!(mx[dim] > mn[dim])) return id;  // all points coincide: one (large) leaf
    const int32_t mid = lo + (hi - lo) / 2;
    std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi, [&](int32_t a, int32_t b) {
      const float va = coord(pts[(size_t)a], dim), vb = coord(pts[(size_t)b], dim);
      return va < vb || (va == vb && a < b);
    });
    float lo_max = coord(pts[(size_t)order_[(size_t)lo]], dim), hi_min = coord(pts[(size_t)order_[(size_t)mid]], dim);
    for (int32_t i = lo; i < mid; ++i) lo_max = std::max(lo_max, coord(pts[(size_t)order_[(size_t)i]], dim));
    for (int32_t i = mid; i < hi; ++i) hi_min = std::min(hi_min, coord(pts[(size_t)order_[(size_t)i]], dim));
    const int32_t l = split(lo, mid);
    const int32_t r = split(mid, hi);
    Node& n = nodes_[(size_t)id];
    n.left = l; n.right = r; n.dim = dim; n.lo_max = lo_max; n.hi_min = hi_min;
    return id;
  }
  // A subtree is skipped only when a LOWER BOUND of the float distance of all its points exceeds the current
  // k-th key's distance: along one axis, fl(fl(q - edge)^2) <= fl-distance of every point beyond the edge (float
  // subtraction, squaring and the addition of non-negative terms are monotonic), so no neighbour -- not even an
  // exact tie with a lower index -- is lost.
  void search(int32_t id, const float* q, std::vector<uint64_t>& best) const {
    const Node& n = nodes_[(size_t)id];
    const auto& pts = cloud_->points;
    if (n.left < 0) {
      for (int32_t i = n.lo; i < n.hi; ++i) {
        const int32_t idx = order_[(size_t)i];
        const PointT& p = pts[(size_t)idx];
        const float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
        float d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        uint32_t bits;
        std::memcpy(&bits, &d, 4);
        const uint64_t key = ((uint64_t)bits << 32) | (uint32_t)idx;
        if (key < best.back()) {
          size_t j = best.size() - 1;
          while (j > 0 && best[j - 1] > key) { best[j] = best[j - 1]; --j; }
          best[j] = key;
        }
      }
      return;
    }
    const float v = q[n.dim];
    const bool left_first = v <= n.lo_max || (v < n.hi_min && (v - n.lo_max) <= (n.hi_min - v));
    const int32_t first = left_first ? n.left : n.right, second = left_first ? n.right : n.left;
    search(first, q, best);
    const float edge = left_first ? n.hi_min : n.lo_max;
    const float gap = left_first ? (edge - v) : (v - edge);
    if (gap > 0) {
      const float lb = gap * gap;
      const uint32_t wbits = (uint32_t)(best.back() >> 32);
      float worst;
      std::memcpy(&worst, &wbits, 4);
      if (best.back() != ~uint64_t(0) && lb > worst) return;
    }
    search(second, q, best);
  }

  PointCloudConstPtr cloud_;
  mutable std::mutex m_;
  mutable bool built_ = false;
  mutable std::vector<int32_t> order_;
  mutable std::vector<Node> nodes_;
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
  tmp.width = in.width * in.height == in.size() ? in.width : (uint32_t)tmp.points.size();
  tmp.height = in.width * in.height == in.size() ? in.height : 1;
  tmp.is_dense = in.is_dense;
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

// LZF decompression (Marc Lehmann's liblzf format, what PCL's binary_compressed PCD bodies use): a control
// byte < 32 starts a run of ctrl+1 literal bytes; otherwise a back reference of length (ctrl >> 5) + 2
// (length 7 takes one more length byte) at distance ((ctrl & 31) << 8 | next byte) + 1, copied byte by
// byte (it may overlap its own output).  Returns the number of bytes written, 0 on malformed input.
inline size_t lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      ++ctrl;
      if (op + ctrl > out_len || ip + ctrl > in_len) return 0;
      std::memcpy(out + op, in + ip, ctrl);
      op += ctrl;
      ip += ctrl;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7) {
        if (ip >= in_len) return 0;
        len += in[ip++];
      }
      if (ip >= in_len) return 0;
      const size_t dist = ((size_t)(ctrl & 0x1f) << 8 | in[ip++]) + 1;
      len += 2;
      if (dist > op || op + len > out_len) return 0;
      for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];
    }
  }
  return op;
}
}  // namespace detail

// Reads FIELDS containing x y z (each `F 4`, COUNT 1) and optionally label (`U 4` or `I 4`, COUNT 1); other
// fields are skipped.  DATA ascii | binary | binary_compressed (per-field planes behind two uint32 sizes, LZF).
// Returns 0 on success, -1 on failure (like pcl::io::loadPCDFile) -- including coordinate or label fields of any
// other type or width, which this reader does not convert.
template <typename PointT>
int loadPCDFile(const std::string& path, PointCloud<PointT>& cloud) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return -1;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t npoints = 0, width = 0, height = 0;
  bool have_points = false;
  std::string line, data_kind;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS" || key == "COLUMNS") { std::string s; while (ss >> s) fields.push_back(s); }
    else if (key == "SIZE") { int v; while (ss >> v) sizes.push_back(v); }
    else if (key == "TYPE") { char c; while (ss >> c) types.push_back(c); }
    else if (key == "COUNT") { int v; while (ss >> v) counts.push_back(v); }
    else if (key == "WIDTH") { ss >> width; }
    else if (key == "HEIGHT") { ss >> height; }
    else if (key == "POINTS") { ss >> npoints; have_points = true; }
    else if (key == "DATA") { ss >> data_kind; break; }
  }
  if (!have_points) npoints = width * (height ? height : 1);
  const size_t nf = fields.size();
  if (nf == 0 || sizes.size() != nf || types.size() != nf) return -1;
  if (counts.size() != nf) counts.assign(nf, 1);
  int ix = -1, iy = -1, iz = -1, il = -1;
  for (size_t k = 0; k < nf; ++k) {
    if (sizes[k] <= 0 || counts[k] < 0) return -1;
    if (fields[k] == "x") ix = (int)k; else if (fields[k] == "y") iy = (int)k;
    else if (fields[k] == "z") iz = (int)k; else if (fields[k] == "label") il = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) return -1;
  for (int k : {ix, iy, iz})
    if (types[(size_t)k] != 'F' || sizes[(size_t)k] != 4 || counts[(size_t)k] != 1) return -1;
  if (!detail::has_label<PointT>::value) il = -1;
  if (il >= 0 && ((types[(size_t)il] != 'U' && types[(size_t)il] != 'I') || sizes[(size_t)il] != 4 || counts[(size_t)il] != 1)) return -1;
  cloud.clear();
  cloud.points.reserve(npoints);
  if (data_kind == "ascii") {
    for (size_t n = 0; n < npoints && std::getline(f, line); ++n) {
      std::istringstream ss(line);
      PointT p;
      for (size_t k = 0; k < nf; ++k)
        for (int c = 0; c < counts[k]; ++c) {
          std::string tok;
          if (!(ss >> tok)) return -1;
          if ((int)k == il) { detail::set_label(p, (uint32_t)std::strtoll(tok.c_str(), nullptr, 10)); continue; }
          if ((int)k != ix && (int)k != iy && (int)k != iz) continue;
          const float v = std::strtof(tok.c_str(), nullptr);  // "nan" / "inf" included, as PCL's reader accepts them
          if ((int)k == ix) p.x = v; else if ((int)k == iy) p.y = v; else p.z = v;
        }
      cloud.points.push_back(p);
    }
  } else if (data_kind == "binary") {
    size_t stride = 0;
    std::vector<size_t> off(nf);
    for (size_t k = 0; k < nf; ++k) { off[k] = stride; stride += (size_t)sizes[k] * counts[k]; }
    std::vector<char> buf(stride);
    for (size_t n = 0; n < npoints && f.read(buf.data(), (std::streamsize)stride); ++n) {
      PointT p;
      std::memcpy(&p.x, &buf[off[ix]], 4); std::memcpy(&p.y, &buf[off[iy]], 4); std::memcpy(&p.z, &buf[off[iz]], 4);
      if (il >= 0) { uint32_t l; std::memcpy(&l, &buf[off[il]], 4); detail::set_label(p, l); }
      cloud.points.push_back(p);
    }
  } else if (data_kind == "binary_compressed") {
    uint32_t csize = 0, usize = 0;
    if (!f.read((char*)&csize, 4) || !f.read((char*)&usize, 4)) return -1;
    size_t stride = 0;
    std::vector<size_t> plane(nf);  // the fields lie one after the other, each as npoints consecutive values
    for (size_t k = 0; k < nf; ++k) { plane[k] = stride * npoints; stride += (size_t)sizes[k] * counts[k]; }
    if ((size_t)usize != stride * npoints) return -1;
    std::vector<unsigned char> in(csize), out(usize);
    if (csize && !f.read((char*)in.data(), csize)) return -1;
    if (detail::lzf_decompress(in.data(), csize, out.data(), usize) != usize) return -1;
    cloud.points.resize(npoints);
    for (size_t n = 0; n < npoints; ++n) {
      PointT& p = cloud.points[n];
      std::memcpy(&p.x, &out[plane[ix] + 4 * n], 4); std::memcpy(&p.y, &out[plane[iy] + 4 * n], 4); std::memcpy(&p.z, &out[plane[iz] + 4 * n], 4);
      if (il >= 0) { uint32_t l; std::memcpy(&l, &out[plane[il] + 4 * n], 4); detail::set_label(p, l); }
    }
  } else {
    return -1;
  }
  if (cloud.points.size() != npoints) { cloud.clear(); return -1; }
  const bool organised = width * height == npoints && height > 0;
  cloud.width = (uint32_t)(organised ? width : npoints);
  cloud.height = (uint32_t)(organised ? height : 1);
  cloud.is_dense = true;
  for (const auto& p : cloud.points)
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) { cloud.is_dense = false; break; }
  return 0;
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
