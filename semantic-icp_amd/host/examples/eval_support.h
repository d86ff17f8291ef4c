// eval_support.h -- host helpers of the headless eval drivers: the pieces of exec/csv.h,
// exec/read_confusion_matrix.h, exec/filter_range.h and exec/kitti_metrics.h that the drivers need,
// re-implemented on the compat types (reference definitions cited at each function).
#ifndef SICP_EVAL_SUPPORT_H_
#define SICP_EVAL_SUPPORT_H_
#include <dirent.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sicp_engine.hpp"

namespace evalsupport {

// rows of space separated cells (exec/csv.h:17-47 splits on ' ')
inline std::vector<std::vector<std::string>> read_rows(const std::string& path) {
  std::vector<std::vector<std::string>> rows;
  std::ifstream f(path);
  std::string line;
  while (std::getline(f, line)) {
    std::vector<std::string> cells;
    std::stringstream ss(line);
    std::string cell;
    while (std::getline(ss, cell, ' ')) if (!cell.empty()) cells.push_back(cell);
    if (!cells.empty()) rows.push_back(cells);
  }
  return rows;
}

// exec/read_confusion_matrix.h:6-19
template <int N>
Eigen::Matrix<double, N, N> ReadConfusionMatrix(const std::string& file_name) {
  Eigen::Matrix<double, N, N> out;
  const auto rows = read_rows(file_name);
  for (size_t j = 0; j < rows.size() && j < (size_t)N; ++j)
    for (size_t k = 0; k < rows[j].size() && k < (size_t)N; ++k) out((int)j, (int)k) = std::stod(rows[j][k]);
  return out;
}

// exec/filter_range.h:6-18: drop points farther than `range` (strict >), order preserved
inline void filterRange(pcl::PointCloud<pcl::PointXYZL>::Ptr cloud, const double range) {
  std::vector<pcl::PointXYZL> keep;
  keep.reserve(cloud->size());
  for (const auto& pt : cloud->points)
    if (!((pt.x * pt.x + pt.y * pt.y + pt.z * pt.z) > range * range)) keep.push_back(pt);
  cloud->points.swap(keep);
  cloud->width = (uint32_t)cloud->points.size();
  cloud->height = 1;
}

// exec/kitti_eval.cc:25-43
inline std::vector<std::string> get_pcd_in_dir(const std::string& dir_name) {
  std::vector<std::string> out;
  if (DIR* d = opendir(dir_name.c_str())) {
    while (struct dirent* e = readdir(d)) {
      const size_t len = std::strlen(e->d_name);
      if (len >= 4 && std::strcmp(e->d_name + len - 4, ".pcd") == 0) out.push_back(dir_name + "/" + e->d_name);
    }
    closedir(d);
  }
  std::sort(out.begin(), out.end());  // exec/kitti_eval.cc:87
  return out;
}

// exec/kitti_metrics.h:6-118: ground-truth poses (12 numbers per row -> fitToSE3), the error of an
// estimate against poseA^-1 * poseB, and one CSV row per evaluation in the reference's column order:
//   idA, idB, ||log(dT)||^2, ||rot log(dT)||^2, ||trans(dT)||^2, time, dT (16, row major), T (16), outer
// (time is written in seconds as a real number; the reference truncates it to whole seconds)
//
// The same class serves exec/scenenet_metrics.h:10-75, whose ground-truth rows hold 16 numbers
// (4x4, row major) + the frame index and are stored INVERTED (scenenet_metrics.h:17-29).
class KittiMetrics {
 public:
  enum Format { KITTI_3x4 = 0, SCENENET_4x4_INVERTED = 1 };
  explicit KittiMetrics(const std::string& gtFileName, std::ostream* out = &std::cout, Format fmt = KITTI_3x4) : out_(out) {
    for (const auto& row : read_rows(gtFileName)) {
      Eigen::Matrix4d mat = Eigen::Matrix4d::Identity();
      const size_t want = fmt == KITTI_3x4 ? 12 : 16;
      for (size_t n = 0; n < row.size() && n < want; ++n) mat((int)(n / 4), (int)(n % 4)) = std::stod(row[n]);
      const Sophus::SE3d T = Sophus::SE3d::fitToSE3(mat);
      gtPoses_.push_back(fmt == KITTI_3x4 ? T : T.inverse());
    }
  }
  size_t numPoses() const { return gtPoses_.size(); }
  Sophus::SE3d getGTtransfrom(size_t poseIDA, size_t poseIDB) const { return gtPoses_[poseIDA].inverse() * gtPoses_[poseIDB]; }
  double evaluate(const Sophus::SE3d& transform, size_t poseIDA, size_t poseIDB, double timeSeconds, int outer_iter) {
    const Sophus::SE3d diff = getGTtransfrom(poseIDA, poseIDB) * transform.inverse();
    const double transformError = diff.log().squaredNorm();
    const double rotError = diff.so3().log().squaredNorm();
    const double transError = diff.translation().squaredNorm();
    transformMSE_ += transformError; rotMSE_ += rotError; transMSE_ += transError; count_++;
    std::ostream& o = *out_;
    o.precision(17);
    o << poseIDA << ", " << poseIDB << ", " << transformError << ", " << rotError << ", " << transError << ", " << timeSeconds;
    const Eigen::Matrix4d a = diff.matrix(), b = transform.matrix();
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o << ", " << a(i, j);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o << ", " << b(i, j);
    o << ", " << outer_iter << std::endl;
    return transformError;
  }
  double getTransformMSE() const { return transformMSE_ / double(count_); }
  double getRotMSE() const { return rotMSE_ / double(count_); }
  double getTransMSE() const { return transMSE_ / double(count_); }

 private:
  std::vector<Sophus::SE3d> gtPoses_;
  double transformMSE_ = 0, rotMSE_ = 0, transMSE_ = 0;
  size_t count_ = 0;
  std::ostream* out_;
};

// exec/nyu_metrics.h:8-113: label agreement between a registered source cloud and the target.
// For every source point the nearest target point (any label) is looked up; pairs closer than
// sqrt(25) (strict <, float compare) are tallied.  The reference searches a pcl::KdTreeFLANN on the
// host; here the lookup is one K = 1 correspondence search of the engine at the identity (the
// source passed in is already transformed, as in the reference).
class NYUMetrics {
 public:
  typedef pcl::PointXYZL PointT;
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef PointCloud::Ptr PointCloudPtr;

  NYUMetrics(const std::string& testFileName, std::string outName, size_t countStart = 0, size_t numClasses = 895)
      : numClasses_(numClasses), count_(countStart), outName_(std::move(outName)) {
    for (const auto& row : read_rows(testFileName)) {  // exec/nyu_metrics.h:25-31 (CSVIterator splits on ' ')
      std::vector<size_t> data;
      for (const std::string& cell : row) data.push_back((size_t)std::stoi(cell));
      testPairs_.push_back(data);
    }
    confusion_.assign(numClasses_ * numClasses_, 0);
    it_ = std::min(countStart, testPairs_.size());
  }

  double evaluate(PointCloudPtr source, PointCloudPtr target, const std::string& label) {
    const int num = std::stoi(label);
    std::ostringstream oss;
    oss << "Label" << num << "-";
    count_++;
    std::ofstream out(dir_of(outName_) + oss.str() + base_of(outName_));
    const int ns = (int)source->size(), nt = (int)target->size();
    std::vector<int32_t> idx(ns > 0 ? ns : 1);
    std::vector<float> d2(ns > 0 ? ns : 1);
    {
      sicp_handle h = engine_.get();
      sicp_params p;
      semanticicp::detail::check(sicp_default_params(SICP_MODE_GICP, &p), h, "sicp_default_params");
      p.knn = 1;
      p.gate_sq = 25.0;  // nyu_metrics.h:57
      semanticicp::detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
      semanticicp::detail::FlatCloud s = semanticicp::detail::flatten(*source), t = semanticicp::detail::flatten(*target);
      semanticicp::detail::check(sicp_set_cloud(h, SICP_SOURCE, ns, s.x.data(), s.y.data(), s.z.data(), nullptr), h, "sicp_set_cloud");
      semanticicp::detail::check(sicp_set_cloud(h, SICP_TARGET, nt, t.x.data(), t.y.data(), t.z.data(), nullptr), h, "sicp_set_cloud");
      const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
      semanticicp::detail::check(sicp_correspondences(h, ident, idx.data(), d2.data(), nullptr), h, "sicp_correspondences");
    }
    double inlier = 0, total = 0, dist = 0;
    for (int i = 0; i < ns; ++i) {
      if (idx[i] < 0) continue;  // gated out: nn_dist_sq >= 25
      const uint32_t labelSource = (*source)[i].label, labelTarget = (*target)[idx[i]].label;
      if (labelSource < numClasses_ && labelTarget < numClasses_) confusion_[labelSource * numClasses_ + labelTarget]++;
      out << labelSource << ", " << labelTarget << std::endl;
      total++;
      dist += std::sqrt(d2[i]);
      if (labelSource == labelTarget) inlier++;
    }
    out.close();
    std::ofstream matrixOut(dir_of(outName_) + "Matrix" + base_of(outName_));  // Eigen's operator<<: rows, space separated
    for (size_t r = 0; r < numClasses_; ++r) {
      for (size_t c = 0; c < numClasses_; ++c) matrixOut << (c ? " " : "") << confusion_[r * numClasses_ + c];
      matrixOut << "\n";
    }
    matrixOut.close();
    std::ofstream sumOut(outName_, std::ios_base::app | std::ios_base::out);
    sumOut << num << ", " << inlier / total << ", " << dist / total << ", " << total << std::endl;
    return inlier / total;
  }

  std::vector<size_t> getPairs() {
    std::vector<size_t> out;
    if (it_ < testPairs_.size()) out = testPairs_[it_++];
    return out;
  }
  bool morePairs() const { return it_ < testPairs_.size(); }
  const std::vector<int>& getConfusionMatrix() const { return confusion_; }

 private:
  static std::string dir_of(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? "" : p.substr(0, k + 1); }
  static std::string base_of(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
  std::vector<std::vector<size_t>> testPairs_;
  size_t it_ = 0;
  std::vector<int> confusion_;
  size_t numClasses_;
  size_t count_;
  std::string outName_;
  semanticicp::detail::Engine engine_;
};

// One registration method of an experiment as an OPEN STREAM (sicp_stream_*, include/sicp.h): a scan is added once
// (cloud_of_scan[file index]) however many registrations it takes part in, a pair is submitted when both its scans are
// there, results come back in order of completion and are filed by pair.  What the three headless drivers' -S modes share.
struct MethodStream {
  sicp_stream s = nullptr;
  std::vector<int64_t> cloud_of_scan;   // cloud id per file index, 0 = not uploaded
  std::vector<int64_t> ticket_of_pair;
  std::vector<sicp_stream_result> result_of_pair;
  MethodStream() = default;
  MethodStream(const MethodStream&) = delete;
  MethodStream& operator=(const MethodStream&) = delete;
  ~MethodStream() { if (s) sicp_stream_destroy(s); }
  void check(int rc, const char* where) {
    if (rc != SICP_OK) throw std::runtime_error(std::string(where) + ": " + sicp_strerror(rc) + " " + (s ? sicp_stream_last_error(s) : ""));
  }
  // epsilon <= 0: the constructors' default (em_icp.h:43)
  void open(int mode, int classes, const double* cm, int in_flight, size_t n_files, size_t n_pairs, int device, double epsilon = 0.0) {
    sicp_params p;
    check(sicp_default_params(mode, &p), "sicp_default_params");
    p.num_classes = classes;
    if (epsilon > 0) p.epsilon = epsilon;
    // the drivers run two streams (two methods) on one device: neither is ever alone on it, so the persistent
    // one-workgroup-per-CU solve of a draining stream could not become resident beside the other's ticks -- ticks only
    p.lm_on_device = 2;
    check(sicp_stream_create(device, &p, in_flight, &s), "sicp_stream_create");
    if (cm) check(sicp_stream_set_confusion(s, classes, cm), "sicp_stream_set_confusion");
    cloud_of_scan.assign(n_files, 0);
    ticket_of_pair.assign(n_pairs, 0);
    result_of_pair.resize(n_pairs);
  }
  template <typename PointT>
  void add(size_t scan, const pcl::PointCloud<PointT>& c, bool with_labels) {
    check(semanticicp::detail::stream_add_cloud(s, c, with_labels, &cloud_of_scan[scan]), "sicp_stream_add_cloud_strided");
  }
  // source -> target from the identity (exec/kitti_eval.cc:172-176); flags: SICP_SUBMIT_*
  void submit(size_t pair, size_t target, size_t source, uint32_t flags = 0) {
    const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
    check(sicp_stream_submit_ex(s, cloud_of_scan[source], cloud_of_scan[target], ident, flags, &ticket_of_pair[pair]), "sicp_stream_submit_ex");
  }
  void collect(int wait) {
    sicp_stream_result r[64];
    for (;;) {
      int32_t n = 0;
      check(sicp_stream_poll(s, wait, 64, r, &n), "sicp_stream_poll");
      for (int32_t k = 0; k < n; ++k)
        for (size_t q = 0; q < ticket_of_pair.size(); ++q)
          if (ticket_of_pair[q] == r[k].ticket) { result_of_pair[q] = r[k]; break; }
      if (n < 64) break;
      wait = 0;
    }
  }
  // getFusedLabels of a pair submitted with SICP_SUBMIT_FUSED_LABELS (once)
  std::vector<uint32_t> labels(size_t pair, size_t n_source) {
    std::vector<uint32_t> out(n_source);
    check(sicp_stream_take_labels(s, ticket_of_pair[pair], (int32_t)n_source, out.data()), "sicp_stream_take_labels");
    return out;
  }
};

inline int device_from_env() {
  const char* dev = std::getenv("SICP_DEVICE");
  return dev ? std::atoi(dev) : 0;
}

inline const char* arg(int argc, char** argv, const char* flag) {
  for (int i = 1; i + 1 < argc; ++i)
    if (!std::strcmp(argv[i], flag)) return argv[i + 1];
  return nullptr;
}

}  // namespace evalsupport
#endif
