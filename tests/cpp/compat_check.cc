// compat_check.cc -- exercises the look-alike Eigen / Sophus / PCL slices (semantic-icp_amd/host/compat) the way
// the reference's drivers use them and prints the results as JSON lines for tests/test_compat_headers.py to
// compare with numpy / scipy.  No GPU, no libsicp: header-only code.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>

#include "compat/eigen_lite.h"
#include "compat/pcl_lite.h"
#include "compat/sophus_lite.h"

template <typename M>
static void dump(const char* name, const M& m) {
  std::printf("{\"name\": \"%s\", \"rows\": %d, \"cols\": %d, \"v\": [", name, (int)m.rows(), (int)m.cols());
  for (int r = 0; r < m.rows(); ++r)
    for (int c = 0; c < m.cols(); ++c) std::printf("%s%.17g", (r || c) ? ", " : "", (double)m(r, c));
  std::printf("]}\n");
}
static void dump_raw(const char* name, const double* p, int n) {
  std::printf("{\"name\": \"%s\", \"v\": [", name);
  for (int i = 0; i < n; ++i) std::printf("%s%.17g", i ? ", " : "", p[i]);
  std::printf("]}\n");
}
static void dump_text(const char* name, const std::string& s) {
  std::printf("{\"name\": \"%s\", \"text\": \"", name);
  for (char ch : s) { if (ch == '\n') std::printf("\\n"); else std::putchar(ch); }
  std::printf("\"}\n");
}

int main(int argc, char** argv) {
  // --- exec/kitti_metrics.h:17-24: 12 numbers, row major, into the top 3x4 block of an identity ---
  double data[12];
  for (int i = 0; i < 12; ++i) data[i] = 0.5 * i - 2.25;
  Eigen::Matrix4d mat = Eigen::Matrix4d::Identity();
  mat.block<3, 4>(0, 0) = Eigen::Map<Eigen::Matrix<double, 3, 4, Eigen::RowMajor>>(data);
  dump("block_from_rowmajor_map", mat);
  // --- exec/scenenet_metrics.h:19-25: 16 numbers, row major ---
  double d16[16];
  for (int i = 0; i < 16; ++i) d16[i] = i * i - 3.0;
  Eigen::Matrix4d m16;
  m16 = Eigen::Map<Eigen::Matrix<double, 4, 4, Eigen::RowMajor>>(d16);
  dump("assign_from_rowmajor_map", m16);
  // --- exec/kitti_metrics.h:52-57: a RowMajor copy, walked through data() + size() ---
  Eigen::Matrix<double, 4, 4, Eigen::RowMajor> temp = m16;
  dump_raw("rowmajor_data", temp.data(), (int)temp.size());
  dump_raw("colmajor_data", m16.data(), (int)m16.size());
  // --- products, transposes, casts ---
  Eigen::Matrix4f mf = m16.cast<float>();
  dump("cast_float", mf);
  dump("product", m16 * mat.transpose());
  Eigen::Vector3d a(1, 2, 3), b(-2, 0.5, 4);
  dump("cross", a.cross(b));
  std::printf("{\"name\": \"dot\", \"v\": [%.17g, %.17g, %.17g]}\n", a.dot(b), a.squaredNorm(), (a - b).norm());
  // --- comma initialiser (exec/test_gradient.cc:37-39 style) ---
  Eigen::Matrix3d cov;
  cov << 0.674143, 0.460412, 0.085842, 0.460412, 0.349471, -0.121288, 0.085842, -0.121288, 0.977386;
  dump("comma", cov);
  // --- exec/nyu_metrics.h:32,62,75: a dynamic integer matrix, counted into and printed ---
  Eigen::MatrixXi conf = Eigen::MatrixXi::Zero(4, 4);
  conf(1, 2)++; conf(1, 2)++; conf(3, 0) += 120; conf(0, 0) = 7;
  {
    std::ostringstream os;
    os << conf;
    dump_text("matrixxi_print", os.str());
  }
  {
    std::ostringstream os;
    os << m16 * 0.37;
    dump_text("matrix4d_print", os.str());
  }
  // --- Sophus: fitToSE3 of perturbed rotations (incl. one that needs the det(U)det(V) correction), so3().log() ---
  std::mt19937_64 rng(7);
  std::normal_distribution<double> N(0, 1);
  for (int t = 0; t < 6; ++t) {
    Eigen::Matrix<double, 6, 1> tw;
    for (int i = 0; i < 6; ++i) tw(i) = N(rng) * (i < 3 ? 2.0 : 0.8);
    Sophus::SE3d T = Sophus::SE3d::exp(tw);
    Eigen::Matrix4d M = T.matrix();
    const double noise = t < 4 ? 1e-3 : 0.3;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) += noise * N(rng);
    if (t == 5) for (int j = 0; j < 3; ++j) M(2, j) = -M(2, j);  // determinant < 0
    char name[64];
    std::snprintf(name, sizeof name, "fit_in_%d", t); dump(name, M);
    Sophus::SE3d F = Sophus::SE3d::fitToSE3(M);
    std::snprintf(name, sizeof name, "fit_out_%d", t); dump(name, F.matrix());
    std::snprintf(name, sizeof name, "fit_so3log_%d", t); dump(name, F.so3().log());
    std::snprintf(name, sizeof name, "fit_log_%d", t); dump(name, F.log());
    std::snprintf(name, sizeof name, "fit_trans_%d", t); dump(name, F.translation());
  }
  // --- pcl::KdTreeFLANN::nearestKSearch against brute force, ties and non-finite points included ---
  {
    pcl::PointCloud<pcl::PointXYZL>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZL>);
    std::uniform_real_distribution<float> U(-5.f, 5.f);
    const int n = 5000;
    for (int i = 0; i < n; ++i) {
      pcl::PointXYZL p;
      p.x = U(rng); p.y = U(rng); p.z = U(rng); p.label = (uint32_t)(i % 13);
      if (i % 7 == 0) { p.x = std::round(p.x); p.y = std::round(p.y); p.z = std::round(p.z); }  // a lattice: exact ties
      if (i == 100) p.x = NAN;
      if (i == 200) p.z = INFINITY;
      cloud->push_back(p);
    }
    pcl::KdTreeFLANN<pcl::PointXYZL>::Ptr tree(new pcl::KdTreeFLANN<pcl::PointXYZL>());
    tree->setInputCloud(cloud);
    long mismatches = 0, checked = 0;
    for (int qi = 0; qi < 800; ++qi) {
      pcl::PointXYZL q;
      q.x = U(rng); q.y = U(rng); q.z = U(rng);
      if (qi % 5 == 0) { q.x = std::round(q.x) + 0.5f; q.y = std::round(q.y) + 0.5f; q.z = std::round(q.z) + 0.5f; }  // equidistant corners
      for (int k : {1, 4, 20}) {
        std::vector<int> idx;
        std::vector<float> d2;
        const int got = tree->nearestKSearch(q, k, idx, d2);
        std::vector<std::pair<uint64_t, float>> all;
        for (int i = 0; i < n; ++i) {
          const auto& p = cloud->points[i];
          if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
          const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
          float d = dx * dx; d = d + dy * dy; d = d + dz * dz;
          uint32_t bits; std::memcpy(&bits, &d, 4);
          all.push_back({((uint64_t)bits << 32) | (uint32_t)i, d});
        }
        std::sort(all.begin(), all.end());
        if (got != k) ++mismatches;
        for (int j = 0; j < k && j < got; ++j) {
          ++checked;
          if (idx[j] != (int)(uint32_t)all[j].first || d2[j] != all[j].second) ++mismatches;
        }
      }
    }
    // k larger than the cloud is clamped; a non-finite query finds nothing
    pcl::PointCloud<pcl::PointXYZL>::Ptr tiny(new pcl::PointCloud<pcl::PointXYZL>);
    for (int i = 0; i < 3; ++i) { pcl::PointXYZL p; p.x = (float)i; tiny->push_back(p); }
    pcl::KdTreeFLANN<pcl::PointXYZL> t2;
    t2.setInputCloud(tiny);
    std::vector<int> idx; std::vector<float> d2;
    pcl::PointXYZL q; q.x = 1.9f;
    const int got = t2.nearestKSearch(q, 10, idx, d2);
    q.x = NAN;
    const int got_nan = t2.nearestKSearch(q, 1, idx, d2);
    std::printf("{\"name\": \"kdtree\", \"checked\": %ld, \"mismatches\": %ld, \"clamped\": %d, \"nan_query\": %d}\n", checked, mismatches, got, got_nan);
  }
  // --- PCD files named on the command line: counts + checksums as the loader sees them ---
  for (int i = 1; i < argc; ++i) {
    pcl::PointCloud<pcl::PointXYZL> c;
    const int rc = pcl::io::loadPCDFile<pcl::PointXYZL>(argv[i], c);
    pcl::PointCloud<pcl::PointXYZ> c3;
    const int rc3 = pcl::io::loadPCDFile<pcl::PointXYZ>(argv[i], c3);
    double sx = 0, sy = 0, sz = 0; unsigned long long sl = 0;
    for (const auto& p : c.points) {  // non-finite coordinates are left out of the checksum (JSON has no NaN)
      if (std::isfinite(p.x)) sx += p.x;
      if (std::isfinite(p.y)) sy += p.y;
      if (std::isfinite(p.z)) sz += p.z;
      sl += p.label;
    }
    std::printf("{\"name\": \"pcd\", \"file\": \"%s\", \"rc\": %d, \"rc_xyz\": %d, \"n\": %zu, \"n_xyz\": %zu, \"width\": %u, \"height\": %u, \"dense\": %d, "
                "\"sum\": [%.17g, %.17g, %.17g], \"label_sum\": %llu}\n",
                argv[i], rc, rc3, c.size(), c3.size(), c.width, c.height, (int)c.is_dense, sx, sy, sz, sl);
  }
  return 0;
}
