// semantic_cloud_check.cc -- the device-resident SemanticPointCloud of the class shims (semantic-icp_amd/host) against the C ABI
// it is built on.  Run by tests/test_host_shims.py on a GPU.  usage: semantic_cloud_check a.pcd b.pcd
//   1. labeledCovariances (fetched on first read) == sicp_covariances of every label cloud on its own, bit for bit
//      (impl/semantic_point_cloud.hpp:25-84: per label cloud, at addSemanticCloud)
//   2. read AFTER transform() they are still the covariances of the cloud as it was added (the reference never updates them)
//   3. SemanticIterativeClosestPoint::align on the shared device clouds == sicp_align on the flattened clouds, bit for bit,
//      twice in a row with the same target object (its device cloud and covariances are reused)
//   5. a caller's own vector in labeledCovariances is handed to the engine (the engine's form: product kernels; another symmetric form:
//      the full-matrix path; no covariance at all: refused loudly)
//   4. sicp_destroy parks a handle, sicp_create hands it out again, sicp_release_pool lets go of it
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <cstring>
#include <memory>
#include <vector>

#include <pcl_2_semantic.h>
#include <semantic_icp.h>
#include <semantic_point_cloud.h>

typedef semanticicp::SemanticPointCloud<pcl::PointXYZ, uint32_t> SemCloud;
using semanticicp::detail::check;

static std::vector<double> label_cov(sicp_handle h, const pcl::PointCloud<pcl::PointXYZ>& c, int k, double eps) {
  sicp_params p;
  check(sicp_default_params(SICP_MODE_GICP, &p), h, "params");
  p.k_cov = k; p.epsilon = eps;
  check(sicp_set_params(h, &p), h, "set_params");
  check(semanticicp::detail::set_cloud(h, SICP_SOURCE, c, false), h, "set_cloud");
  std::vector<double> c9(c.size() * 9);
  check(sicp_covariances(h, SICP_SOURCE, c9.data(), nullptr, nullptr, nullptr), h, "covariances");
  return c9;
}

static bool same(const SemCloud::MatricesVector& v, const std::vector<double>& c9) {
  if (v.size() * 9 != c9.size()) return false;
  for (size_t i = 0; i < v.size(); ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        const double x = v[i](a, b);
        if (std::memcmp(&x, &c9[i * 9 + 3 * a + b], sizeof(double)) != 0) return false;
      }
  return true;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  pcl::PointCloud<pcl::PointXYZL>::Ptr A(new pcl::PointCloud<pcl::PointXYZL>), B(new pcl::PointCloud<pcl::PointXYZL>);
  if (pcl::io::loadPCDFile<pcl::PointXYZL>(argv[1], *A) == -1 || pcl::io::loadPCDFile<pcl::PointXYZL>(argv[2], *B) == -1) return 3;
  try {
    sicp_handle raw = nullptr;
    check(sicp_create(0, &raw), nullptr, "create");
    // ---- 1
    std::shared_ptr<SemCloud> sa(new SemCloud()), sb(new SemCloud()), sa2(new SemCloud());
    semanticicp::pcl_2_semantic(A, sa);
    semanticicp::pcl_2_semantic(B, sb);
    semanticicp::pcl_2_semantic(A, sa2);
    std::vector<std::vector<double>> ref_a;
    bool ok1 = sa->labeledCovariances.size() == sa->semanticLabels.size();
    for (uint32_t l : sa->semanticLabels) {
      ref_a.push_back(label_cov(raw, *sa->labeledPointClouds[l], sa->getK(), sa->getEpsilon()));
      ok1 = ok1 && sa->labeledCovariances.count(l) == 1 && same(*sa->labeledCovariances[l], ref_a.back());
    }
    std::printf("lazy_covariances_equal_per_label_clouds %d labels %zu\n", (int)ok1, sa->semanticLabels.size());
    // ---- 2
    Eigen::Matrix4f M = Eigen::Matrix4f::Identity();
    M(0, 3) = 1.5f; M(0, 0) = 0.f; M(0, 1) = -1.f; M(1, 0) = 1.f; M(1, 1) = 0.f;  // 90 degrees about z + a shift
    sa2->transform(M);
    bool ok2 = true;
    size_t li = 0;
    for (uint32_t l : sa2->semanticLabels) ok2 = ok2 && same(*sa2->labeledCovariances.at(l), ref_a[li++]);
    std::printf("covariances_after_transform_are_those_of_the_added_cloud %d\n", (int)ok2);
    // ---- 3
    sicp_params p;
    check(sicp_default_params(SICP_MODE_SEMANTIC, &p), raw, "params");
    check(sicp_set_params(raw, &p), raw, "set_params");
    check(semanticicp::detail::set_cloud(raw, SICP_SOURCE, *sa->getpclPointCloud(), true), raw, "set_cloud");
    check(semanticicp::detail::set_cloud(raw, SICP_TARGET, *sb->getpclPointCloud(), true), raw, "set_cloud");
    const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
    double want[7];
    sicp_stats st0;
    check(sicp_align(raw, ident, want, nullptr, &st0), raw, "align");
    bool ok3 = true;
    for (int rep = 0; rep < 2; ++rep) {
      std::shared_ptr<SemCloud> src(new SemCloud()), fin(new SemCloud());
      semanticicp::pcl_2_semantic(A, src);
      semanticicp::pcl_2_semantic(A, fin);
      semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> icp;
      icp.setInputSource(src);
      icp.setInputTarget(sb);   // the same target object both times: its device cloud and covariances are reused
      icp.align(fin);
      ok3 = ok3 && std::memcmp(icp.getFinalTransFormation().data(), want, sizeof want) == 0;
    }
    std::printf("align_on_shared_device_clouds_equals_flat_c_abi %d\n", (int)ok3);
    // ---- 5: caller-supplied covariances (impl/semantic_icp.hpp:73,77 reads whatever sits in labeledCovariances)
    {
      std::shared_ptr<SemCloud> src(new SemCloud()), fin(new SemCloud()), src2(new SemCloud()), fin2(new SemCloud());
      semanticicp::pcl_2_semantic(A, src); semanticicp::pcl_2_semantic(A, fin);
      semanticicp::pcl_2_semantic(A, src2); semanticicp::pcl_2_semantic(A, fin2);
      const uint32_t l0 = src->semanticLabels[0];
      // (a) the caller's own vector holding the very matrices the engine computes: taken, same registration (the normal is
      //     rebuilt from the matrix: equal to rounding, so the pose is equal to ~1e-12, not bit for bit)
      SemCloud::MatricesVectorPtr copy(new SemCloud::MatricesVector(*src->labeledCovariances[l0]));
      src->labeledCovariances[l0] = copy;
      semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> icp;
      icp.setInputSource(src); icp.setInputTarget(sb);
      icp.align(fin);
      double dmax = 0;
      for (int i = 0; i < 7; ++i) dmax = std::max(dmax, std::fabs(icp.getFinalTransFormation().data()[i] - want[i]));
      std::printf("supplied_covariances_of_the_engines_form_are_taken %d\n", (int)(dmax < 1e-9));
      // (b) a matrix that is no covariance (not symmetric): refused with an exception, never ignored
      SemCloud::MatricesVectorPtr bad(new SemCloud::MatricesVector(*copy));
      (*bad)[3](0, 1) += 0.25;
      src2->labeledCovariances[l0] = bad;
      semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> icp2;
      icp2.setInputSource(src2); icp2.setInputTarget(sb);
      bool thrown = false;
      try { icp2.align(fin2); } catch (const std::runtime_error& e) { thrown = std::strstr(e.what(), "sicp_set_covariances") != nullptr; }
      std::printf("supplied_covariances_that_are_no_covariances_are_refused_loudly %d\n", (int)thrown);
      // (c) symmetric matrices of ANOTHER form (here diag(0.5, 1, 2) everywhere in one label): taken, evaluated on the full
      //     matrices -- a different registration problem, so a different answer, and still a registration
      std::shared_ptr<SemCloud> src3(new SemCloud()), fin3(new SemCloud());
      semanticicp::pcl_2_semantic(A, src3); semanticicp::pcl_2_semantic(A, fin3);
      SemCloud::MatricesVectorPtr gen(new SemCloud::MatricesVector(*src3->labeledCovariances[l0]));
      for (auto& M3 : *gen) { M3 = Eigen::Matrix3d::Identity(); M3(0, 0) = 0.5; M3(2, 2) = 2.0; }
      src3->labeledCovariances[l0] = gen;
      semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> icp3;
      icp3.setInputSource(src3); icp3.setInputTarget(sb);
      icp3.align(fin3);
      double d3 = 0;
      bool fin_ok = true;
      for (int i = 0; i < 7; ++i) { d3 = std::max(d3, std::fabs(icp3.getFinalTransFormation().data()[i] - want[i])); fin_ok = fin_ok && std::isfinite(icp3.getFinalTransFormation().data()[i]); }
      std::printf("supplied_covariances_of_general_form_are_taken %d max_pose_component_difference %.3e\n", (int)(fin_ok && d3 > 1e-9 && d3 < 0.2), d3);
    }
    // ---- 4
    sicp_handle h1 = nullptr, h2 = nullptr;
    check(sicp_create(0, &h1), nullptr, "create");
    check(sicp_destroy(h1), nullptr, "destroy");
    check(sicp_create(0, &h2), nullptr, "create");
    sicp_params q;
    check(sicp_get_params(h2, &q), h2, "get_params");
    int32_t n = -1;
    const bool fresh = q.mode == SICP_MODE_GICP && sicp_cloud_size(h2, SICP_SOURCE, &n, nullptr) == SICP_ERR_NOT_READY;
    std::printf("destroyed_handle_is_reused_and_fresh %d\n", (int)(h1 == h2 && fresh));
    check(sicp_destroy(h2), nullptr, "destroy");
    check(sicp_destroy(raw), nullptr, "destroy");
    sa.reset(); sb.reset(); sa2.reset();
    check(sicp_release_pool(0), nullptr, "release_pool");
    long long held = -1;
    check(sicp_memory_reserved(0, (int64_t*)&held), nullptr, "memory_reserved");
    std::printf("release_pool_frees_parked_handles %d held %lld\n", (int)(held == 0), held);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "semantic_cloud_check: %s\n", e.what());
    return 1;
  }
  return 0;
}
