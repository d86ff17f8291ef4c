// kitti_eval_headless.cc -- the KITTI experiment loop of the reference (exec/kitti_eval.cc:124-249)
// on the MI355X engine: for every stride-3 pair (target n, source n+3) of the PCD files in -s, run
// EM-ICP<11> (:184-192) and SE3-GICP (:211-217) from the identity, score them against the ground
// truth poses in -t (exec/kitti_metrics.h) and write one CSV row per pair and method to
// <prefix>EMICPkitti.csv / <prefix>se3GICPkitti.csv (-o prefix; the reference uses a date string).
// Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison (third-party algorithm), the
// disabled FPFH bootstrap, and quirk Q7 (the reference hands SE3-GICP the kd-tree of another scan,
// exec/kitti_eval.cc:213; here every align() sees its own clouds).
// -b <pairs>: register that many pairs at a time (alignBatch, sicp_align_batch) instead of one after
// the other; the rows are the same, the run is several times faster.
// -r: every scan is read, uploaded and indexed ONCE: the source of pair (n, n+3) stays on the GPU and is
// the target of pair (n+3, n+6) (setTargetCloudSharedWithSourceOf), and its normals / histograms are
// kept (keepFeatures) instead of being recomputed by both registrations.  Same rows again.
// -S <in flight>: the whole sequence as an OPEN STREAM (sicp_stream_*, include/sicp.h): one stream per method,
// every scan read, uploaded and indexed once, its two registrations submitted as soon as it is there, up to
// <in flight> registrations sharing the GPU while this thread reads the next files -- no closed batches, no
// pair waiting for the slowest pair of its batch.  Same rows again (the time column is the run's wall time
// per pair).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <em_icp.h>
#include <gicp.h>

#include "eval_support.h"

namespace {

// one registration method of the experiment as a stream: scans are added once (ids[scan index]), pair
// (target n, source n + 3) is submitted when both are there, a scan is released after its second submit
struct MethodStream {
  sicp_stream s = nullptr;
  std::vector<int64_t> cloud_of_scan;   // cloud id per file index, 0 = not uploaded
  std::vector<int64_t> ticket_of_pair;
  std::vector<sicp_stream_result> result_of_pair;
  ~MethodStream() { if (s) sicp_stream_destroy(s); }
  void check(int rc, const char* where) {
    if (rc != SICP_OK) throw std::runtime_error(std::string(where) + ": " + sicp_strerror(rc) + " " + (s ? sicp_stream_last_error(s) : ""));
  }
  void open(int mode, int classes, const double* cm, int in_flight, size_t n_files, size_t n_pairs) {
    sicp_params p;
    check(sicp_default_params(mode, &p), "sicp_default_params");
    p.num_classes = classes;
    // two streams (one per method) share the device: neither is ever alone on it, so the persistent one-workgroup-per-CU
    // solve of a draining stream could not become resident beside the other's ticks -- ticks only
    p.lm_on_device = 2;
    const char* dev = std::getenv("SICP_DEVICE");
    check(sicp_stream_create(dev ? std::atoi(dev) : 0, &p, in_flight, &s), "sicp_stream_create");
    if (cm) check(sicp_stream_set_confusion(s, classes, cm), "sicp_stream_set_confusion");
    cloud_of_scan.assign(n_files, 0);
    ticket_of_pair.assign(n_pairs, 0);
    result_of_pair.resize(n_pairs);
  }
  template <typename PointT>
  void add(size_t scan, const pcl::PointCloud<PointT>& c, bool with_labels) {
    check(semanticicp::detail::stream_add_cloud(s, c, with_labels, &cloud_of_scan[scan]), "sicp_stream_add_cloud_strided");
  }
  void submit(size_t pair, size_t target, size_t source) {
    const double ident[7] = {0, 0, 0, 1, 0, 0, 0};   // exec/kitti_eval.cc:172-176
    check(sicp_stream_submit(s, cloud_of_scan[source], cloud_of_scan[target], ident, &ticket_of_pair[pair]), "sicp_stream_submit");
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
};

}  // namespace

int main(int argc, char** argv) {
  using namespace evalsupport;
  const char *dir = arg(argc, argv, "-s"), *gt = arg(argc, argv, "-t"), *cmf = arg(argc, argv, "-m"), *prefix = arg(argc, argv, "-o");
  const char* barg = arg(argc, argv, "-b");
  const size_t batch = barg ? (size_t)std::max(1, std::atoi(barg)) : 1;
  bool share = false;
  for (int i = 1; i < argc; ++i) share = share || std::string(argv[i]) == "-r";
  if (!dir) { std::cout << "Need source directory (-s)\n"; return -1; }
  if (!gt) { std::cout << "Need ground truth file (-t)\n"; return -1; }
  if (!cmf) { std::cout << "Need ground confusion matrix file (-m)\n"; return -1; }
  const std::string pre = prefix ? prefix : "";
  const Eigen::Matrix<double, 11, 11> cm = ReadConfusionMatrix<11>(cmf);
  const std::vector<std::string> pcd_fns = get_pcd_in_dir(dir);
  std::ofstream foutSICP(pre + "EMICPkitti.csv"), foutse3GICP(pre + "se3GICPkitti.csv");
  const std::string gtFile = gt;
  KittiMetrics semanticICPMetrics(gtFile, &foutSICP), se3GICPMetrics(gtFile, &foutse3GICP);
  typedef semanticicp::EmIterativeClosestPoint<11> Em;
  typedef semanticicp::GICP<pcl::PointXYZ> Gicp;
  const char* sarg = arg(argc, argv, "-S");
  if (sarg) {
    try {
      const int in_flight = std::max(1, std::atoi(sarg));
      std::vector<size_t> starts;
      for (size_t n = 0; n + 3 < pcd_fns.size(); n += 3) starts.push_back(n);  // exec/kitti_eval.cc:124-129
      double cmv[121];
      for (int r = 0; r < 11; ++r)
        for (int c = 0; c < 11; ++c) cmv[11 * r + c] = cm(r, c);
      MethodStream em, gi;
      em.open(SICP_MODE_EM, 11, cmv, in_flight, pcd_fns.size(), starts.size());
      gi.open(SICP_MODE_GICP, 0, nullptr, in_flight, pcd_fns.size(), starts.size());
      auto upload = [&](size_t scan) {
        if (em.cloud_of_scan[scan]) return true;
        pcl::PointCloud<pcl::PointXYZL>::Ptr cl(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[scan], *cl) == -1) return false;
        filterRange(cl, 40.0);  // :138, :159
        em.add(scan, *cl, true);
        // the reference re-loads the files as PointXYZ (:201-204): same points, no labels, no range filter
        pcl::PointCloud<pcl::PointXYZ>::Ptr raw(new pcl::PointCloud<pcl::PointXYZ>);
        pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[scan], *raw);
        gi.add(scan, *raw, false);
        return true;
      };
      const auto begin = std::chrono::steady_clock::now();
      for (size_t q = 0; q < starts.size(); ++q) {
        const size_t t = starts[q], sidx = t + 3;
        if (!upload(t) || !upload(sidx)) { std::cerr << "Couldn't read scan file\n"; return -1; }
        em.submit(q, t, sidx);
        gi.submit(q, t, sidx);
        // scan t has now been the source of pair q - 1 and the target of pair q: the caller is done with it
        sicp_stream_release_cloud(em.s, em.cloud_of_scan[t]);
        sicp_stream_release_cloud(gi.s, gi.cloud_of_scan[t]);
        if (q % 16 == 15) { em.collect(0); gi.collect(0); }
      }
      em.collect(2);
      gi.collect(2);
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(std::max<size_t>(1, starts.size()));
      for (size_t q = 0; q < starts.size(); ++q) {
        const size_t t = starts[q], sidx = t + 3;
        if (em.result_of_pair[q].status != SICP_OK || gi.result_of_pair[q].status != SICP_OK)
          throw std::runtime_error(std::string("registration failed: ") + sicp_strerror(em.result_of_pair[q].status ? em.result_of_pair[q].status : gi.result_of_pair[q].status));
        const double e1 = semanticICPMetrics.evaluate(semanticicp::detail::to_se3(em.result_of_pair[q].qt), t, sidx, secs, em.result_of_pair[q].outer_iters);
        const double e2 = se3GICPMetrics.evaluate(semanticicp::detail::to_se3(gi.result_of_pair[q].qt), t, sidx, secs, gi.result_of_pair[q].outer_iters);
        std::printf("pair %zu<-%zu  SICP MSE %.3e  se3GICP MSE %.3e\n", t, sidx, e1, e2);
      }
    } catch (const std::exception& e) {
      std::cerr << "error: " << e.what() << "\n";
      return 2;
    }
    std::printf("SICP FINAL MSE %.6e rot %.6e trans %.6e\n", semanticICPMetrics.getTransformMSE(), semanticICPMetrics.getRotMSE(), semanticICPMetrics.getTransMSE());
    std::printf("se3GICP FINAL MSE %.6e rot %.6e trans %.6e\n", se3GICPMetrics.getTransformMSE(), se3GICPMetrics.getRotMSE(), se3GICPMetrics.getTransMSE());
    return 0;
  }
  try {
    // one engine per method and batch slot for the whole run: device buffers and the captured solver
    // graphs are reused
    std::vector<std::unique_ptr<Em>> em(batch);
    std::vector<std::unique_ptr<Gicp>> gi(batch);
    for (size_t b = 0; b < batch; ++b) {
      em[b].reset(new Em());
      gi[b].reset(new Gicp());
      em[b]->setConfusionMatrix(cm);
      em[b]->keepFeatures(share);
      gi[b]->keepFeatures(share);
    }
    size_t prev_slot = 0;  // batch slot of the previous pair (its source scan is this pair's target scan)
    std::vector<size_t> starts;
    for (size_t n = 0; n + 3 < pcd_fns.size(); n += 3) starts.push_back(n);  // exec/kitti_eval.cc:124-129
    for (size_t g0 = 0; g0 < starts.size(); g0 += batch) {
      const size_t cnt = std::min(batch, starts.size() - g0);
      std::vector<Em*> eo(cnt);
      std::vector<Gicp*> go(cnt);
      std::vector<pcl::PointCloud<pcl::PointXYZL>::Ptr> finalEm(cnt);
      std::vector<pcl::PointCloud<pcl::PointXYZ>::Ptr> finalGi(cnt);
      std::vector<Sophus::SE3d> inits(cnt);  // identity (:172-176)
      for (size_t b = 0; b < cnt; ++b) {
        const size_t indxTarget = starts[g0 + b], indxSource = indxTarget + 3;
        const bool shared = share && g0 + b > 0;  // the target scan is the previous pair's source scan
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxSource], *cloudA) == -1) { std::cerr << "Couldn't read source file\n"; return -1; }
        filterRange(cloudA, 40.0);  // :138
        eo[b] = em[b].get();
        if (shared) {
          eo[b]->setTargetCloudSharedWithSourceOf(*em[prev_slot]);  // before this slot's own source is replaced
        } else {
          if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxTarget], *cloudB) == -1) { std::cerr << "Couldn't read target file\n"; return -1; }
          filterRange(cloudB, 40.0);  // :159
          eo[b]->setTargetCloud(cloudB);
        }
        eo[b]->setSourceCloud(cloudA);
        finalEm[b].reset(new pcl::PointCloud<pcl::PointXYZL>);
        // the reference re-loads the files as PointXYZ (:201-204): same points, no labels, no range filter
        pcl::PointCloud<pcl::PointXYZ>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZ>), cloudBnoL(new pcl::PointCloud<pcl::PointXYZ>);
        pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxSource], *cloudAnoL);
        go[b] = gi[b].get();
        if (shared) {
          go[b]->setTargetCloudSharedWithSourceOf(*gi[prev_slot]);  // before this slot's own source is replaced (prev_slot may be b)
        } else {
          pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxTarget], *cloudBnoL);
          go[b]->setTargetCloud(cloudBnoL);
        }
        go[b]->setSourceCloud(cloudAnoL);
        finalGi[b].reset(new pcl::PointCloud<pcl::PointXYZ>);
        prev_slot = b;
      }
      auto begin = std::chrono::steady_clock::now();
      if (batch == 1) eo[0]->align(finalEm[0], inits[0]);
      else Em::alignBatch(eo, finalEm, inits);
      const double secsEm = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(cnt);
      begin = std::chrono::steady_clock::now();
      if (batch == 1) go[0]->align(finalGi[0]);
      else Gicp::alignBatch(go, finalGi, inits);
      const double secsGi = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(cnt);
      for (size_t b = 0; b < cnt; ++b) {
        const size_t indxTarget = starts[g0 + b], indxSource = indxTarget + 3;
        const double e1 = semanticICPMetrics.evaluate(eo[b]->getFinalTransFormation(), indxTarget, indxSource, secsEm, eo[b]->getOuterIter());
        const double e2 = se3GICPMetrics.evaluate(go[b]->getFinalTransFormation(), indxTarget, indxSource, secsGi, go[b]->getOuterIter());
        std::printf("pair %zu<-%zu  SICP MSE %.3e  se3GICP MSE %.3e\n", indxTarget, indxSource, e1, e2);
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 2;
  }
  std::printf("SICP FINAL MSE %.6e rot %.6e trans %.6e\n", semanticICPMetrics.getTransformMSE(), semanticICPMetrics.getRotMSE(), semanticICPMetrics.getTransMSE());
  std::printf("se3GICP FINAL MSE %.6e rot %.6e trans %.6e\n", se3GICPMetrics.getTransformMSE(), se3GICPMetrics.getRotMSE(), se3GICPMetrics.getTransMSE());
  return 0;
}
