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
#include <chrono>
#include <cstdio>
#include <fstream>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <em_icp.h>
#include <gicp.h>

#include "eval_support.h"

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
