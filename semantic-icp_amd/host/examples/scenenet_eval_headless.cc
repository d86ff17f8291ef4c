// scenenet_eval_headless.cc -- the SceneNet experiment loop of the reference
// (exec/scenenet_eval.cc:110-250) on the MI355X engine: consecutive frames (target n, source n+1) of
// the PCD files in -s; EM-ICP<13>(20, 1e-6) from the identity (:174-186), its fused labels written
// as <out-prefix><source index>.pcd (:193-198), SE3-GICP(20, 1e-6) (:206-214); both scored against
// the ground truth in -t with the SceneNet row format (exec/scenenet_metrics.h:17-29) and written as
// CSV rows to <prefix>EMICPscenenet.csv / <prefix>se3GICPscenenet.csv.
// Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison and the disabled bootstrap.
// -b <pairs>: register that many frame pairs at a time in lock step (alignBatch); same rows and files.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <em_icp.h>
#include <gicp.h>

#include "eval_support.h"

int main(int argc, char** argv) {
  using namespace evalsupport;
  const char *dir = arg(argc, argv, "-s"), *gt = arg(argc, argv, "-t"), *cmf = arg(argc, argv, "-m"), *prefix = arg(argc, argv, "-o");
  if (!dir) { std::cout << "Need source directory (-s)\n"; return -1; }
  if (!gt) { std::cout << "Need ground truth file (-t)\n"; return -1; }
  if (!cmf) { std::cout << "Need ground confusion matrix file (-m)\n"; return -1; }
  const std::string pre = prefix ? prefix : "";
  const Eigen::Matrix<double, 13, 13> cm = ReadConfusionMatrix<13>(cmf);
  const std::vector<std::string> pcd_fns = get_pcd_in_dir(dir);
  std::ofstream foutSICP(pre + "EMICPscenenet.csv"), foutse3GICP(pre + "se3GICPscenenet.csv");
  KittiMetrics semanticICPMetrics(gt, &foutSICP, KittiMetrics::SCENENET_4x4_INVERTED);
  KittiMetrics se3GICPMetrics(gt, &foutse3GICP, KittiMetrics::SCENENET_4x4_INVERTED);
  const int STEP = 1;  // exec/scenenet_eval.cc:112
  const char* barg = arg(argc, argv, "-b");
  const size_t batch = barg ? (size_t)std::max(1, std::atoi(barg)) : 1;
  typedef semanticicp::EmIterativeClosestPoint<13> Em;
  typedef semanticicp::GICP<pcl::PointXYZ> Gicp;
  try {
    std::vector<std::unique_ptr<Em>> em(batch);
    std::vector<std::unique_ptr<Gicp>> gi(batch);
    for (size_t b = 0; b < batch; ++b) {
      em[b].reset(new Em(20, 1e-6));
      gi[b].reset(new Gicp(20, 1e-6));
      em[b]->setConfusionMatrix(cm);
    }
    std::vector<size_t> starts;
    for (size_t n = 0; n + STEP < pcd_fns.size(); n += STEP) starts.push_back(n);
    for (size_t g0 = 0; g0 < starts.size(); g0 += batch) {
      const size_t cnt = std::min(batch, starts.size() - g0);
      std::vector<Em*> eo(cnt);
      std::vector<Gicp*> go(cnt);
      std::vector<pcl::PointCloud<pcl::PointXYZL>::Ptr> finalEm(cnt);
      std::vector<pcl::PointCloud<pcl::PointXYZ>::Ptr> finalGi(cnt);
      std::vector<Sophus::SE3d> inits(cnt);
      for (size_t b = 0; b < cnt; ++b) {
        const size_t indxTarget = starts[g0 + b], indxSource = indxTarget + STEP;
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxSource], *cloudA) == -1) { std::cerr << "Couldn't read source file\n"; return -1; }
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxTarget], *cloudB) == -1) { std::cerr << "Couldn't read target file\n"; return -1; }
        eo[b] = em[b].get();
        eo[b]->setSourceCloud(cloudA);
        eo[b]->setTargetCloud(cloudB);
        finalEm[b].reset(new pcl::PointCloud<pcl::PointXYZL>);
        pcl::PointCloud<pcl::PointXYZ>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZ>), cloudBnoL(new pcl::PointCloud<pcl::PointXYZ>);
        pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxSource], *cloudAnoL);
        pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxTarget], *cloudBnoL);
        go[b] = gi[b].get();
        go[b]->setSourceCloud(cloudAnoL);
        go[b]->setTargetCloud(cloudBnoL);
        finalGi[b].reset(new pcl::PointCloud<pcl::PointXYZ>);
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
        const size_t indxTarget = starts[g0 + b], indxSource = indxTarget + STEP;
        const Sophus::SE3d sicpTranform = eo[b]->getFinalTransFormation();
        const double e1 = semanticICPMetrics.evaluate(sicpTranform, indxTarget, indxSource, secsEm, eo[b]->getOuterIter());
        pcl::PointCloud<pcl::PointXYZL>::Ptr labeledCloudem(new pcl::PointCloud<pcl::PointXYZL>);
        eo[b]->getFusedLabels(labeledCloudem, sicpTranform);  // :193-195
        std::ostringstream name;
        name << pre << indxSource << ".pcd";
        pcl::io::savePCDFileASCII(name.str(), *labeledCloudem);  // :196-198
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
