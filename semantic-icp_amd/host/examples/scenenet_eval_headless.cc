// scenenet_eval_headless.cc -- the SceneNet experiment loop of the reference
// (exec/scenenet_eval.cc:110-250) on the MI355X engine: consecutive frames (target n, source n+1) of
// the PCD files in -s; EM-ICP<13>(20, 1e-6) from the identity (:174-186), its fused labels written
// as <out-prefix><source index>.pcd (:193-198), SE3-GICP(20, 1e-6) (:206-214); both scored against
// the ground truth in -t with the SceneNet row format (exec/scenenet_metrics.h:17-29) and written as
// CSV rows to <prefix>EMICPscenenet.csv / <prefix>se3GICPscenenet.csv.
// Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison and the disabled bootstrap.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

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
  try {
    semanticicp::EmIterativeClosestPoint<13> emicp(20, 1e-6);
    semanticicp::GICP<pcl::PointXYZ> gicpse3(20, 1e-6);
    emicp.setConfusionMatrix(cm);
    for (size_t n = 0; n + STEP < pcd_fns.size(); n += STEP) {
      const size_t indxTarget = n, indxSource = n + STEP;
      pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
      if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxSource], *cloudA) == -1) { std::cerr << "Couldn't read source file\n"; return -1; }
      if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxTarget], *cloudB) == -1) { std::cerr << "Couldn't read target file\n"; return -1; }
      Sophus::SE3d initTransform;
      pcl::PointCloud<pcl::PointXYZL>::Ptr finalCloudem(new pcl::PointCloud<pcl::PointXYZL>);
      auto begin = std::chrono::steady_clock::now();
      emicp.setSourceCloud(cloudA);
      emicp.setTargetCloud(cloudB);
      emicp.align(finalCloudem, initTransform);
      double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
      const Sophus::SE3d sicpTranform = emicp.getFinalTransFormation();
      const double e1 = semanticICPMetrics.evaluate(sicpTranform, indxTarget, indxSource, secs, emicp.getOuterIter());
      pcl::PointCloud<pcl::PointXYZL>::Ptr labeledCloudem(new pcl::PointCloud<pcl::PointXYZL>);
      emicp.getFusedLabels(labeledCloudem, sicpTranform);  // :193-195
      std::ostringstream name;
      name << pre << indxSource << ".pcd";
      pcl::io::savePCDFileASCII(name.str(), *labeledCloudem);  // :196-198

      pcl::PointCloud<pcl::PointXYZ>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZ>), cloudBnoL(new pcl::PointCloud<pcl::PointXYZ>);
      pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxSource], *cloudAnoL);
      pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[indxTarget], *cloudBnoL);
      pcl::PointCloud<pcl::PointXYZ>::Ptr finalCloudse3(new pcl::PointCloud<pcl::PointXYZ>);
      begin = std::chrono::steady_clock::now();
      gicpse3.setSourceCloud(cloudAnoL);
      gicpse3.setTargetCloud(cloudBnoL);
      gicpse3.align(finalCloudse3);
      secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
      const double e2 = se3GICPMetrics.evaluate(gicpse3.getFinalTransFormation(), indxTarget, indxSource, secs, gicpse3.getOuterIter());
      std::printf("pair %zu<-%zu  SICP MSE %.3e  se3GICP MSE %.3e\n", indxTarget, indxSource, e1, e2);
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 2;
  }
  std::printf("SICP FINAL MSE %.6e rot %.6e trans %.6e\n", semanticICPMetrics.getTransformMSE(), semanticICPMetrics.getRotMSE(), semanticICPMetrics.getTransMSE());
  std::printf("se3GICP FINAL MSE %.6e rot %.6e trans %.6e\n", se3GICPMetrics.getTransformMSE(), se3GICPMetrics.getRotMSE(), se3GICPMetrics.getTransMSE());
  return 0;
}
