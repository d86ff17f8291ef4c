// kitti_eval_headless.cc -- the KITTI experiment loop of the reference (exec/kitti_eval.cc:124-249)
// on the MI355X engine: for every stride-3 pair (target n, source n+3) of the PCD files in -s, run
// EM-ICP<11> (:184-192) and SE3-GICP (:211-217) from the identity, score them against the ground
// truth poses in -t (exec/kitti_metrics.h) and write one CSV row per pair and method to
// <prefix>EMICPkitti.csv / <prefix>se3GICPkitti.csv (-o prefix; the reference uses a date string).
// Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison (third-party algorithm), the
// disabled FPFH bootstrap, and quirk Q7 (the reference hands SE3-GICP the kd-tree of another scan,
// exec/kitti_eval.cc:213; here every align() sees its own clouds).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
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
  const Eigen::Matrix<double, 11, 11> cm = ReadConfusionMatrix<11>(cmf);
  const std::vector<std::string> pcd_fns = get_pcd_in_dir(dir);
  std::ofstream foutSICP(pre + "EMICPkitti.csv"), foutse3GICP(pre + "se3GICPkitti.csv");
  const std::string gtFile = gt;
  KittiMetrics semanticICPMetrics(gtFile, &foutSICP), se3GICPMetrics(gtFile, &foutse3GICP);
  try {
    // one engine per method for the whole run: device buffers and the captured solver graph are reused
    semanticicp::EmIterativeClosestPoint<11> emicp;
    semanticicp::GICP<pcl::PointXYZ> gicpse3;
    emicp.setConfusionMatrix(cm);
    for (size_t n = 0; n + 3 < pcd_fns.size(); n += 3) {  // exec/kitti_eval.cc:124-129
      const size_t indxTarget = n, indxSource = n + 3;
      pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
      if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxSource], *cloudA) == -1) { std::cerr << "Couldn't read source file\n"; return -1; }
      if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[indxTarget], *cloudB) == -1) { std::cerr << "Couldn't read target file\n"; return -1; }
      filterRange(cloudA, 40.0);  // :138
      filterRange(cloudB, 40.0);  // :159
      Sophus::SE3d initTransform;  // identity (:172-176)

      pcl::PointCloud<pcl::PointXYZL>::Ptr finalCloudem(new pcl::PointCloud<pcl::PointXYZL>);
      auto begin = std::chrono::steady_clock::now();
      emicp.setSourceCloud(cloudA);
      emicp.setTargetCloud(cloudB);
      emicp.align(finalCloudem, initTransform);
      double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
      const double e1 = semanticICPMetrics.evaluate(emicp.getFinalTransFormation(), indxTarget, indxSource, secs, emicp.getOuterIter());

      // the reference re-loads the files as PointXYZ (:201-204): same points, no labels, no range filter
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
