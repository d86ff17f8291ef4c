// test_icp_headless.cc -- the reference's manual integration run (exec/test_icp.cc:19-127) without
// the viewer and without the pcl::GeneralizedIterativeClosestPoint comparison: loads -s / -t PCD
// files, runs SemanticICP (:77-82), SE3-GICP (:94-100) and, when -m gives a confusion matrix file
// (one row per line, space separated, as exec/read_confusion_matrix.h reads), EM-ICP as
// exec/kitti_eval.cc:184-192 does.  Prints each final pose as `NAME qx qy qz qw tx ty tz iters`.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>

#include <em_icp.h>
#include <gicp.h>
#include <pcl_2_semantic.h>
#include <semantic_icp.h>
#include <semantic_point_cloud.h>

#ifndef EM_CLASSES
#define EM_CLASSES 4
#endif

static const char* arg(int argc, char** argv, const char* flag) {
  for (int i = 1; i + 1 < argc; ++i)
    if (!std::strcmp(argv[i], flag)) return argv[i + 1];
  return nullptr;
}

static void print_pose(const char* name, const Sophus::SE3d& T, int iters, double secs) {
  const double* d = T.data();
  std::printf("%s %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.3f\n", name, d[0], d[1], d[2], d[3], d[4], d[5], d[6], iters, secs);
}

int main(int argc, char** argv) {
  const char *fs = arg(argc, argv, "-s"), *ft = arg(argc, argv, "-t"), *fm = arg(argc, argv, "-m");
  if (!fs) { std::cout << "Need source file (-s)\n"; return -1; }
  if (!ft) { std::cout << "Need target file (-t)\n"; return -1; }
  pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
  if (pcl::io::loadPCDFile<pcl::PointXYZL>(fs, *cloudA) == -1) { std::cerr << "Couldn't read source\n"; return -1; }
  if (pcl::io::loadPCDFile<pcl::PointXYZL>(ft, *cloudB) == -1) { std::cerr << "Couldn't read target\n"; return -1; }
  typedef semanticicp::SemanticPointCloud<pcl::PointXYZ, uint32_t> SemCloud;
  try {
    std::shared_ptr<SemCloud> semanticAfinal(new SemCloud()), semanticA(new SemCloud()), semanticB(new SemCloud());
    semanticicp::pcl_2_semantic(cloudA, semanticAfinal);
    semanticicp::pcl_2_semantic(cloudA, semanticA);
    semanticicp::pcl_2_semantic(cloudB, semanticB);
    for (uint32_t drop : {3u, 10u, 11u}) {  // exec/test_icp.cc:53-55,73-75
      semanticA->removeSemanticClass(drop);
      semanticB->removeSemanticClass(drop);
    }
    semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> sicp;
    auto begin = std::chrono::steady_clock::now();
    sicp.setInputSource(semanticA);
    sicp.setInputTarget(semanticB);
    sicp.align(semanticAfinal);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
    print_pose("SEMANTIC", sicp.getFinalTransFormation(), 0, secs);

    pcl::PointCloud<pcl::PointXYZ>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZ>), cloudBnoL(new pcl::PointCloud<pcl::PointXYZ>);
    pcl::io::loadPCDFile<pcl::PointXYZ>(fs, *cloudAnoL);
    pcl::io::loadPCDFile<pcl::PointXYZ>(ft, *cloudBnoL);
    semanticicp::GICP<pcl::PointXYZ> gicpse3;
    pcl::PointCloud<pcl::PointXYZ>::Ptr finalCloudse3(new pcl::PointCloud<pcl::PointXYZ>);
    begin = std::chrono::steady_clock::now();
    gicpse3.setSourceCloud(cloudAnoL);
    gicpse3.setTargetCloud(cloudBnoL);
    gicpse3.align(finalCloudse3);
    secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
    print_pose("GICP", gicpse3.getFinalTransFormation(), gicpse3.getOuterIter(), secs);
    std::printf("GICP_FINAL_CLOUD %zu %.9g %.9g %.9g\n", finalCloudse3->size(), (*finalCloudse3)[0].x, (*finalCloudse3)[0].y, (*finalCloudse3)[0].z);

    if (fm) {
      Eigen::Matrix<double, EM_CLASSES, EM_CLASSES> cm;
      std::ifstream f(fm);
      std::string line;
      for (int r = 0; r < EM_CLASSES && std::getline(f, line); ++r) {
        std::istringstream ss(line);
        for (int c = 0; c < EM_CLASSES; ++c) ss >> cm(r, c);
      }
      semanticicp::EmIterativeClosestPoint<EM_CLASSES> emicp;
      pcl::PointCloud<pcl::PointXYZL>::Ptr finalCloudem(new pcl::PointCloud<pcl::PointXYZL>);
      Sophus::SE3d init;
      begin = std::chrono::steady_clock::now();
      emicp.setSourceCloud(cloudA);
      emicp.setTargetCloud(cloudB);
      emicp.setConfusionMatrix(cm);
      emicp.align(finalCloudem, init);
      secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
      print_pose("EM", emicp.getFinalTransFormation(), emicp.getOuterIter(), secs);
      pcl::PointCloud<pcl::PointXYZL>::Ptr fused(new pcl::PointCloud<pcl::PointXYZL>);
      emicp.getFusedLabels(fused, emicp.getFinalTransFormation());
      size_t same = 0;
      for (size_t i = 0; i < fused->size(); ++i) same += (*fused)[i].label == (*cloudA)[i].label;
      std::printf("EM_FUSED %zu %zu\n", fused->size(), same);
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 2;
  }
  return 0;
}
