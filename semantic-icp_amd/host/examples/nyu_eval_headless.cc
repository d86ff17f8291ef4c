// nyu_eval_headless.cc -- the NYU experiment loop of the reference (exec/nyu_eval.cc:45-222) on the
// MI355X engine: for every row of the test file -t (indices into the sorted PCD files of -s) and
// every consecutive pair (source n, target n+1) of the row, SemanticICP on the labelled clouds
// (:126-139) and SemanticICP on the same clouds with all labels set to 0 ("single class",
// :150-181), both from the identity; each result is scored by the label agreement of the
// registered source with the target (exec/nyu_metrics.h:36-84) into <prefix>SICPnyu.csv /
// <prefix>se3GICPnyu.csv (+ Label<num>-*, Matrix*).  The reference names its outputs by date; here
// the prefix is -o.  Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison.
#include <chrono>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include <pcl_2_semantic.h>
#include <semantic_icp.h>
#include <semantic_point_cloud.h>

#include "eval_support.h"

int main(int argc, char** argv) {
  using namespace evalsupport;
  typedef semanticicp::SemanticPointCloud<pcl::PointXYZ, uint32_t> SemanticCloud;
  const char *dir = arg(argc, argv, "-s"), *test = arg(argc, argv, "-t"), *prefix = arg(argc, argv, "-o"), *nc = arg(argc, argv, "-c");
  if (!dir) { std::cout << "Need source directory (-s)\n"; return -1; }
  if (!test) { std::cout << "Need file (-t)\n"; return -1; }
  const std::string pre = prefix ? prefix : "";
  const size_t numClasses = nc ? (size_t)std::stoul(nc) : 895;  // nyu_metrics.h:19
  const std::vector<std::string> pcd_fns = get_pcd_in_dir(dir);
  try {
    NYUMetrics semanticICPMetrics(test, pre + "SICPnyu.csv", 0, numClasses);
    NYUMetrics se3GICPMetrics(test, pre + "se3GICPnyu.csv", 0, numClasses);
    while (semanticICPMetrics.morePairs()) {
      const std::vector<size_t> pairs = semanticICPMetrics.getPairs();
      for (size_t n = 0; n + 1 < pairs.size(); ++n) {
        const size_t indxS = pairs[n], indxT = pairs[n + 1];
        if (indxS >= pcd_fns.size() || indxT >= pcd_fns.size()) { std::cerr << "pair index beyond the PCD files\n"; return -1; }
        const std::string strSource = pcd_fns[indxS], strTarget = pcd_fns[indxT];
        // exec/nyu_eval.cc:103-107: the digits of the source file name are the cloud number
        std::string numSource;
        for (char c : strSource.substr(strSource.find_last_of('/') + 1))
          if (c >= '0' && c <= '9') numSource.push_back(c);
        if (numSource.empty()) numSource = "0";
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(strSource, *cloudA) == -1) { std::cerr << "Couldn't read source file\n"; return -1; }
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(strTarget, *cloudB) == -1) { std::cerr << "Couldn't read target file\n"; return -1; }
        std::shared_ptr<SemanticCloud> semanticA(new SemanticCloud()), semanticB(new SemanticCloud());
        semanticicp::pcl_2_semantic(cloudA, semanticA);
        semanticicp::pcl_2_semantic(cloudB, semanticB);
        Sophus::SE3d initTransform;  // identity (:121-125)

        semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> sicp;
        sicp.setInputSource(semanticA);
        sicp.setInputTarget(semanticB);
        auto begin = std::chrono::steady_clock::now();
        sicp.align(semanticA, initTransform);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
        const Sophus::SE3d sicpTranform = sicp.getFinalTransFormation();
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudASICP(new pcl::PointCloud<pcl::PointXYZL>());
        pcl::transformPointCloud(*cloudA, *cloudASICP, (sicpTranform.matrix()).cast<float>());  // :141-142
        const double acc1 = semanticICPMetrics.evaluate(cloudASICP, cloudB, numSource);
        const double* q = sicpTranform.data();
        std::printf("pair %zu->%zu SICP pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g accuracy %.6f time %.3f\n", indxS, indxT, q[0], q[1],
                    q[2], q[3], q[4], q[5], q[6], acc1, secs);

        // the same clouds with every label set to 0 (:147-170)
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZL>()), cloudBnoL(new pcl::PointCloud<pcl::PointXYZL>());
        for (pcl::PointXYZL p : cloudA->points) { p.label = 0; cloudAnoL->push_back(p); }
        for (pcl::PointXYZL p : cloudB->points) { p.label = 0; cloudBnoL->push_back(p); }
        std::shared_ptr<SemanticCloud> semanticAnoL(new SemanticCloud()), semanticBnoL(new SemanticCloud());
        semanticicp::pcl_2_semantic(cloudAnoL, semanticAnoL);
        semanticicp::pcl_2_semantic(cloudBnoL, semanticBnoL);
        semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> sicp2;
        sicp2.setInputSource(semanticAnoL);
        sicp2.setInputTarget(semanticBnoL);
        begin = std::chrono::steady_clock::now();
        sicp2.align(semanticAnoL, initTransform);
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
        const Sophus::SE3d gicpTransform = sicp2.getFinalTransFormation();
        pcl::PointCloud<pcl::PointXYZL>::Ptr cloudAse3GICP(new pcl::PointCloud<pcl::PointXYZL>());
        pcl::transformPointCloud(*cloudA, *cloudAse3GICP, (gicpTransform.matrix()).cast<float>());
        const double acc2 = se3GICPMetrics.evaluate(cloudAse3GICP, cloudB, numSource);
        q = gicpTransform.data();
        std::printf("pair %zu->%zu se3GICP pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g accuracy %.6f time %.3f\n", indxS, indxT, q[0],
                    q[1], q[2], q[3], q[4], q[5], q[6], acc2, secs);
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 2;
  }
  return 0;
}
