// nyu_eval_headless.cc -- the NYU experiment loop of the reference (exec/nyu_eval.cc:45-222) on the
// MI355X engine: for every row of the test file -t (indices into the sorted PCD files of -s) and
// every consecutive pair (source n, target n+1) of the row, SemanticICP on the labelled clouds
// (:126-139) and SemanticICP on the same clouds with all labels set to 0 ("single class",
// :150-181), both from the identity; each result is scored by the label agreement of the
// registered source with the target (exec/nyu_metrics.h:36-84) into <prefix>SICPnyu.csv /
// <prefix>se3GICPnyu.csv (+ Label<num>-*, Matrix*).  The reference names its outputs by date; here
// the prefix is -o.  Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison.
// -S <in flight>: every pair of the test file through two OPEN STREAMS (labelled / single class; sicp_stream_*,
// SICP_MODE_SEMANTIC): a frame is read, uploaded, grouped by label and indexed once per stream however many pairs
// name it, up to <in flight> registrations share the GPU, and the metrics are evaluated in the file's order once the
// poses are back.  Same lines, same files.
#include <chrono>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

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
  if (const char* sarg = arg(argc, argv, "-S")) {
    try {
      const int in_flight = std::max(1, std::atoi(sarg));
      NYUMetrics semanticICPMetrics(test, pre + "SICPnyu.csv", 0, numClasses);
      NYUMetrics se3GICPMetrics(test, pre + "se3GICPnyu.csv", 0, numClasses);
      std::vector<std::pair<size_t, size_t>> pairs;   // (source, target), exec/nyu_eval.cc:95-99
      while (semanticICPMetrics.morePairs()) {
        const std::vector<size_t> row = semanticICPMetrics.getPairs();
        for (size_t n = 0; n + 1 < row.size(); ++n) {
          if (row[n] >= pcd_fns.size() || row[n + 1] >= pcd_fns.size()) { std::cerr << "pair index beyond the PCD files\n"; return -1; }
          pairs.push_back({row[n], row[n + 1]});
        }
      }
      MethodStream lab, nol;
      lab.open(SICP_MODE_SEMANTIC, 0, nullptr, in_flight, pcd_fns.size(), pairs.size(), device_from_env());
      nol.open(SICP_MODE_SEMANTIC, 0, nullptr, in_flight, pcd_fns.size(), pairs.size(), device_from_env());
      std::vector<pcl::PointCloud<pcl::PointXYZL>::Ptr> frame(pcd_fns.size());
      auto upload = [&](size_t k) {
        if (lab.cloud_of_scan[k]) return true;
        frame[k].reset(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[k], *frame[k]) == -1) return false;
        lab.add(k, *frame[k], true);
        pcl::PointCloud<pcl::PointXYZL> zero;   // the same cloud with every label set to 0 (:147-170)
        for (pcl::PointXYZL p : frame[k]->points) { p.label = 0; zero.push_back(p); }
        nol.add(k, zero, true);
        return true;
      };
      const auto begin = std::chrono::steady_clock::now();
      for (size_t q = 0; q < pairs.size(); ++q) {
        if (!upload(pairs[q].first) || !upload(pairs[q].second)) { std::cerr << "Couldn't read frame file\n"; return -1; }
        lab.submit(q, pairs[q].second, pairs[q].first);
        nol.submit(q, pairs[q].second, pairs[q].first);
        if (q % 16 == 15) { lab.collect(0); nol.collect(0); }
      }
      lab.collect(2);
      nol.collect(2);
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(std::max<size_t>(1, pairs.size()));
      for (size_t q = 0; q < pairs.size(); ++q) {
        const size_t indxS = pairs[q].first, indxT = pairs[q].second;
        std::string numSource;
        for (char c : pcd_fns[indxS].substr(pcd_fns[indxS].find_last_of('/') + 1))
          if (c >= '0' && c <= '9') numSource.push_back(c);
        if (numSource.empty()) numSource = "0";
        int which = 0;
        for (MethodStream* M : {&lab, &nol}) {
          const sicp_stream_result& r = M->result_of_pair[q];
          if (r.status != SICP_OK) throw std::runtime_error(std::string("registration failed: ") + sicp_strerror(r.status));
          const Sophus::SE3d T = semanticicp::detail::to_se3(r.qt);
          pcl::PointCloud<pcl::PointXYZL>::Ptr moved(new pcl::PointCloud<pcl::PointXYZL>());
          pcl::transformPointCloud(*frame[indxS], *moved, (T.matrix()).cast<float>());  // :141-142
          const double acc = (which == 0 ? semanticICPMetrics : se3GICPMetrics).evaluate(moved, frame[indxT], numSource);
          const double* qd = T.data();
          std::printf("pair %zu->%zu %s pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g accuracy %.6f time %.3f\n", indxS, indxT,
                      which == 0 ? "SICP" : "se3GICP", qd[0], qd[1], qd[2], qd[3], qd[4], qd[5], qd[6], acc, secs);
          ++which;
        }
      }
    } catch (const std::exception& e) {
      std::cerr << "error: " << e.what() << "\n";
      return 2;
    }
    return 0;
  }
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
