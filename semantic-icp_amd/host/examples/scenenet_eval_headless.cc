// scenenet_eval_headless.cc -- the SceneNet experiment loop of the reference
// (exec/scenenet_eval.cc:110-250) on the MI355X engine: consecutive frames (target n, source n+1) of
// the PCD files in -s; EM-ICP<13>(20, 1e-6) from the identity (:174-186), its fused labels written
// as <out-prefix><source index>.pcd (:193-198), SE3-GICP(20, 1e-6) (:206-214); both scored against
// the ground truth in -t with the SceneNet row format (exec/scenenet_metrics.h:17-29) and written as
// CSV rows to <prefix>EMICPscenenet.csv / <prefix>se3GICPscenenet.csv.
// Not reproduced: the pcl::GeneralizedIterativeClosestPoint comparison and the disabled bootstrap.
// -b <pairs>: register that many frame pairs at a time in lock step (alignBatch); same rows and files.
// -S <in flight>: the whole sequence as an OPEN STREAM per method (sicp_stream_*): every frame read, uploaded and indexed
// once (it is the source of one registration and the target of the next), up to <in flight> registrations sharing the
// GPU; the fused labels of a pair come back WITH its registration (SICP_SUBMIT_FUSED_LABELS: one more K = 4 search and
// the label kernel when it retires -- no second pass over the sequence).  Same rows and files.
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
  if (const char* sarg = arg(argc, argv, "-S")) {
    try {
      const int in_flight = std::max(1, std::atoi(sarg));
      std::vector<size_t> starts;
      for (size_t n = 0; n + STEP < pcd_fns.size(); n += STEP) starts.push_back(n);
      double cmv[169];
      for (int r = 0; r < 13; ++r)
        for (int c = 0; c < 13; ++c) cmv[13 * r + c] = cm(r, c);
      MethodStream em, gi;
      em.open(SICP_MODE_EM, 13, cmv, in_flight, pcd_fns.size(), starts.size(), device_from_env(), 1e-6);   // Em(20, 1e-6), :174
      gi.open(SICP_MODE_GICP, 0, nullptr, in_flight, pcd_fns.size(), starts.size(), device_from_env(), 1e-6);  // Gicp(20, 1e-6), :206
      std::vector<pcl::PointCloud<pcl::PointXYZL>::Ptr> frame(pcd_fns.size());  // kept: the label files repeat the source's points
      auto upload = [&](size_t k) {
        if (em.cloud_of_scan[k]) return true;
        frame[k].reset(new pcl::PointCloud<pcl::PointXYZL>);
        if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[k], *frame[k]) == -1) return false;
        em.add(k, *frame[k], true);
        pcl::PointCloud<pcl::PointXYZ>::Ptr raw(new pcl::PointCloud<pcl::PointXYZ>);
        pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[k], *raw);
        gi.add(k, *raw, false);
        return true;
      };
      const auto begin = std::chrono::steady_clock::now();
      for (size_t q = 0; q < starts.size(); ++q) {
        const size_t t = starts[q], sidx = t + STEP;
        if (!upload(t) || !upload(sidx)) { std::cerr << "Couldn't read frame file\n"; return -1; }
        em.submit(q, t, sidx, SICP_SUBMIT_FUSED_LABELS);
        gi.submit(q, t, sidx);
        sicp_stream_release_cloud(em.s, em.cloud_of_scan[t]);
        sicp_stream_release_cloud(gi.s, gi.cloud_of_scan[t]);
        if (q % 16 == 15) { em.collect(0); gi.collect(0); }
      }
      em.collect(2);
      gi.collect(2);
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(std::max<size_t>(1, starts.size()));
      for (size_t q = 0; q < starts.size(); ++q) {
        const size_t t = starts[q], sidx = t + STEP;
        if (em.result_of_pair[q].status != SICP_OK || gi.result_of_pair[q].status != SICP_OK)
          throw std::runtime_error(std::string("registration failed: ") + sicp_strerror(em.result_of_pair[q].status ? em.result_of_pair[q].status : gi.result_of_pair[q].status));
        const double e1 = semanticICPMetrics.evaluate(semanticicp::detail::to_se3(em.result_of_pair[q].qt), t, sidx, secs, em.result_of_pair[q].outer_iters);
        const std::vector<uint32_t> lab = em.labels(q, frame[sidx]->size());   // getFusedLabels(out, final pose), :193-195
        pcl::PointCloud<pcl::PointXYZL> labeled;
        for (size_t i = 0; i < frame[sidx]->size(); ++i) {
          pcl::PointXYZL p = frame[sidx]->points[i];
          p.label = lab[i];
          labeled.push_back(p);
        }
        std::ostringstream name;
        name << pre << sidx << ".pcd";
        pcl::io::savePCDFileASCII(name.str(), labeled);  // :196-198
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
