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
// -G <n>: the sequence sharded over the GPUs of the node (BASELINE config 5; exec/kitti_eval.cc:124-249 is the loop that
// shards): the pair list is cut into n contiguous runs, each an open stream per method with its own host thread on device
// g % (devices visible), no exchange between them; the rows are merged in pair order.  Same rows again.  -S sets the
// registrations in flight per stream (default 64).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <em_icp.h>
#include <gicp.h>

#include "eval_support.h"

namespace {

using evalsupport::MethodStream;

// One contiguous run [q0, q1) of the pair list through two streams (EM-ICP and SE3-GICP) on one device: every scan of the run
// read, uploaded and indexed once, its registrations submitted as soon as it is there.  Runs in its own host thread under -G;
// nothing is shared between runs but the read-only file list.
struct ShardResult {
  size_t q0 = 0;
  int device = 0;
  std::vector<sicp_stream_result> em, gi;  // per pair of the run
  double secs_per_pair = 0;
  std::string error;
};

void run_shard(ShardResult& out, int device, int in_flight, const double* cmv, const std::vector<std::string>& pcd_fns,
               const std::vector<size_t>& starts, size_t q0, size_t q1) {
  out.q0 = q0;
  out.device = device;
  try {
    const size_t n_pairs = q1 - q0;
    if (n_pairs == 0) return;
    MethodStream em, gi;
    em.open(SICP_MODE_EM, 11, cmv, in_flight, pcd_fns.size(), n_pairs, device);
    gi.open(SICP_MODE_GICP, 0, nullptr, in_flight, pcd_fns.size(), n_pairs, device);
    auto upload = [&](size_t scan) {
      if (em.cloud_of_scan[scan]) return true;
      pcl::PointCloud<pcl::PointXYZL>::Ptr cl(new pcl::PointCloud<pcl::PointXYZL>);
      if (pcl::io::loadPCDFile<pcl::PointXYZL>(pcd_fns[scan], *cl) == -1) return false;
      evalsupport::filterRange(cl, 40.0);  // :138, :159
      em.add(scan, *cl, true);
      // the reference re-loads the files as PointXYZ (:201-204): same points, no labels, no range filter
      pcl::PointCloud<pcl::PointXYZ>::Ptr raw(new pcl::PointCloud<pcl::PointXYZ>);
      pcl::io::loadPCDFile<pcl::PointXYZ>(pcd_fns[scan], *raw);
      gi.add(scan, *raw, false);
      return true;
    };
    const auto begin = std::chrono::steady_clock::now();
    for (size_t q = q0; q < q1; ++q) {
      const size_t t = starts[q], sidx = t + 3;
      if (!upload(t) || !upload(sidx)) throw std::runtime_error("couldn't read scan file " + pcd_fns[t] + " / " + pcd_fns[sidx]);
      em.submit(q - q0, t, sidx);
      gi.submit(q - q0, t, sidx);
      // scan t has now been the source of pair q - 1 and the target of pair q: the caller is done with it
      sicp_stream_release_cloud(em.s, em.cloud_of_scan[t]);
      sicp_stream_release_cloud(gi.s, gi.cloud_of_scan[t]);
      if ((q - q0) % 16 == 15) { em.collect(0); gi.collect(0); }
    }
    em.collect(2);
    gi.collect(2);
    out.secs_per_pair = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count() / double(n_pairs);
    out.em = em.result_of_pair;
    out.gi = gi.result_of_pair;
  } catch (const std::exception& e) {
    out.error = e.what();
  }
}

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
  const char* garg = arg(argc, argv, "-G");
  if (sarg || garg) {
    try {
      const int in_flight = std::max(1, sarg ? std::atoi(sarg) : 64);
      std::vector<size_t> starts;
      for (size_t n = 0; n + 3 < pcd_fns.size(); n += 3) starts.push_back(n);  // exec/kitti_eval.cc:124-129
      double cmv[121];
      for (int r = 0; r < 11; ++r)
        for (int c = 0; c < 11; ++c) cmv[11 * r + c] = cm(r, c);
      // -G <n>: the pair list cut into n contiguous runs, one host thread + one stream per method for each, on device
      // g % (devices visible) -- independent scan pairs shard across the GPUs of a node without any exchange; a scan on
      // the border of two runs is uploaded by both.  One run = -S.
      int n_dev = 1;
      if (sicp_device_count(&n_dev) != SICP_OK || n_dev < 1) throw std::runtime_error(std::string("sicp_device_count: ") + sicp_strerror(SICP_ERR_NO_DEVICE));
      const char* dev_env = std::getenv("SICP_DEVICE");
      const int shards = std::max(1, std::min<int>(garg ? std::atoi(garg) : 1, (int)std::max<size_t>(1, starts.size())));
      std::vector<ShardResult> shard(shards);
      std::vector<std::thread> threads;
      for (int g = 0; g < shards; ++g) {
        const size_t q0 = starts.size() * g / shards, q1 = starts.size() * (g + 1) / shards;
        const int device = garg ? g % n_dev : (dev_env ? std::atoi(dev_env) : 0);
        auto work = [&, g, q0, q1, device] { run_shard(shard[g], device, in_flight, cmv, pcd_fns, starts, q0, q1); };
        if (shards == 1) work(); else threads.emplace_back(work);
      }
      for (std::thread& t : threads) t.join();
      for (int g = 0; g < shards; ++g)
        if (!shard[g].error.empty()) throw std::runtime_error("run " + std::to_string(g) + ": " + shard[g].error);
      // the rows, in pair order whatever run produced them
      for (int g = 0; g < shards; ++g) {
        const ShardResult& R = shard[g];
        for (size_t k = 0; k < R.em.size(); ++k) {
          const size_t t = starts[R.q0 + k], sidx = t + 3;
          if (R.em[k].status != SICP_OK || R.gi[k].status != SICP_OK)
            throw std::runtime_error(std::string("registration failed: ") + sicp_strerror(R.em[k].status ? R.em[k].status : R.gi[k].status));
          const double e1 = semanticICPMetrics.evaluate(semanticicp::detail::to_se3(R.em[k].qt), t, sidx, R.secs_per_pair, R.em[k].outer_iters);
          const double e2 = se3GICPMetrics.evaluate(semanticicp::detail::to_se3(R.gi[k].qt), t, sidx, R.secs_per_pair, R.gi[k].outer_iters);
          std::printf("pair %zu<-%zu  SICP MSE %.3e  se3GICP MSE %.3e\n", t, sidx, e1, e2);
        }
        if (shards > 1) std::printf("run %d: pairs [%zu, %zu) on device %d, %.3f s per pair\n", g, R.q0, R.q0 + R.em.size(), R.device, R.secs_per_pair);
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
