// dropin_calls.cc -- wall time of every call of the reference drivers' per-pair sequences, through the class shims
// (the drop-in path): exec/kitti_eval.cc:176-211 (EmIterativeClosestPoint<11>, then GICP with the previous pair's kd-tree and
// covariances) and exec/nyu_eval.cc:118-146 (pcl_2_semantic of both frames, SemanticIterativeClosestPoint).  Every object is
// constructed per pair, as the drivers do.  usage: dropin_calls kitti|nyu a.pcd b.pcd [cm.txt] [repetitions]
// One JSON line: microseconds per call, median over the repetitions (the first one is a warm-up and is dropped).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <em_icp.h>
#include <gicp.h>
#include <pcl_2_semantic.h>
#include <semantic_icp.h>
#include <semantic_point_cloud.h>
#include "examples/eval_support.h"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Timer {
  std::map<std::string, std::vector<double>> t;
  std::vector<std::string> order;
  double t0 = 0;
  void start() { t0 = now_us(); }
  void lap(const char* name) {
    const double t1 = now_us();
    if (!t.count(name)) order.push_back(name);
    t[name].push_back(t1 - t0);
    t0 = now_us();
  }
  void print(const char* what, int n_a, int n_b) {
    std::printf("{\"sequence\": \"%s\", \"points\": [%d, %d], \"us_per_call_median\": {", what, n_a, n_b);
    double total = 0;
    for (size_t i = 0; i < order.size(); ++i) {
      std::vector<double> v(t[order[i]].begin() + 1, t[order[i]].end());
      std::sort(v.begin(), v.end());
      const double med = v[v.size() / 2];
      total += med;
      std::printf("%s\"%s\": %.1f", i ? ", " : "", order[i].c_str(), med);
    }
    std::printf("}, \"us_per_pair\": %.1f}\n", total);
  }
};

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: dropin_calls kitti|nyu a.pcd b.pcd [cm.txt] [reps]\n"); return 2; }
  const bool kitti = !std::strcmp(argv[1], "kitti");
  const int reps = argc > 5 ? std::atoi(argv[5]) : 8;
  pcl::PointCloud<pcl::PointXYZL>::Ptr cloudA(new pcl::PointCloud<pcl::PointXYZL>), cloudB(new pcl::PointCloud<pcl::PointXYZL>);
  if (pcl::io::loadPCDFile<pcl::PointXYZL>(argv[2], *cloudA) == -1 || pcl::io::loadPCDFile<pcl::PointXYZL>(argv[3], *cloudB) == -1) return 3;
  Timer T;
  try {
    if (kitti) {
      Eigen::Matrix<double, 11, 11> cm = evalsupport::ReadConfusionMatrix<11>(argv[4]);
      pcl::PointCloud<pcl::PointXYZ>::Ptr cloudAnoL(new pcl::PointCloud<pcl::PointXYZ>), cloudBnoL(new pcl::PointCloud<pcl::PointXYZ>);
      pcl::io::loadPCDFile<pcl::PointXYZ>(argv[2], *cloudAnoL);
      pcl::io::loadPCDFile<pcl::PointXYZ>(argv[3], *cloudBnoL);
      // exec/kitti_eval.cc:117-122: the first scan's kd-tree and covariances
      semanticicp::GICP<pcl::PointXYZ> first;
      first.setTargetCloud(cloudBnoL);
      auto kdtree = first.getTargetKdTree();
      auto covs = first.getTargetCovariances();
      for (int r = 0; r <= reps; ++r) {
        Sophus::SE3d init;
        {
          T.start();
          semanticicp::EmIterativeClosestPoint<11> emicp;
          pcl::PointCloud<pcl::PointXYZL>::Ptr fin(new pcl::PointCloud<pcl::PointXYZL>);
          T.lap("em: construct");
          emicp.setSourceCloud(cloudA); T.lap("em: setSourceCloud");
          emicp.setTargetCloud(cloudB); T.lap("em: setTargetCloud");
          emicp.setConfusionMatrix(cm); T.lap("em: setConfusionMatrix");
          emicp.align(fin, init); T.lap("em: align (+ final cloud)");
          (void)emicp.getFinalTransFormation();
          T.start();
        }
        T.lap("em: destruct");
        {
          T.start();
          semanticicp::GICP<pcl::PointXYZ> gicp;
          pcl::PointCloud<pcl::PointXYZ>::Ptr fin(new pcl::PointCloud<pcl::PointXYZ>);
          T.lap("gicp: construct");
          gicp.setSourceCloud(cloudAnoL, kdtree, covs); T.lap("gicp: setSourceCloud(cloud, kdtree, covs)");
          gicp.setTargetCloud(cloudBnoL); T.lap("gicp: setTargetCloud");
          gicp.align(fin); T.lap("gicp: align (+ final cloud)");
          kdtree = gicp.getTargetKdTree(); T.lap("gicp: getTargetKdTree");
          covs = gicp.getTargetCovariances(); T.lap("gicp: getTargetCovariances");
          T.start();
        }
        T.lap("gicp: destruct");
      }
      T.print("exec/kitti_eval.cc:176-211 per pair (EM-ICP<11>, then SE3-GICP)", (int)cloudA->size(), (int)cloudB->size());
    } else {
      typedef semanticicp::SemanticPointCloud<pcl::PointXYZ, uint32_t> SemCloud;
      for (int r = 0; r <= reps; ++r) {
        T.start();
        std::shared_ptr<SemCloud> semanticA(new SemCloud()), semanticB(new SemCloud());
        T.lap("construct 2 SemanticPointClouds");
        semanticicp::pcl_2_semantic(cloudA, semanticA); T.lap("pcl_2_semantic(A)");
        semanticicp::pcl_2_semantic(cloudB, semanticB); T.lap("pcl_2_semantic(B)");
        {
          semanticicp::SemanticIterativeClosestPoint<pcl::PointXYZ, uint32_t> sicp;
          T.lap("sicp: construct");
          sicp.setInputSource(semanticA);
          sicp.setInputTarget(semanticB); T.lap("sicp: setInputSource/Target");
          sicp.align(semanticA); T.lap("sicp: align (+ transform of the source in place)");
          (void)sicp.getFinalTransFormation();
          T.start();
        }
        T.lap("sicp: destruct");
        semanticA.reset(); semanticB.reset();
        T.lap("destruct 2 SemanticPointClouds");
      }
      T.print("exec/nyu_eval.cc:118-146 per pair (pcl_2_semantic x2, SemanticICP)", (int)cloudA->size(), (int)cloudB->size());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "dropin_calls: %s\n", e.what());
    return 1;
  }
  return 0;
}
