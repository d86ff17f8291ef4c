// semantic_point_cloud.h -- semanticicp::SemanticPointCloud with the reference's public surface
// (reference: semantic_icp/semantic_point_cloud.h:15-63, impl/semantic_point_cloud.hpp).
//
// On the MI355X engine a SemanticPointCloud owns ONE device-resident cloud -- all its label clouds, in semanticLabels
// order, as the per-label segments of a SICP_MODE_SEMANTIC cloud (search trees, normals) -- uploaded on first need and
// shared, not copied, with every SemanticIterativeClosestPoint::align() that registers it (sicp_share_cloud): the
// per-point covariances of addSemanticCloud (impl/semantic_point_cloud.hpp:25-84) are computed once per cloud, on the GPU,
// as in the reference they are computed once, at construction.  `labeledCovariances` is filled on its first read: the
// drivers (exec/nyu_eval.cc, exec/test_icp.cc) never read it, and 72 bytes per point would otherwise travel back for nothing.
#ifndef SEMANTIC_POINT_CLOUD_H_
#define SEMANTIC_POINT_CLOUD_H_
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "sicp_engine.hpp"

namespace semanticicp {

template <typename PointT, typename SemanticT>
class SemanticIterativeClosestPoint;

template <typename PointT, typename SemanticT>
class SemanticPointCloud {
 public:
  typedef std::shared_ptr<SemanticPointCloud<PointT, SemanticT>> Ptr;
  typedef std::shared_ptr<const SemanticPointCloud<PointT, SemanticT>> ConstPtr;
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef pcl::KdTreeFLANN<PointT> KdTree;
  typedef typename KdTree::Ptr KdTreePtr;
  typedef std::vector<Eigen::Matrix3d, Eigen::aligned_allocator<Eigen::Matrix3d>> MatricesVector;
  typedef std::shared_ptr<MatricesVector> MatricesVectorPtr;

  // std::map<SemanticT, MatricesVectorPtr> (semantic_point_cloud.h:38) whose readers trigger the fetch from the GPU.
  // A key exists from addSemanticCloud on (count / size / erase need nothing); its vector is filled when it is looked at.
  class CovarianceMap : public std::map<SemanticT, MatricesVectorPtr> {
    typedef std::map<SemanticT, MatricesVectorPtr> Base;
   public:
    typedef typename Base::iterator iterator;
    typedef typename Base::const_iterator const_iterator;
    MatricesVectorPtr& operator[](const SemanticT& k) { fetch(); return Base::operator[](k); }
    MatricesVectorPtr& at(const SemanticT& k) { fetch(); return Base::at(k); }
    const MatricesVectorPtr& at(const SemanticT& k) const { fetch(); return Base::at(k); }
    iterator find(const SemanticT& k) { fetch(); return Base::find(k); }
    const_iterator find(const SemanticT& k) const { fetch(); return Base::find(k); }
    iterator begin() { fetch(); return Base::begin(); }
    const_iterator begin() const { fetch(); return Base::begin(); }
    iterator end() { return Base::end(); }
    const_iterator end() const { return Base::end(); }
   private:
    friend class SemanticPointCloud;
    void fetch() const { if (owner_) owner_->materialise_covariances(); }
    SemanticPointCloud* owner_ = nullptr;
  };

  SemanticPointCloud(int k = 20, double epsilon = 0.001) : k_correspondences_(k), epsilon_(epsilon) { labeledCovariances.owner_ = this; }
  SemanticPointCloud(const SemanticPointCloud&) = delete;
  SemanticPointCloud& operator=(const SemanticPointCloud&) = delete;

  std::vector<SemanticT> semanticLabels;
  std::map<SemanticT, PointCloudPtr> labeledPointClouds;
  CovarianceMap labeledCovariances;
  std::map<SemanticT, KdTreePtr> labeledKdTrees;

  // reference: impl/semantic_point_cloud.hpp:12-86
  void addSemanticCloud(SemanticT label, PointCloudPtr cloud_ptr, bool computeKd = true, bool computeCov = true) {
    if (device_ == PRE_TRANSFORM) materialise_covariances();  // (what is still owed comes from the cloud as it was)
    semanticLabels.push_back(label);
    labeledPointClouds[label] = cloud_ptr;
    device_ = NONE;
    if (!computeKd) return;
    KdTreePtr tree(new KdTree());
    tree->setInputCloud(cloud_ptr);
    labeledKdTrees[label] = tree;
    if (!computeCov) return;
    MatricesVectorPtr mine(new MatricesVector());
    static_cast<std::map<SemanticT, MatricesVectorPtr>&>(labeledCovariances)[label] = mine;
    ours_[label] = mine.get();
    owed_.insert(label);
  }

  // reference: semantic_point_cloud.h:44-52
  void removeSemanticClass(SemanticT label) {
    auto it = std::find(semanticLabels.begin(), semanticLabels.end(), label);
    if (it != semanticLabels.end()) {
      if (device_ == PRE_TRANSFORM) materialise_covariances();
      semanticLabels.erase(it);
      labeledPointClouds.erase(label);
      labeledCovariances.erase(label);
      labeledKdTrees.erase(label);
      owed_.erase(label);
      ours_.erase(label);
      device_ = NONE;
    }
  }

  // reference: impl/semantic_point_cloud.hpp:88-103
  typename pcl::PointCloud<pcl::PointXYZL>::Ptr getpclPointCloud() {
    typename pcl::PointCloud<pcl::PointXYZL>::Ptr out(new pcl::PointCloud<pcl::PointXYZL>());
    for (SemanticT s : semanticLabels)
      for (const PointT& p : *(labeledPointClouds[s])) {
        pcl::PointXYZL t;
        t.x = p.x; t.y = p.y; t.z = p.z; t.label = uint32_t(s);
        out->push_back(t);
      }
    return out;
  }

  // reference: impl/semantic_point_cloud.hpp:105-111 (the points move, the covariances stay what they were)
  void transform(Eigen::Matrix4f trans) {
    if (!owed_.empty() && device_ == NONE) upload();  // the covariances still owed belong to the cloud as it is NOW
    for (SemanticT s : semanticLabels) pcl::transformPointCloud(*(labeledPointClouds[s]), *(labeledPointClouds[s]), trans);
    if (device_ == CURRENT) device_ = PRE_TRANSFORM;
  }

  int getK() const { return k_correspondences_; }
  double getEpsilon() const { return epsilon_; }

 private:
  friend class SemanticIterativeClosestPoint<PointT, SemanticT>;
  friend class CovarianceMap;

  // The handle whose SICP_SOURCE slot is this cloud on the GPU, as it is now (uploaded when it is not there yet).
  sicp_handle device() {
    if (device_ != CURRENT) {
      if (device_ == PRE_TRANSFORM) materialise_covariances();
      upload();
    } else {
      int32_t n = 0;
      detail::check(sicp_cloud_size(engine_.get(), SICP_SOURCE, &n, nullptr), engine_.get(), "sicp_cloud_size");
      push_supplied_covariances(engine_.get(), n);  // (a vector the caller assigned since the upload)
    }
    return engine_.get();
  }

  void upload() {
    sicp_handle h = engine_.get();
    sicp_params p;
    detail::check(sicp_default_params(SICP_MODE_SEMANTIC, &p), h, "sicp_default_params");
    p.k_cov = k_correspondences_;
    p.epsilon = epsilon_;
    p.reuse_features = 1;  // once per cloud (impl/semantic_point_cloud.hpp:25-84: at construction)
    detail::check(sicp_set_params(h, &p), h, "sicp_set_params");
    detail::FlatCloud f;
    size_t n = 0;
    for (SemanticT s : semanticLabels) n += labeledPointClouds[s]->size();
    f.x.reserve(n); f.y.reserve(n); f.z.reserve(n); f.label.reserve(n);
    // label clouds concatenated in semanticLabels order == the order the reference iterates them (impl/semantic_icp.hpp:48)
    for (SemanticT s : semanticLabels)
      for (const PointT& q : *(labeledPointClouds[s])) f.push(q.x, q.y, q.z, (uint32_t)s);
    detail::check(sicp_set_cloud(h, SICP_SOURCE, f.size(), f.x.data(), f.y.data(), f.z.data(), f.label.data()), h, "sicp_set_cloud");
    device_ = CURRENT;
    pushed_.clear();
    push_supplied_covariances(h, f.size());
  }

  // impl/semantic_icp.hpp:73,77 registers with whatever sits in labeledCovariances.  A vector that is not the one this class
  // made for its label -- the caller assigned its own, or filled the entry of a label added with computeCov = false -- is
  // handed to the engine (sicp_set_covariances): matrices of the form I - (1 - epsilon) n n^T keep the product kernels, other
  // symmetric matrices are evaluated as they are on the full-matrix path (one pair at a time), and what is no covariance at
  // all (not symmetric, not finite) is REFUSED -- check() throws, nothing is silently replaced.  (Matrices edited in place inside
  // a vector this class made are not seen: assign a vector.)
  void push_supplied_covariances(sicp_handle h, int n) {
    std::map<SemanticT, MatricesVectorPtr>& plain = labeledCovariances;
    bool any = false;
    for (SemanticT s : semanticLabels) {
      auto it = plain.find(s);
      if (it == plain.end() || !it->second || owed_.count(s)) continue;
      auto mine = ours_.find(s);
      const bool foreign = mine == ours_.end() || mine->second != it->second.get();
      auto sent = pushed_.find(s);
      if (foreign && (sent == pushed_.end() || sent->second != it->second.get())) any = true;  // not handed over yet
    }
    if (!any || n == 0) return;
    std::vector<double> c9((size_t)n * 9);
    std::vector<std::pair<SemanticT, const MatricesVector*>> sent_now;
    detail::check(sicp_covariances(h, SICP_SOURCE, c9.data(), nullptr, nullptr, nullptr), h, "sicp_covariances");  // the labels that are the engine's own
    size_t at = 0;
    for (SemanticT s : semanticLabels) {
      const size_t m = labeledPointClouds[s]->size();
      auto it = plain.find(s);
      auto mine = ours_.find(s);
      if (it != plain.end() && it->second && !owed_.count(s) && (mine == ours_.end() || mine->second != it->second.get())) {
        if (it->second->size() != m) throw std::runtime_error("SemanticPointCloud: labeledCovariances of a label does not have one matrix per point");
        for (size_t i = 0; i < m; ++i)
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) c9[(at + i) * 9 + 3 * a + b] = (*it->second)[i](a, b);
        sent_now.push_back(std::make_pair(s, (const MatricesVector*)it->second.get()));
      }
      at += m;
    }
    detail::check(sicp_set_covariances(h, SICP_SOURCE, c9.data()), h, "sicp_set_covariances");  // (throws: a refused vector is offered again next time)
    for (const auto& e : sent_now) pushed_[e.first] = e.second;
  }

  // labeledCovariances[label] for every label that still owes them: one read-back of what the GPU computed per segment
  void materialise_covariances() {
    if (owed_.empty()) return;
    if (device_ == NONE) upload();
    sicp_handle h = engine_.get();
    int32_t n = 0;
    detail::check(sicp_cloud_size(h, SICP_SOURCE, &n, nullptr), h, "sicp_cloud_size");
    std::vector<double> c9((size_t)n * 9);
    if (n > 0) detail::check(sicp_covariances(h, SICP_SOURCE, c9.data(), nullptr, nullptr, nullptr), h, "sicp_covariances");
    std::map<SemanticT, MatricesVectorPtr>& plain = labeledCovariances;
    size_t at = 0;
    for (SemanticT s : semanticLabels) {
      const size_t m = labeledPointClouds[s]->size();
      if (owed_.count(s)) {
        MatricesVectorPtr& v = plain[s];
        if (!v) v = MatricesVectorPtr(new MatricesVector());
        v->resize(m);
        if (m > 0) std::memcpy(detail::matrix3d_storage(v->data()), &c9[at * 9], sizeof(double) * 9 * m);  // (bit-symmetric: rows == columns)
      }
      at += m;
    }
    owed_.clear();
  }

  int k_correspondences_;
  double epsilon_;
  enum DeviceCopy { NONE, CURRENT, PRE_TRANSFORM };  // what the engine's copy of this cloud is
  DeviceCopy device_ = NONE;
  std::set<SemanticT> owed_;  // labels whose covariance vector has not been fetched yet
  std::map<SemanticT, const MatricesVector*> pushed_;  // the caller's vector of a label the engine's copy of the cloud holds
  std::map<SemanticT, const MatricesVector*> ours_;  // the vector this class made for a label (anything else there is the caller's)
  detail::Engine engine_;
};

}  // namespace semanticicp
#endif  // SEMANTIC_POINT_CLOUD_H_
