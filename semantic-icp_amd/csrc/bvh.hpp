// bvh.hpp -- host-side build of the exact-search structure the kNN kernels traverse.
//
// The reference searches with pcl::KdTreeFLANN (built in setSourceCloud / setTargetCloud,
// em_icp.h:50-66); here every cloud segment is stored in Morton order and covered by an implicit
// 4-ary tree of axis-aligned boxes over leaves of LEAF consecutive points:
//   level 0 : leaf j covers points [j*LEAF, (j+1)*LEAF) of the segment
//   level k : node j covers nodes [4j, 4j+4) of level k-1
// Boxes are exact float min/max of the member points, so the float32 box distance is a true lower
// bound of FLANN's float32 point distance (rounding is monotone) and pruning never drops a
// neighbour: the search is exact, not approximate.
#ifndef SICP_BVH_HPP_
#define SICP_BVH_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sicp {

constexpr int kLeaf = 16;        // points per leaf
constexpr int kFan = 4;          // children per node
constexpr int kMaxLevels = 14;   // 16 * 4^13 points
constexpr int kLutBits = 15;     // seed look-up table: top 5 bits per axis of the Morton code

struct TreeLevels {
  int n_levels;            // >= 1
  int off[kMaxLevels];     // node offset of each level inside the segment's box array
  int cnt[kMaxLevels];     // nodes per level
};

struct HostTree {
  int n = 0;                       // points in the segment
  TreeLevels lv;
  std::vector<float> box_lo;       // 4 floats per node (x y z pad), all levels
  std::vector<float> box_hi;
  std::vector<int> lut;            // 1 << kLutBits leaf indices
  float lo[3] = {0, 0, 0};
  float scale = 0;                 // quantisation: cell = (p - lo) * scale, 21 bits per axis
  int total_nodes() const { return (int)(box_lo.size() / 4); }
};

inline uint64_t expand21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

inline uint32_t quant21(float p, float lo, float scale) {
  float v = (p - lo) * scale;
  if (!(v > 0.f)) return 0;
  if (v >= 2097151.f) return 2097151u;
  return (uint32_t)v;
}

inline uint64_t morton63(float x, float y, float z, const float lo[3], float scale) {
  return expand21(quant21(x, lo[0], scale)) | (expand21(quant21(y, lo[1], scale)) << 1) |
         (expand21(quant21(z, lo[2], scale)) << 2);
}

// Orders the points of one segment (indices `ids`, caller order) by (Morton code, caller index)
// and builds the boxes.  On return `ids` is the device order of the segment.
inline void build_segment_tree(const float* x, const float* y, const float* z, std::vector<int>& ids, HostTree& t) {
  const int n = (int)ids.size();
  t.n = n;
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int e = 0; e < n; ++e) {
    const int i = ids[e];
    const float p[3] = {x[i], y[i], z[i]};
    for (int d = 0; d < 3; ++d) {
      if (e == 0 || p[d] < lo[d]) lo[d] = p[d];
      if (e == 0 || p[d] > hi[d]) hi[d] = p[d];
    }
  }
  float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  if (!(ext > 0.f) || !std::isfinite(ext)) ext = 1.f;
  t.lo[0] = lo[0]; t.lo[1] = lo[1]; t.lo[2] = lo[2];
  t.scale = 2097151.f / ext;
  std::vector<std::pair<uint64_t, int>> keyed(n);
  for (int e = 0; e < n; ++e) {
    const int i = ids[e];
    keyed[e] = {morton63(x[i], y[i], z[i], t.lo, t.scale), i};
  }
  std::sort(keyed.begin(), keyed.end());
  for (int e = 0; e < n; ++e) ids[e] = keyed[e].second;

  // levels
  TreeLevels& lv = t.lv;
  std::memset(&lv, 0, sizeof lv);
  int cnt = std::max(1, (n + kLeaf - 1) / kLeaf), off = 0, L = 0;
  for (;;) {
    lv.off[L] = off; lv.cnt[L] = cnt;
    off += cnt; ++L;
    if (cnt == 1 || L == kMaxLevels) break;
    cnt = (cnt + kFan - 1) / kFan;
  }
  lv.n_levels = L;
  t.box_lo.assign((size_t)off * 4, 0.f);
  t.box_hi.assign((size_t)off * 4, 0.f);
  const float inf = INFINITY;
  for (int j = 0; j < lv.cnt[0]; ++j) {
    float bl[3] = {inf, inf, inf}, bh[3] = {-inf, -inf, -inf};
    for (int e = j * kLeaf; e < std::min(n, (j + 1) * kLeaf); ++e) {
      const int i = ids[e];
      const float p[3] = {x[i], y[i], z[i]};
      for (int d = 0; d < 3; ++d) { bl[d] = std::min(bl[d], p[d]); bh[d] = std::max(bh[d], p[d]); }
    }
    for (int d = 0; d < 3; ++d) { t.box_lo[4 * (size_t)j + d] = bl[d]; t.box_hi[4 * (size_t)j + d] = bh[d]; }
  }
  for (int k = 1; k < L; ++k)
    for (int j = 0; j < lv.cnt[k]; ++j) {
      float bl[3] = {inf, inf, inf}, bh[3] = {-inf, -inf, -inf};
      for (int c = kFan * j; c < std::min(lv.cnt[k - 1], kFan * (j + 1)); ++c) {
        const size_t s = 4 * (size_t)(lv.off[k - 1] + c);
        for (int d = 0; d < 3; ++d) { bl[d] = std::min(bl[d], t.box_lo[s + d]); bh[d] = std::max(bh[d], t.box_hi[s + d]); }
      }
      const size_t o = 4 * (size_t)(lv.off[k] + j);
      for (int d = 0; d < 3; ++d) { t.box_lo[o + d] = bl[d]; t.box_hi[o + d] = bh[d]; }
    }
  // seed table: Morton prefix -> a leaf near that prefix (any leaf is a valid seed; a near one
  // gives a tight first bound)
  const int nl = 1 << kLutBits, shift = 63 - kLutBits;
  t.lut.assign(nl, 0);
  int e = 0;
  for (int p = 0; p < nl; ++p) {
    while (e < n && (int)(keyed[e].first >> shift) < p) ++e;
    t.lut[p] = std::min(std::max(0, n - 1), e) / kLeaf;
  }
}

}  // namespace sicp
#endif
