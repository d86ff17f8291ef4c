// bvh.hpp -- host-side build of the exact-search structure the kNN kernels traverse.
//
// The reference searches with pcl::KdTreeFLANN (built in setSourceCloud / setTargetCloud,
// em_icp.h:50-66); here every cloud segment is stored in Hilbert-curve order and covered by an implicit
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
  std::vector<uint64_t> leaf_code; // curve index of the first point of every leaf (seed search)
  float lo[3] = {0, 0, 0};
  float scale = 0;                 // quantisation: cell = (p - lo) * scale, 21 bits per axis
  int total_nodes() const { return (int)(box_lo.size() / 4); }
};

// 63-bit 3-D Hilbert index of the 21-bit cell coordinates (Skilling, "Programming the Hilbert
// curve", 2004: axes -> transposed index, then bit interleave).  Unlike Morton order, cells that
// are consecutive on the curve are always face neighbours, so LEAF consecutive points (and the
// 64-point seed group) are always spatially compact.  SICP_HD is empty on the host and
// __host__ __device__ when kernels.hip includes this header, so both sides share one definition.
#ifndef SICP_HD
#define SICP_HD
#endif
SICP_HD inline uint64_t hilbert63_cells(uint32_t cx, uint32_t cy, uint32_t cz) {
  uint32_t X[3] = {cx, cy, cz};
  const uint32_t M = 1u << 20;
  for (uint32_t Q = M; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) {
        X[0] ^= P;
      } else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  X[1] ^= X[0];
  X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = M; Q > 1; Q >>= 1)
    if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  // interleave: X[0] holds the most significant bit of every 3-bit digit
  uint64_t h = 0;
  for (int b = 20; b >= 0; --b)
    h = (h << 3) | (uint64_t)((((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u));
  return h;
}

SICP_HD inline uint32_t quant21_cell(float p, float lo, float scale) {
  const float v = (p - lo) * scale;
  return !(v > 0.f) ? 0u : (v >= 2097151.f ? 2097151u : (uint32_t)v);
}

SICP_HD inline uint64_t curve_code(float x, float y, float z, float lox, float loy, float loz, float scale) {
  return hilbert63_cells(quant21_cell(x, lox, scale), quant21_cell(y, loy, scale), quant21_cell(z, loz, scale));
}

// Orders the points of one segment (indices `ids`, caller order) by (curve index, caller index)
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
    keyed[e] = {curve_code(x[i], y[i], z[i], t.lo[0], t.lo[1], t.lo[2], t.scale), i};
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
  // seed search: a query's Morton code is located among these by binary search; the leaf found
  // is spatially adjacent to the query, which gives a tight first bound
  t.leaf_code.assign(lv.cnt[0], 0);
  for (int j = 0; j < lv.cnt[0]; ++j) t.leaf_code[j] = j * kLeaf < n ? keyed[(size_t)j * kLeaf].first : ~0ull;
}

}  // namespace sicp
#endif
