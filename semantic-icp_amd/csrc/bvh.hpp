// bvh.hpp -- shape of the exact-search structure the kNN kernels traverse (built on the GPU by
// build_tree.hip).
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
constexpr int kMaxLevels = 12;   // 16 * 4^11 = 67 M points per cloud segment

struct TreeLevels {
  int n_levels;            // >= 1
  int off[kMaxLevels];     // node offset of each level inside the segment's box array
  int cnt[kMaxLevels];     // nodes per level
};

// 63-bit 3-D Hilbert index of the 21-bit cell coordinates (Skilling, "Programming the Hilbert
// curve", 2004: axes -> transposed index, then bit interleave).  Unlike Morton order, cells that
// are consecutive on the curve are always face neighbours, so LEAF consecutive points (and the
// 64-point seed group) are always spatially compact.  SICP_HD is empty on the host and
// __host__ __device__ when the kernel files include this header, so both sides share one definition.
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

// The curve index at BITS bits per axis, aligned with the 63-bit indices: the Hilbert index is
// hierarchical (the index of the coarse cell is a prefix of the index of every fine cell inside it),
// so this is the first 63-bit index of the coarse cell -- all a seed search needs, at BITS/21 of the
// cost (the unrolled 21-bit transform is ~900 instructions).
template <int BITS>
SICP_HD inline uint64_t curve_code_coarse(float x, float y, float z, float lox, float loy, float loz, float scale) {
  uint32_t X[3] = {quant21_cell(x, lox, scale) >> (21 - BITS), quant21_cell(y, loy, scale) >> (21 - BITS),
                   quant21_cell(z, loz, scale) >> (21 - BITS)};
  const uint32_t M = 1u << (BITS - 1);
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
  uint64_t h = 0;
  for (int b = BITS - 1; b >= 0; --b)
    h = (h << 3) | (uint64_t)((((X[0] >> b) & 1u) << 2) | (((X[1] >> b) & 1u) << 1) | ((X[2] >> b) & 1u));
  return h << (3 * (21 - BITS));
}

// Levels of the implicit tree over n points.  The tree is COMPLETE: 4^top leaves (top = the smallest
// height that holds the ceil(n / 16) real ones), 4^(top - L) nodes on level L, leaves first.  Leaves
// beyond the real ones hold sentinel points (+inf, caller index -1) and, like every node without a real
// point below it, an inverted box (lo = +inf, hi = -inf: its distance bound is +inf).  So a walk needs
// no node counts and no clamps, and a level's offset is arithmetic (level_offset) instead of a table
// load in the dependent chain of every node visit.
inline TreeLevels make_levels(int n) {
  TreeLevels lv;
  std::memset(&lv, 0, sizeof lv);
  const int real_leaves = std::max(1, (n + kLeaf - 1) / kLeaf);
  int top = 0;
  while ((1 << (2 * top)) < real_leaves && top + 1 < kMaxLevels) ++top;
  int off = 0;
  for (int L = 0; L <= top; ++L) {
    lv.off[L] = off; lv.cnt[L] = 1 << (2 * (top - L));
    off += lv.cnt[L];
  }
  lv.n_levels = top + 1;
  return lv;
}
// offset of level L in a complete tree of height top: sum of 4^(top - i), i < L
SICP_HD inline int level_offset(int top, int L) {
  const unsigned t4 = 1u << (2 * (top + 1));
  return (int)((t4 - (t4 >> (2 * L))) / 3u);
}
inline int total_nodes(const TreeLevels& lv) { return lv.off[lv.n_levels - 1] + lv.cnt[lv.n_levels - 1]; }

}  // namespace sicp
#endif
