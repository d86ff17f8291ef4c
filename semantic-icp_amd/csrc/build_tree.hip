// build_tree.hip -- GPU build of the search structure of one cloud (the role of
// pcl::KdTreeFLANN::setInputCloud in setSourceCloud / setTargetCloud, em_icp.h:50-66).
//
// Per label segment: 63-bit Hilbert index of every point -> stable radix sort of (index, caller
// index) pairs -> gather into curve order -> exact float boxes of the 16-point leaves -> boxes of
// the upper levels of the implicit 4-ary tree.  The sort is rocPRIM's device radix sort (the one
// library primitive in this engine; it is not on the align() path); everything else is written
// here.  The result is identical to the host build in bvh.hpp (same curve function, stable sort).
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#define SICP_HD __host__ __device__
#include "build_tree.h"

namespace sicp {

namespace {

__global__ __launch_bounds__(256) void codes_kernel(int cnt, const int* __restrict__ ids, int id_base, const float* __restrict__ rx,
                                                    const float* __restrict__ ry, const float* __restrict__ rz, float lox, float loy,
                                                    float loz, float scale, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cnt) return;
  const int i = ids ? ids[e] : id_base + e;
  keys[e] = curve_code(rx[i], ry[i], rz[i], lox, loy, loz, scale);
  vals[e] = i;
}

// curve order: SoA + packed points (+ padding) + index maps of one segment
__global__ __launch_bounds__(256) void gather_kernel(int cnt, int padded, int seg_off, int pt_begin, const int* __restrict__ sorted,
                                                     const float* __restrict__ rx, const float* __restrict__ ry,
                                                     const float* __restrict__ rz, const uint32_t* __restrict__ rl, float* __restrict__ x,
                                                     float* __restrict__ y, float* __restrict__ z, uint32_t* __restrict__ label,
                                                     int* __restrict__ perm, int* __restrict__ inv, float4* __restrict__ pts4) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= padded) return;
  if (e < cnt) {
    const int i = sorted[e];
    const float px = rx[i], py = ry[i], pz = rz[i];
    const int d = seg_off + e;
    x[d] = px; y[d] = py; z[d] = pz;
    if (rl) label[d] = rl[i];
    perm[d] = i;
    inv[i] = d;
    pts4[pt_begin + e] = make_float4(px, py, pz, __uint_as_float((unsigned)i));
  } else {
    pts4[pt_begin + e] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));
  }
}

__global__ __launch_bounds__(256) void leaf_box_kernel(int n_leaf, int cnt, int pt_begin, int node_begin, int code_begin,
                                                       const float4* __restrict__ pts4, const unsigned long long* __restrict__ sorted_keys,
                                                       float4* __restrict__ box_lo, float4* __restrict__ box_hi,
                                                       unsigned long long* __restrict__ leaf_code) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_leaf) return;
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  const int e0 = j * kLeaf, e1 = min(cnt, e0 + kLeaf);
  for (int e = e0; e < e1; ++e) {
    const float4 p = pts4[pt_begin + e];
    lx = fminf(lx, p.x); ly = fminf(ly, p.y); lz = fminf(lz, p.z);
    hx = fmaxf(hx, p.x); hy = fmaxf(hy, p.y); hz = fmaxf(hz, p.z);
  }
  box_lo[node_begin + j] = make_float4(lx, ly, lz, 0.f);
  box_hi[node_begin + j] = make_float4(hx, hy, hz, 0.f);
  leaf_code[code_begin + j] = e0 < cnt ? sorted_keys[e0] : ~0ull;
}

__global__ __launch_bounds__(256) void level_box_kernel(int n_nodes, int child_cnt, int node_off, int child_off, float4* __restrict__ box_lo,
                                                        float4* __restrict__ box_hi) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_nodes) return;
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  const int c0 = kFan * j, c1 = min(child_cnt, c0 + kFan);
  for (int c = c0; c < c1; ++c) {
    const float4 lo = box_lo[child_off + c], hi = box_hi[child_off + c];
    lx = fminf(lx, lo.x); ly = fminf(ly, lo.y); lz = fminf(lz, lo.z);
    hx = fmaxf(hx, hi.x); hy = fmaxf(hy, hi.y); hz = fmaxf(hz, hi.z);
  }
  box_lo[node_off + j] = make_float4(lx, ly, lz, 0.f);
  box_hi[node_off + j] = make_float4(hx, hy, hz, 0.f);
}

// The narrow upper levels of a segment's tree, from level `first` on, by ONE workgroup: level after level with a barrier in
// between (a level has a quarter of the nodes of the one below: from 1024 nodes down the whole rest is one round of the
// workgroup per level).  One launch instead of six at 100K points -- and per label segment of a SICP_MODE_SEMANTIC cloud.
// The level just written is the next one's input, and the levels lie back to back in memory: a cache line that holds the last
// boxes of the level below may have been pulled into this CU's L1 while the first boxes of this level -- the same line -- were
// not written yet.  So the children are read PAST the L1 (agent-scope loads; the stores are write-through and drained by the
// barrier's s_waitcnt, the workgroup sits on one CU of one XCD: its L2 is the meeting point).  (A first version that kept plain
// loads and put release / acquire fences around the barrier took 85 us per call: the fences write back and invalidate whole
// caches, seven times.)
constexpr int kUpperThreads = 1024, kUpperMaxNodes = 1024;
__device__ __forceinline__ float4 load_box_past_l1(const float4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
__global__ __launch_bounds__(kUpperThreads) void upper_levels_kernel(TreeLevels lv, int first, int node_begin, float4* __restrict__ box_lo,
                                                                     float4* __restrict__ box_hi) {
  for (int k = first; k < lv.n_levels; ++k) {
    const int n_nodes = lv.cnt[k], child_cnt = lv.cnt[k - 1], node_off = node_begin + lv.off[k], child_off = node_begin + lv.off[k - 1];
    for (int j = threadIdx.x; j < n_nodes; j += kUpperThreads) {
      float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
      const int c0 = kFan * j, c1 = min(child_cnt, c0 + kFan);
      for (int c = c0; c < c1; ++c) {
        const float4 lo = load_box_past_l1(box_lo + child_off + c), hi = load_box_past_l1(box_hi + child_off + c);
        lx = fminf(lx, lo.x); ly = fminf(ly, lo.y); lz = fminf(lz, lo.z);
        hx = fmaxf(hx, hi.x); hy = fmaxf(hy, hi.y); hz = fmaxf(hz, hi.z);
      }
      box_lo[node_off + j] = make_float4(lx, ly, lz, 0.f);
      box_hi[node_off + j] = make_float4(hx, hy, hz, 0.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's (write-through) stores have reached L2 ...
    __syncthreads();                                   // ... before any wave reads them past the L1
  }
}

// ---- several segments (SICP_MODE_SEMANTIC: one per label): every stage ONE launch over all segments.  Round 6: a 13-label
// frame of 307 200 points took 3.2 ms of host time per upload as one sort + six launches PER SEGMENT.
// segment of element e of a concatenation whose parts begin at begins[0 .. n_seg): the last part that begins at or before e
template <class F>
__device__ __forceinline__ int find_segment(const BuildSegmentDev* __restrict__ segs, int n_seg, int e, F begin_of) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (begin_of(segs[mid]) <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void codes_all_kernel(int n, int n_seg, const BuildSegmentDev* __restrict__ segs, const int* __restrict__ ids,
                                                        const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const BuildSegmentDev g = segs[find_segment(segs, n_seg, e, [](const BuildSegmentDev& s) { return s.off; })];
  const int i = ids[e];
  keys[e] = curve_code(rx[i], ry[i], rz[i], g.lox, g.loy, g.loz, g.scale);
  vals[e] = i;
}

__global__ __launch_bounds__(256) void gather_all_kernel(int pt_total, int n_seg, const BuildSegmentDev* __restrict__ segs, const int* __restrict__ sorted,
                                                         const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                         const uint32_t* __restrict__ rl, float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ z, uint32_t* __restrict__ label, int* __restrict__ perm,
                                                         int* __restrict__ inv, float4* __restrict__ pts4) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pt_total) return;
  const BuildSegmentDev g = segs[find_segment(segs, n_seg, p, [](const BuildSegmentDev& s) { return s.pt_begin; })];
  const int e = p - g.pt_begin;
  if (e < g.cnt) {
    const int i = sorted[g.off + e];
    const float px = rx[i], py = ry[i], pz = rz[i];
    const int d = g.off + e;
    x[d] = px; y[d] = py; z[d] = pz;
    if (rl) label[d] = rl[i];
    perm[d] = i;
    inv[i] = d;
    pts4[p] = make_float4(px, py, pz, __uint_as_float((unsigned)i));
  } else {
    pts4[p] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));
  }
}

__global__ __launch_bounds__(256) void leaf_box_all_kernel(int leaf_total, int n_seg, const BuildSegmentDev* __restrict__ segs,
                                                           const float4* __restrict__ pts4, const unsigned long long* __restrict__ sorted_keys,
                                                           float4* __restrict__ box_lo, float4* __restrict__ box_hi,
                                                           unsigned long long* __restrict__ leaf_code) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= leaf_total) return;
  const BuildSegmentDev g = segs[find_segment(segs, n_seg, q, [](const BuildSegmentDev& s) { return s.code_begin; })];
  const int j = q - g.code_begin;
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  const int e0 = j * kLeaf, e1 = min(g.cnt, e0 + kLeaf);
  for (int e = e0; e < e1; ++e) {
    const float4 p = pts4[g.pt_begin + e];
    lx = fminf(lx, p.x); ly = fminf(ly, p.y); lz = fminf(lz, p.z);
    hx = fmaxf(hx, p.x); hy = fmaxf(hy, p.y); hz = fmaxf(hz, p.z);
  }
  box_lo[g.node_begin + j] = make_float4(lx, ly, lz, 0.f);
  box_hi[g.node_begin + j] = make_float4(hx, hy, hz, 0.f);
  leaf_code[q] = e0 < g.cnt ? sorted_keys[g.off + e0] : ~0ull;
}

// the narrow upper levels of EVERY segment: workgroup s builds those of segment s (upper_levels_kernel's loop; a complete
// 4-ary tree: level L has 4^(top - L) nodes at level_offset(top, L))
__global__ __launch_bounds__(kUpperThreads) void upper_levels_all_kernel(const BuildSegmentDev* __restrict__ segs, float4* __restrict__ box_lo,
                                                                         float4* __restrict__ box_hi) {
  const BuildSegmentDev g = segs[blockIdx.x];
  int first = 1;
  while (first <= g.top && (1 << (2 * (g.top - first))) > kUpperMaxNodes) ++first;
  for (int k = first; k <= g.top; ++k) {
    const int n_nodes = 1 << (2 * (g.top - k)), child_cnt = 1 << (2 * (g.top - k + 1));
    const int node_off = g.node_begin + level_offset(g.top, k), child_off = g.node_begin + level_offset(g.top, k - 1);
    for (int j = threadIdx.x; j < n_nodes; j += kUpperThreads) {
      float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
      const int c0 = kFan * j, c1 = min(child_cnt, c0 + kFan);
      for (int c = c0; c < c1; ++c) {
        const float4 lo = load_box_past_l1(box_lo + child_off + c), hi = load_box_past_l1(box_hi + child_off + c);
        lx = fminf(lx, lo.x); ly = fminf(ly, lo.y); lz = fminf(lz, lo.z);
        hx = fmaxf(hx, hi.x); hy = fmaxf(hy, hi.y); hz = fmaxf(hz, hi.z);
      }
      box_lo[node_off + j] = make_float4(lx, ly, lz, 0.f);
      box_hi[node_off + j] = make_float4(hx, hy, hz, 0.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

}  // namespace

size_t build_sort_temp_bytes(int max_segment_points) {
  size_t bytes = 0;
  unsigned long long* k = nullptr;
  int* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(max_segment_points > 0 ? max_segment_points : 1), 0, 63,
                                  (hipStream_t) nullptr);
  return bytes;
}

// several segments: one launch per stage over all of them, the sorts one per segment
static hipError_t build_segments_together(const BuildBuffers& b, const BuildSegment* segs, int n_seg, hipStream_t st) {
  int n = 0, pt_total = 0, leaf_total = 0;
  for (int s = 0; s < n_seg; ++s) {
    const BuildSegment& g = segs[s];
    BuildSegmentDev& d = b.h_segs[s];
    d.off = g.off; d.cnt = g.cnt; d.padded = g.padded; d.pt_begin = g.pt_begin; d.node_begin = g.node_begin; d.code_begin = g.code_begin;
    d.n_leaf = g.lv.cnt[0]; d.top = g.lv.n_levels - 1;
    d.lox = g.lo[0]; d.loy = g.lo[1]; d.loz = g.lo[2]; d.scale = g.scale;
    n += g.cnt; pt_total += g.padded; leaf_total += g.lv.cnt[0];
  }
  hipError_t e = hipMemcpyAsync(b.d_segs, b.h_segs, sizeof(BuildSegmentDev) * n_seg, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return e;
  if (n > 0) {
    hipLaunchKernelGGL(codes_all_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, n_seg, b.d_segs, b.ids, b.rx, b.ry, b.rz, b.keys_in, b.vals_in);
    // one plain radix sort per segment, each on its own range of the shared key / value buffers.  (rocPRIM's SEGMENTED radix
    // sort was tried first: it gives every segment to ONE workgroup, and a 13-label frame whose largest label has ~100K points
    // then sorts for 4 ms instead of 0.4.)
    for (int s2 = 0; s2 < n_seg; ++s2) {
      const BuildSegment& g = segs[s2];
      if (g.cnt <= 0) continue;
      size_t tmp = b.sort_temp_bytes;
      e = rocprim::radix_sort_pairs(b.sort_temp, tmp, b.keys_in + g.off, b.keys_out + g.off, b.vals_in + g.off, b.vals_out + g.off, (size_t)g.cnt, 0, 63, st);
      if (e != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL(gather_all_kernel, dim3((pt_total + 255) / 256), dim3(256), 0, st, pt_total, n_seg, b.d_segs, b.vals_out, b.rx, b.ry, b.rz, b.rl, b.x,
                     b.y, b.z, b.label, b.perm, b.inv, b.pts4);
  hipLaunchKernelGGL(leaf_box_all_kernel, dim3((leaf_total + 255) / 256), dim3(256), 0, st, leaf_total, n_seg, b.d_segs, b.pts4, b.keys_out, b.box_lo, b.box_hi,
                     b.leaf_code);
  for (int s = 0; s < n_seg; ++s) {  // the wide levels of the big segments: a launch of their own each
    const BuildSegment& g = segs[s];
    for (int k = 1; k < g.lv.n_levels && g.lv.cnt[k] > kUpperMaxNodes; ++k)
      hipLaunchKernelGGL(level_box_kernel, dim3((g.lv.cnt[k] + 255) / 256), dim3(256), 0, st, g.lv.cnt[k], g.lv.cnt[k - 1], g.node_begin + g.lv.off[k],
                         g.node_begin + g.lv.off[k - 1], b.box_lo, b.box_hi);
  }
  hipLaunchKernelGGL(upper_levels_all_kernel, dim3(n_seg), dim3(kUpperThreads), 0, st, b.d_segs, b.box_lo, b.box_hi);
  return hipGetLastError();
}

hipError_t build_tree_device(const BuildBuffers& b, const BuildSegment* segs, int n_seg, hipStream_t st) {
  if (n_seg > 1 && b.ids && b.d_segs) return build_segments_together(b, segs, n_seg, st);
  for (int s = 0; s < n_seg; ++s) {
    const BuildSegment& g = segs[s];
    if (g.cnt <= 0) {
      // an empty segment still owns one padded leaf and one (empty) box
      if (g.padded > 0)
        hipLaunchKernelGGL(gather_kernel, dim3((g.padded + 255) / 256), dim3(256), 0, st, 0, g.padded, g.off, g.pt_begin, b.vals_out, b.rx, b.ry,
                           b.rz, b.rl, b.x, b.y, b.z, b.label, b.perm, b.inv, b.pts4);
      hipLaunchKernelGGL(leaf_box_kernel, dim3(1), dim3(256), 0, st, g.lv.cnt[0], 0, g.pt_begin, g.node_begin, g.code_begin, b.pts4, b.keys_out,
                         b.box_lo, b.box_hi, b.leaf_code);
      continue;
    }
    const int* ids = b.ids ? b.ids + g.off : nullptr;
    hipLaunchKernelGGL(codes_kernel, dim3((g.cnt + 255) / 256), dim3(256), 0, st, g.cnt, ids, g.off, b.rx, b.ry, b.rz, g.lo[0], g.lo[1], g.lo[2],
                       g.scale, b.keys_in, b.vals_in);
    size_t tmp = b.sort_temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(b.sort_temp, tmp, b.keys_in, b.keys_out, b.vals_in, b.vals_out, (size_t)g.cnt, 0, 63, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_kernel, dim3((g.padded + 255) / 256), dim3(256), 0, st, g.cnt, g.padded, g.off, g.pt_begin, b.vals_out, b.rx, b.ry,
                       b.rz, b.rl, b.x, b.y, b.z, b.label, b.perm, b.inv, b.pts4);
    hipLaunchKernelGGL(leaf_box_kernel, dim3((g.lv.cnt[0] + 255) / 256), dim3(256), 0, st, g.lv.cnt[0], g.cnt, g.pt_begin, g.node_begin,
                       g.code_begin, b.pts4, b.keys_out, b.box_lo, b.box_hi, b.leaf_code);
    int k = 1;
    for (; k < g.lv.n_levels && g.lv.cnt[k] > kUpperMaxNodes; ++k)  // the wide levels: a launch of their own
      hipLaunchKernelGGL(level_box_kernel, dim3((g.lv.cnt[k] + 255) / 256), dim3(256), 0, st, g.lv.cnt[k], g.lv.cnt[k - 1],
                         g.node_begin + g.lv.off[k], g.node_begin + g.lv.off[k - 1], b.box_lo, b.box_hi);
    if (k < g.lv.n_levels)  // everything above them: one workgroup
      hipLaunchKernelGGL(upper_levels_kernel, dim3(1), dim3(kUpperThreads), 0, st, g.lv, k, g.node_begin, b.box_lo, b.box_hi);
  }
  return hipGetLastError();
}

}  // namespace sicp
