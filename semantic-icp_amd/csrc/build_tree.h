// build_tree.h -- interface of the GPU tree build (build_tree.hip).
#ifndef SICP_BUILD_TREE_H_
#define SICP_BUILD_TREE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bvh.hpp"

namespace sicp {

// one label segment of a cloud (everything here is derived from point counts on the host)
struct BuildSegment {
  int off, cnt;        // device-order range of the segment
  int padded;          // cnt rounded up to whole leaves (>= one leaf)
  int pt_begin;        // first packed point
  int node_begin;      // first box
  int code_begin;      // first leaf code
  float lo[3], scale;  // curve quantisation of the segment (bounding box corner, cells per metre)
  TreeLevels lv;
};

// the same, as the device reads it (several segments are built by launches over ALL of them: the kernels find an element's
// segment by binary search of the begin offsets)
struct BuildSegmentDev {
  int off, cnt, padded, pt_begin, node_begin, code_begin, n_leaf, top;
  float lox, loy, loz, scale;
};

struct BuildBuffers {
  // in: the cloud in caller order, and (several segments only) caller indices grouped by segment
  const float *rx, *ry, *rz;
  const uint32_t* rl;  // nullable
  const int* ids;      // nullable = identity
  // scratch
  unsigned long long *keys_in, *keys_out;
  int *vals_in, *vals_out;
  void* sort_temp;
  size_t sort_temp_bytes;
  // several segments only: their descriptions on the device (uploaded from `h_segs`, pinned); the offset arrays are spare
  BuildSegmentDev* d_segs;
  BuildSegmentDev* h_segs;
  int *d_seg_begin, *d_seg_end;
  int *h_seg_begin, *h_seg_end;
  // out
  float *x, *y, *z;
  uint32_t* label;
  int *perm, *inv;
  float4 *pts4, *box_lo, *box_hi;
  unsigned long long* leaf_code;
};

// keys_in / keys_out / vals_in / vals_out hold `max_segment_points` entries for a one-segment cloud and ALL points for a
// cloud of several segments (every segment sorts its own range)
size_t build_sort_temp_bytes(int max_segment_points);
hipError_t build_tree_device(const BuildBuffers& b, const BuildSegment* segs, int n_seg, hipStream_t st);

}  // namespace sicp
#endif
