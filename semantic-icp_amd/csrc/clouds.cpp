// clouds.cpp -- host side of a cloud: staging (one pass over the caller's points), upload, search-tree build, feature buffers.
#include "engine.hpp"

namespace sicp {
namespace host {

int set_device(sicp_context* h) {
  HIPCHECK(hipSetDevice(h->device));
  return SICP_OK;
}

// ---- cloud layout -----------------------------------------------------------------------------
// A cloud's upload is recorded in an event on the uploading handle's stream, and the cloud may outlive
// that handle (it is shared, or goes back to the pool).  Waiting for an event whose stream has been
// destroyed is not safe with this runtime (it intermittently answers "event last recorded in a
// capturing stream"), so a handle settles every cloud it lets go of while its streams still exist.
void settle_cloud(Cloud& c) {
  if (c.pending && c.ready_ev) (void)hipEventSynchronize(c.ready_ev);
  c.pending = false;
}

int cloud_wait(sicp_context* h, Cloud& c) {
  if (c.pending && h->wait_on_device) {  // (the flag stays up: whoever needs the host copy of `perm` still waits on the host)
    HIPCHECK(hipStreamWaitEvent(h->stream, c.ready_ev, 0));
    return SICP_OK;
  }
  if (c.pending) {
    HIPCHECK(hipEventSynchronize(c.ready_ev));
    c.pending = false;
  }
  return SICP_OK;
}

int stage_cloud(sicp_context* h, Cloud& c, int32_t n, const StridedCloud& in) {
  if ((long long)n > ((long long)sicp::kLeaf << (2 * (sicp::kMaxLevels - 1)))) {
    h->last_error = "cloud too large for the search tree (16 * 4^11 = 67 M points per cloud)";
    return SICP_ERR_INVALID_ARGUMENT;
  }
  SICPCHECK(cloud_wait(h, c));  // a previous upload may still be reading the staging buffers
  // Non-finite points (the NaNs of an organized RGB-D cloud) never enter the device cloud:
  // pcl::KdTreeFLANN::setInputCloud (em_icp.h:50-66) leaves them out of the search index, so the
  // reference can neither find them as neighbours nor -- a NaN query keeps no candidate -- match them.
  // Everything below works on the finite points; outputs are mapped back to the caller's indices.
  c.n_caller = n;
  c.keep.clear(); c.drop_i.clear(); c.drop_xyz.clear();
  c.has_label = in.label != nullptr;
  HIPCHECK(c.hx.resize(n)); HIPCHECK(c.hy.resize(n)); HIPCHECK(c.hz.resize(n)); HIPCHECK(c.hl.resize(in.label ? n : 0));
  const float inf = std::numeric_limits<float>::infinity();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  uint32_t lmin = 0xffffffffu, lmax = 0;
  auto ld = [](const char* base, long long stride, int i) { float v; std::memcpy(&v, base + (long long)i * stride, sizeof v); return v; };
  int m = 0;  // finite points so far: they are stored compacted as they come
  float* const hx = c.hx.data(); float* const hy = c.hy.data(); float* const hz = c.hz.data();
  uint32_t* const hl = c.hl.data();
  for (int i = 0; i < n; ++i) {
    const float px = ld(in.x, in.stride, i), py = ld(in.y, in.stride, i), pz = ld(in.z, in.stride, i);
    // (x - x is 0 for a finite x and NaN otherwise: one test for the three coordinates)
    const float t = (px - px) + (py - py) + (pz - pz);
    if (t == 0.f) {
      hx[m] = px; hy[m] = py; hz[m] = pz;
      lo[0] = px < lo[0] ? px : lo[0]; hi[0] = px > hi[0] ? px : hi[0];
      lo[1] = py < lo[1] ? py : lo[1]; hi[1] = py > hi[1] ? py : hi[1];
      lo[2] = pz < lo[2] ? pz : lo[2]; hi[2] = pz > hi[2] ? pz : hi[2];
      if (in.label) {
        uint32_t lb; std::memcpy(&lb, in.label + (long long)i * in.label_stride, sizeof lb);
        hl[m] = lb;
        lmin = lb < lmin ? lb : lmin; lmax = lb > lmax ? lb : lmax;
      }
      if (m != i) c.keep.push_back(i);  // (only once a point has been dropped; completed below)
      ++m;
    } else {
      if (c.keep.empty() && c.drop_i.empty()) {  // the first dropped point: the kept ones so far map to themselves
        c.keep.reserve(n);
        for (int k = 0; k < m; ++k) c.keep.push_back(k);
      }
      c.drop_i.push_back(i);
      c.drop_xyz.push_back(px); c.drop_xyz.push_back(py); c.drop_xyz.push_back(pz);
    }
  }
  if (m != n) {  // (sizes follow the finite points; keep[] has one entry per kept point)
    HIPCHECK(c.hx.resize(m)); HIPCHECK(c.hy.resize(m)); HIPCHECK(c.hz.resize(m)); HIPCHECK(c.hl.resize(in.label ? m : 0));
  }
  c.n = m;
  c.label_min = lmin; c.label_max = lmax;
  for (int d = 0; d < 3; ++d) { c.bb_lo[d] = lo[d]; c.bb_hi[d] = hi[d]; }
  c.bb_valid = true;
  c.is_set = true;
  c.layout = -1;
  c.feat_valid = false;
  c.cov_general = false;
  c.proj_valid = false;
  return SICP_OK;
}
int stage_cloud(sicp_context* h, Cloud& c, int32_t n, const float* x, const float* y, const float* z, const uint32_t* label) {
  const StridedCloud in = {(const char*)x, (const char*)y, (const char*)z, (const char*)label, 4, 4};
  return stage_cloud(h, c, n, in);
}

int prepare_cloud(sicp_context* h, Cloud& c) {
  const int want = h->params.mode == SICP_MODE_SEMANTIC ? 1 : 0;
  if (!c.is_set) return SICP_ERR_NOT_READY;
  if (want == 1 && !c.has_label) return SICP_ERR_NOT_READY;
  SICPCHECK(cloud_wait(h, c));  // an upload still in flight (possibly queued by another handle or host thread)
  if (c.layout == want) return SICP_OK;
  const int n = c.n;
  // ---- host: segment membership and per-segment bounding boxes (one pass over the cloud)
  c.seg_label.clear();
  std::vector<int> which(want ? n : 0), counts;
  const float inf = std::numeric_limits<float>::infinity();
  std::vector<float> lo, hi;  // [3 per segment]
  if (want == 0) {
    c.seg_label.push_back(0);
    counts.push_back(n);
    lo.assign(3, inf); hi.assign(3, -inf);
    if (c.bb_valid) {  // one segment: its box came with the staging pass
      for (int d = 0; d < 3; ++d) { lo[d] = c.bb_lo[d]; hi[d] = c.bb_hi[d]; }
    } else {
      for (int i = 0; i < n; ++i) {
        const float p[3] = {c.hx[i], c.hy[i], c.hz[i]};
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[d]); }
      }
    }
  } else {
    // pcl_2_semantic.h:24-39: one sub-cloud per label, labels in order of first appearance -- and, in the same pass over the
    // points, every segment's bounding box.  (The previous point's segment first: a cloud that arrives label by label --
    // SemanticPointCloud's -- or in image order, where runs of one label are long, costs one compare per point instead of
    // one per label seen so far.)
    int last = -1;
    for (int i = 0; i < n; ++i) {
      const uint32_t l = c.hl[i];
      int sidx = (last >= 0 && c.seg_label[(size_t)last] == l) ? last : -1;
      for (size_t k = 0; sidx < 0 && k < c.seg_label.size(); ++k)
        if (c.seg_label[k] == l) sidx = (int)k;
      if (sidx < 0) {
        sidx = (int)c.seg_label.size();
        c.seg_label.push_back(l); counts.push_back(0);
        lo.insert(lo.end(), 3, inf); hi.insert(hi.end(), 3, -inf);
      }
      which[i] = sidx;
      counts[sidx]++;
      last = sidx;
      float* L = &lo[3 * (size_t)sidx];
      float* H = &hi[3 * (size_t)sidx];
      const float px = c.hx[i], py = c.hy[i], pz = c.hz[i];
      L[0] = std::min(L[0], px); L[1] = std::min(L[1], py); L[2] = std::min(L[2], pz);
      H[0] = std::max(H[0], px); H[1] = std::max(H[1], py); H[2] = std::max(H[2], pz);
    }
  }
  const int n_seg = (int)c.seg_label.size();
  c.seg_off.assign(n_seg + 1, 0);
  c.trees.assign(n_seg, Cloud::SegTree());
  std::vector<sicp::BuildSegment> segs(n_seg);
  int pt_total = 0, node_total = 0, code_total = 0, max_cnt = 1;
  for (int sg = 0; sg < n_seg; ++sg) {
    sicp::BuildSegment& g = segs[sg];
    g.off = c.seg_off[sg]; g.cnt = counts[sg];
    c.seg_off[sg + 1] = g.off + g.cnt;
    g.lv = sicp::make_levels(g.cnt);
    g.padded = g.lv.cnt[0] * sicp::kLeaf;  // every leaf of the complete tree owns 16 point slots (sentinels beyond the real points)
    g.pt_begin = pt_total; g.node_begin = node_total; g.code_begin = code_total;
    pt_total += g.padded; node_total += sicp::total_nodes(g.lv); code_total += g.lv.cnt[0];
    float ext = 0.f;
    for (int d = 0; d < 3; ++d) { g.lo[d] = g.cnt > 0 ? lo[3 * sg + d] : 0.f; if (g.cnt > 0) ext = std::max(ext, hi[3 * sg + d] - lo[3 * sg + d]); }
    if (!(ext > 0.f) || !std::isfinite(ext)) ext = 1.f;
    g.scale = 2097151.f / ext;
    max_cnt = std::max(max_cnt, g.cnt);
    Cloud::SegTree& st = c.trees[sg];
    st.lv = g.lv; st.n = g.cnt; st.pt_begin = g.pt_begin; st.node_begin = g.node_begin; st.code_begin = g.code_begin;
    st.lo[0] = g.lo[0]; st.lo[1] = g.lo[1]; st.lo[2] = g.lo[2]; st.scale = g.scale;
  }
  HostBuf<int>& ids = c.h_ids;
  if (want) {  // caller indices grouped by segment, cloud order inside a segment
    HIPCHECK(ids.resize(n));
    std::vector<int> fill(c.seg_off.begin(), c.seg_off.end() - 1);
    for (int i = 0; i < n; ++i) ids[fill[which[i]]++] = i;
  }
  // ---- device: upload the caller-order cloud, build curve order + boxes (build_tree.hip)
  const size_t m = (size_t)(n > 0 ? n : 1);
  HIPCHECK(c.rx.reserve(m)); HIPCHECK(c.ry.reserve(m)); HIPCHECK(c.rz.reserve(m)); HIPCHECK(c.rl.reserve(m));
  HIPCHECK(c.ids.reserve(m)); HIPCHECK(c.d_perm.reserve(m));
  // one segment: sort buffers of its size; several: every segment sorts its own range of buffers that hold all points (and the
  // segments' descriptions go to the device: the other stages are ONE launch over all segments)
  const bool together = want && n_seg > 1;
  const size_t sort_n = together ? m : (size_t)max_cnt;
  HIPCHECK(c.keys_in.reserve(sort_n)); HIPCHECK(c.keys_out.reserve(sort_n));
  HIPCHECK(c.vals_in.reserve(sort_n)); HIPCHECK(c.vals_out.reserve(sort_n));
  const size_t temp_bytes = sicp::build_sort_temp_bytes(max_cnt);
  HIPCHECK(c.sort_temp.reserve(temp_bytes + 256));
  if (together) {
    HIPCHECK(c.d_segs.reserve((size_t)n_seg)); HIPCHECK(c.d_seg_begin.reserve((size_t)n_seg)); HIPCHECK(c.d_seg_end.reserve((size_t)n_seg));
    HIPCHECK(c.h_segs.resize((size_t)n_seg)); HIPCHECK(c.h_seg_begin.resize((size_t)n_seg)); HIPCHECK(c.h_seg_end.resize((size_t)n_seg));
  }
  HIPCHECK(c.x.reserve(m)); HIPCHECK(c.y.reserve(m)); HIPCHECK(c.z.reserve(m));
  HIPCHECK(c.label.reserve(m)); HIPCHECK(c.inv.reserve(m));
  HIPCHECK(c.pts4.reserve((size_t)pt_total + 1)); HIPCHECK(c.box_lo.reserve((size_t)node_total + 1));
  HIPCHECK(c.box_hi.reserve((size_t)node_total + 1)); HIPCHECK(c.leaf_code.reserve((size_t)code_total + 1));
  auto up = [&](void* dst, const void* src, size_t bytes) {
    return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess;
  };
  HIPCHECK(up(c.rx.p, c.hx.data(), sizeof(float) * n));
  HIPCHECK(up(c.ry.p, c.hy.data(), sizeof(float) * n));
  HIPCHECK(up(c.rz.p, c.hz.data(), sizeof(float) * n));
  if (c.has_label) HIPCHECK(up(c.rl.p, c.hl.data(), sizeof(uint32_t) * n));
  if (want) HIPCHECK(up(c.ids.p, ids.data(), sizeof(int) * n));
  sicp::BuildBuffers b;
  b.rx = c.rx.p; b.ry = c.ry.p; b.rz = c.rz.p; b.rl = c.has_label ? c.rl.p : nullptr; b.ids = want ? c.ids.p : nullptr;
  b.keys_in = c.keys_in.p; b.keys_out = c.keys_out.p; b.vals_in = c.vals_in.p; b.vals_out = c.vals_out.p;
  b.sort_temp = c.sort_temp.p; b.sort_temp_bytes = temp_bytes;
  b.d_segs = together ? c.d_segs.p : nullptr; b.h_segs = together ? c.h_segs.data() : nullptr;
  b.d_seg_begin = together ? c.d_seg_begin.p : nullptr; b.d_seg_end = together ? c.d_seg_end.p : nullptr;
  b.h_seg_begin = together ? c.h_seg_begin.data() : nullptr; b.h_seg_end = together ? c.h_seg_end.data() : nullptr;
  b.x = c.x.p; b.y = c.y.p; b.z = c.z.p; b.label = c.label.p; b.perm = c.d_perm.p; b.inv = c.inv.p;
  b.pts4 = c.pts4.p; b.box_lo = c.box_lo.p; b.box_hi = c.box_hi.p; b.leaf_code = c.leaf_code.p;
  HIPCHECK(sicp::build_tree_device(b, segs.data(), n_seg, h->stream));
  HIPCHECK(c.perm.resize(n));  // device -> caller order, for returning results in the caller's order
  if (n > 0) HIPCHECK(hipMemcpyAsync(c.perm.data(), c.d_perm.p, sizeof(int) * n, hipMemcpyDeviceToHost, h->stream));
  // no synchronisation here: every staging buffer is pinned memory owned by the cloud, and the next
  // user of the cloud waits for ready_ev (cloud_wait).  A sequence driver can therefore queue the
  // uploads of a whole batch of scans back to back, beside the registrations of the previous batch.
  if (!c.ready_ev) HIPCHECK(hipEventCreateWithFlags(&c.ready_ev, hipEventDisableTiming));
  HIPCHECK(hipEventRecord(c.ready_ev, h->stream));
  c.pending = true;
  c.layout = want;
  c.feat_valid = false;
  c.cov_general = false;
  h->corr_valid = false;
  h->hint_ok = false;
  return SICP_OK;
}

// The buffers a cloud's features will need, taken from the arena when the cloud is SET rather than at its first
// align(): a new arena slab is a hipMalloc of up to 1 GB, which the driver clears before handing it out (~30 ms per
// GB) -- inside a stream that is the worker's turn, i.e. every registration in flight waits (measured: the resident
// leg of the open-stream bench took 1.28 instead of 0.45 s when its 1025 clouds' 27 GB of feature buffers were first
// touched inside the timed region).
int reserve_features(sicp_context* h, Cloud& c) {
  const sicp_params& P = h->params;
  const size_t m = (size_t)(c.n > 0 ? c.n : 1);
  HIPCHECK(c.rec.reserve(m));
  HIPCHECK(c.nn.reserve(m * (size_t)(P.k_cov > 0 ? P.k_cov : 1)));
  HIPCHECK(c.rec_dense.reserve(sicp::dense_rec_bytes(c.n)));
  if (P.mode == SICP_MODE_EM && P.num_classes > 0) {
    HIPCHECK(c.hist.reserve(m * (size_t)sicp::hist_stride(P.num_classes)));
    if (!weights_from_histograms(P, P.knn)) HIPCHECK(c.proj.reserve(m * (size_t)sicp::proj_stride(P.num_classes)));
  }
  return SICP_OK;
}

int set_cloud_common(sicp_handle h, int which, int32_t n, const StridedCloud& in) {
  SICPCHECK(set_device(h));
  if (h->cl[which].use_count() > 1) {  // shared with another handle: leave theirs alone
    settle_cloud(*h->cl[which]);
    h->cl[which] = acquire_cloud(h->device);
  }
  Cloud& c = h->cloud(which);
  SICPCHECK(stage_cloud(h, c, n, in));
  c.is_set = true;
  c.layout = -1;
  c.feat_valid = false;
  c.cov_general = false;
  h->corr_valid = false;
  h->hint_ok = false;
  // upload now for the current mode, so that align() starts with the cloud resident in HBM;
  // a later mode change re-lays it out lazily
  if (h->params.mode != SICP_MODE_SEMANTIC || c.has_label) {
    SICPCHECK(prepare_cloud(h, c));
    SICPCHECK(reserve_features(h, c));
  }
  return SICP_OK;
}

}  // namespace host
}  // namespace sicp
