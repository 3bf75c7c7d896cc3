// api_frame.hip -- frame-level and single-pair helpers: Hamming A/B entry points, place recognition, projectTo3D, point clouds, the environment measurement model, settings, packing
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include "rgbdfe_host.h"

namespace impl {

static int hamming_keys_to_host(rgbdfe_ctx* ctx, uint32_t nq, uint32_t planes, int32_t* out_hd, int32_t* out_idx) {
  const size_t mk = (size_t)ctx->cfg.max_keypoints;
  std::vector<uint32_t> all((size_t)planes * mk), keys(nq);
  hipStream_t st = ctx->lanes[0].stream;
  HIP_TRY(ctx, hipMemcpyAsync(all.data(), ctx->lanes[0].d_keys, all.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (uint32_t i = 0; i < nq; ++i) {
    uint32_t k = all[i];
    for (uint32_t pl = 1; pl < planes; ++pl) k = std::min(k, all[(size_t)pl * mk + i]);
    keys[i] = k;
  }
  for (uint32_t i = 0; i < nq; ++i) {
    const uint32_t hd = keys[i] >> 16;
    if (hd > 256u) {  // nothing searched: (257, -1), features.cpp:172-173
      out_hd[i] = 257;
      out_idx[i] = -1;
    } else {
      out_hd[i] = (int32_t)hd;
      out_idx[i] = (int32_t)(keys[i] & 0xFFFFu);
    }
  }
  return RGBDFE_OK;
}

int rgbdfe_hamming_nn_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* out_hd,
                            int32_t* out_idx) {
  if (!ctx || !out_hd || !out_idx) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  auto q = ctx->nodes.find(query_id);
  auto t = ctx->nodes.find(train_id);
  if (q == ctx->nodes.end() || t == ctx->nodes.end())
    return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "node not resident");
  if (q->second.kind != 0u || t->second.kind != 0u)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_hamming_nn_nodes needs ORB (binary descriptor) nodes");
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  for (auto& sl : ctx->ring) sl.pending = false;  // every lane is idle now
  rgbdfe_ctx::Slot& slot = ctx->ring[0];
  hipStream_t st = ctx->lanes[0].stream;
  PairWork& w = slot.h_work[0];
  w.q_slot = q->second.slot; w.t_slot = t->second.slot;
  w.nq = q->second.n; w.nt = t->second.n;
  w.uid = pair_uid(query_id, train_id); w.qid = query_id; w.tid = train_id; w.pad = 0;
  if (w.nq == 0) return RGBDFE_OK;
  HIP_TRY(ctx, hipMemcpyAsync(slot.d_work, slot.h_work, sizeof(PairWork), hipMemcpyHostToDevice, st));
  const uint32_t planes = launch_hamming(ctx, slot.d_work, ctx->lanes[0].d_keys, 1u, w.nq, w.nt, st);
  HIP_TRY(ctx, hipGetLastError());
  return hamming_keys_to_host(ctx, w.nq, planes, out_hd, out_idx);
}

// Loop-closure prefilter (place_recognition.hip): ranks the candidates of every query node by how many of the query's
// descriptors find one of their k nearest matches in them.  One Hamming launch + one vote launch for the whole batch.
int rgbdfe_place_recognition_batch(rgbdfe_ctx* ctx, const int32_t* query_ids, int32_t n_queries,
                                   const int32_t* candidate_offsets, const int32_t* candidate_ids, int32_t k_neighbours,
                                   int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                                   int32_t* out_counts) {
  if (!ctx || n_queries < 0 || k_neighbours < 1 || k_neighbours > 8 || max_hd < 1 || max_hd > 257 || max_out < 0 ||
      (n_queries > 0 && (!query_ids || !candidate_offsets || !out_counts)) ||
      (n_queries > 0 && max_out > 0 && (!out_ids || !out_scores)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad place recognition arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  for (int32_t s = 0; s < n_queries; ++s) out_counts[s] = 0;
  if (n_queries == 0) return RGBDFE_OK;
  const int32_t total = candidate_offsets[n_queries];
  if (candidate_offsets[0] != 0 || total < 0 || (total > 0 && !candidate_ids))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "candidate_offsets must start at 0 and ascend");
  if (total > ctx->cfg.max_pairs_per_batch)
    return fail(ctx, RGBDFE_ERR_CAPACITY, "more (query, candidate) pairs than max_pairs_per_batch");
  // the whole offsets array is checked before anything indexed by it is written (h_work, rows); queries are the y
  // extent of the vote grid
  if (n_queries > 65535) return fail(ctx, RGBDFE_ERR_CAPACITY, "at most 65535 queries per place recognition batch");
  for (int32_t s = 0; s < n_queries; ++s) {
    const int32_t c0 = candidate_offsets[s], c1 = candidate_offsets[s + 1];
    if (c0 < 0 || c1 < c0 || c1 > total || c1 - c0 > 65535)
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "candidate_offsets must ascend within [0, offsets[n_queries]] (<= 65535 candidates per query)");
  }
  if (total == 0 || max_out == 0) return RGBDFE_OK;
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  for (auto& sl : ctx->ring) sl.pending = false;  // every lane is idle now
  rgbdfe_ctx::Slot& slot = ctx->ring[0];
  rgbdfe_ctx::Lane& lane = ctx->lanes[0];
  hipStream_t st = lane.stream;
  std::vector<uint32_t> rows((size_t)total), seg((size_t)n_queries + 1);
  uint32_t max_nt = 0, max_nq = 0;
  for (int32_t s = 0; s < n_queries; ++s) {
    const int32_t c0 = candidate_offsets[s], c1 = candidate_offsets[s + 1];
    if (c1 < c0 || c1 - c0 > 65535) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "candidate_offsets must ascend (<= 65535 candidates per query)");
    seg[(size_t)s] = (uint32_t)c0;
    auto q = ctx->nodes.find(query_ids[s]);
    if (q == ctx->nodes.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "query node not resident");
    if (q->second.kind != 0u) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "place recognition needs ORB nodes");
    if (c1 > c0) max_nq = std::max(max_nq, q->second.n);
    for (int32_t i = c0; i < c1; ++i) {
      auto t = ctx->nodes.find(candidate_ids[i]);
      if (t == ctx->nodes.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "candidate node not resident");
      if (t->second.kind != 0u) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "place recognition needs ORB nodes");
      PairWork& w = slot.h_work[i];
      w.q_slot = q->second.slot; w.t_slot = t->second.slot;
      w.nq = q->second.n; w.nt = t->second.n;
      w.uid = 0; w.qid = query_ids[s]; w.tid = candidate_ids[i]; w.pad = 0;
      rows[(size_t)i] = t->second.n;
      max_nt = std::max(max_nt, t->second.n);
    }
  }
  seg[(size_t)n_queries] = (uint32_t)total;
  std::vector<uint32_t> votes((size_t)total, 0u);
  if (max_nq > 0 && max_nt > 0) {
    const size_t b_votes = ((size_t)total * 4 + 255) & ~(size_t)255;
    int rc = ensure_scratch(ctx, b_votes + ((size_t)n_queries + 1) * 4 + 256);
    if (rc != RGBDFE_OK) return rc;
    uint32_t* d_votes = (uint32_t*)ctx->d_scratch;
    uint32_t* d_seg = (uint32_t*)((char*)ctx->d_scratch + b_votes);
    HIP_TRY(ctx, hipMemcpyAsync(slot.d_work, slot.h_work, sizeof(PairWork) * (size_t)total, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_seg, seg.data(), ((size_t)n_queries + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(d_votes, 0, (size_t)total * 4, st));
    const uint32_t planes = launch_hamming(ctx, slot.d_work, lane.d_keys, (uint32_t)total, max_nq, max_nt, st);
    launch_place_votes(lane.d_keys, planes, (uint32_t)ctx->cfg.max_keypoints, slot.d_work, d_seg, (uint32_t)n_queries, max_nq,
                       (uint32_t)k_neighbours, (uint32_t)max_hd, d_votes, st);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(votes.data(), d_votes, (size_t)total * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  // score = votes / descriptor count of the candidate (loop_closing.cpp:263); rank by score, ties: listed first
  std::vector<int32_t> order;
  std::vector<float> score;
  for (int32_t s = 0; s < n_queries; ++s) {
    const int32_t c0 = candidate_offsets[s], c1 = candidate_offsets[s + 1];
    order.clear();
    score.assign((size_t)(c1 - c0), 0.f);
    for (int32_t i = c0; i < c1; ++i) {
      if (votes[(size_t)i] == 0u) continue;  // nodes nobody voted for are not in the reference's score map either (:243-248)
      score[(size_t)(i - c0)] = (float)votes[(size_t)i] / (float)rows[(size_t)i];
      order.push_back(i - c0);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return score[(size_t)a] > score[(size_t)b]; });
    const int32_t n = std::min<int32_t>((int32_t)order.size(), max_out);
    for (int32_t i = 0; i < n; ++i) {
      out_ids[(size_t)s * max_out + i] = candidate_ids[c0 + order[(size_t)i]];
      out_scores[(size_t)s * max_out + i] = score[(size_t)order[(size_t)i]];
    }
    out_counts[s] = n;
  }
  return RGBDFE_OK;
}

int rgbdfe_place_recognition(rgbdfe_ctx* ctx, int32_t query_id, const int32_t* candidate_ids, int32_t n_candidates,
                             int32_t k_neighbours, int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                             int32_t* n_out) {
  if (!ctx || !n_out || n_candidates < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad place recognition arguments");
  const int32_t offs[2] = {0, n_candidates};
  *n_out = 0;
  return impl::rgbdfe_place_recognition_batch(ctx, &query_id, 1, offs, candidate_ids, k_neighbours, max_hd, max_out, out_ids,
                                              out_scores, n_out);
}

int rgbdfe_hamming_nn_host(rgbdfe_ctx* ctx, const uint8_t* qdesc, int32_t nq, const uint8_t* tdesc,
                           int32_t nt, int32_t* out_hd, int32_t* out_idx) {
  if (!ctx || nq < 0 || nt < 0 || !out_hd || !out_idx) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  // two temporary nodes with ids outside the int32 range a SLAM graph uses
  const int32_t qid = INT32_MIN + 1, tid = INT32_MIN + 2;
  std::vector<float> zq((size_t)(nq > 0 ? nq : 1) * 4, 0.f), zt((size_t)(nt > 0 ? nt : 1) * 4, 0.f);
  int rc = impl::rgbdfe_upload_node(ctx, qid, qdesc, zq.data(), nq);
  if (rc != RGBDFE_OK) return rc;
  rc = impl::rgbdfe_upload_node(ctx, tid, tdesc, zt.data(), nt);
  if (rc == RGBDFE_OK) rc = impl::rgbdfe_hamming_nn_nodes(ctx, qid, tid, out_hd, out_idx);
  (void)impl::rgbdfe_release_node(ctx, qid);
  (void)impl::rgbdfe_release_node(ctx, tid);
  return rc;
}

int rgbdfe_project_to_3d(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* depth,
                         int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                         double depth_scaling, int32_t max_keypoints, int32_t* kept_idx,
                         float* xyz1, int32_t* n_out) {
  if (!ctx || n_kp < 0 || rows < 1 || cols < 1 || !depth || !kept_idx || !xyz1 || !n_out ||
      max_keypoints < 0 || (n_kp > 0 && !kp_xy))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  *n_out = 0;
  if (n_kp == 0 || max_keypoints == 0) return RGBDFE_OK;
  const size_t b_kp = ((size_t)n_kp * 8 + 255) & ~(size_t)255;
  const size_t b_depth = ((size_t)rows * cols * 4 + 255) & ~(size_t)255;
  const size_t b_idx = ((size_t)n_kp * 4 + 255) & ~(size_t)255;
  const size_t b_xyz = ((size_t)n_kp * 16 + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, b_kp + b_depth + b_idx + b_xyz + 256);
  if (rc != RGBDFE_OK) return rc;
  char* base = (char*)ctx->d_scratch;
  float* d_kp = (float*)base;
  float* d_depth = (float*)(base + b_kp);
  int32_t* d_idx = (int32_t*)(base + b_kp + b_depth);
  float4* d_xyz = (float4*)(base + b_kp + b_depth + b_idx);
  int32_t* d_n = (int32_t*)(base + b_kp + b_depth + b_idx + b_xyz);
  HIP_TRY(ctx, hipMemcpyAsync(d_kp, kp_xy, (size_t)n_kp * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_depth, depth, (size_t)rows * cols * 4, hipMemcpyHostToDevice, ctx->stream));
  // node.cpp:913-916: fxinv = float(1./fx) etc.
  launch_project_to_3d(d_kp, n_kp, d_depth, rows, cols, (float)(1. / fx), (float)(1. / fy), (float)cx,
                       (float)cy, depth_scaling, max_keypoints, d_idx, d_xyz, d_n, ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(kept_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(xyz1, d_xyz, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *n_out = n;
  return RGBDFE_OK;
}

int rgbdfe_set_feature_min_depth(rgbdfe_ctx* ctx, int32_t on) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->feature_min_depth = on != 0;
  return RGBDFE_OK;
}

// removeDepthless / projectTo3D with "use_feature_min_depth" on (node.cpp:82, :940): the keypoint's depth is
// getMinDepthInNeighborhood(depth, pt, size) (misc.cpp:774-793).  kp_size = cv::KeyPoint::size.
int rgbdfe_project_to_3d_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp,
                                   const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                   double cy, double depth_scaling, int32_t max_keypoints, int32_t* kept_idx,
                                   float* xyz1, int32_t* n_out) {
  if (!ctx || n_kp < 0 || rows < 1 || cols < 1 || !depth || !kept_idx || !xyz1 || !n_out || max_keypoints < 0 ||
      (n_kp > 0 && (!kp_xy || !kp_size)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  *n_out = 0;
  if (n_kp == 0 || max_keypoints == 0) return RGBDFE_OK;
  const size_t b_kp = ((size_t)n_kp * 12 + 255) & ~(size_t)255;   // (x, y, size) for the neighbourhood kernel
  const size_t b_xyz_in = ((size_t)n_kp * 12 + 255) & ~(size_t)255;  // x, y pairs followed by the n depths
  const size_t b_depth = ((size_t)rows * cols * 4 + 255) & ~(size_t)255;
  const size_t b_idx = ((size_t)n_kp * 4 + 255) & ~(size_t)255;
  const size_t b_xyz = ((size_t)n_kp * 16 + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, b_kp + b_xyz_in + b_depth + b_idx + b_xyz + 256);
  if (rc != RGBDFE_OK) return rc;
  char* base = (char*)ctx->d_scratch;
  float* d_kp3 = (float*)base;
  float* d_in = (float*)(base + b_kp);
  float* d_depth = (float*)(base + b_kp + b_xyz_in);
  int32_t* d_idx = (int32_t*)(base + b_kp + b_xyz_in + b_depth);
  float4* d_xyz = (float4*)(base + b_kp + b_xyz_in + b_depth + b_idx);
  int32_t* d_n = (int32_t*)(base + b_kp + b_xyz_in + b_depth + b_idx + b_xyz);
  std::vector<float> h3((size_t)n_kp * 3);
  for (int32_t i = 0; i < n_kp; ++i) { h3[3 * i] = kp_xy[2 * i]; h3[3 * i + 1] = kp_xy[2 * i + 1]; h3[3 * i + 2] = kp_size[i]; }
  HIP_TRY(ctx, hipMemcpyAsync(d_kp3, h3.data(), h3.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_in, kp_xy, (size_t)n_kp * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_depth, depth, (size_t)rows * cols * 4, hipMemcpyHostToDevice, ctx->stream));
  launch_min_depth(d_kp3, n_kp, d_depth, rows, cols, d_in + (size_t)2 * n_kp, ctx->stream);
  // projectTo3D proper, with the looked-up depths (its own gather is bypassed): node.cpp:913-916 for the intrinsics
  launch_project_to_3d(d_in, n_kp, nullptr, rows, cols, (float)(1. / fx), (float)(1. / fy), (float)cx, (float)cy,
                       depth_scaling, max_keypoints, d_idx, d_xyz, d_n, ctx->stream, false, d_in + (size_t)2 * n_kp);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(kept_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(xyz1, d_xyz, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *n_out = n;
  return RGBDFE_OK;
}

// Node::projectTo3D, point-cloud overload (node.cpp:855-898).  The organised cloud stays on the host: the point under
// every keypoint, point_cloud->at((int)x, (int)y), is gathered here (16 bytes per keypoint cross PCIe instead of the
// whole cloud); filter, compaction and the max_keypoints cut run on the device.
static int project_cloud_common(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* cloud, int32_t rows,
                                int32_t cols, double maximum_depth, int32_t max_keypoints, int32_t* kept_idx, float* xyz1,
                                int32_t* n_out) {
  *n_out = 0;
  if (n_kp == 0 || max_keypoints == 0) return RGBDFE_OK;
  const size_t b_kp = ((size_t)n_kp * 8 + 255) & ~(size_t)255;
  const size_t b_pts = ((size_t)n_kp * 16 + 255) & ~(size_t)255;
  const size_t b_idx = ((size_t)n_kp * 4 + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, b_kp + 2 * b_pts + b_idx + 256);
  if (rc != RGBDFE_OK) return rc;
  char* base = (char*)ctx->d_scratch;
  float* d_kp = (float*)base;
  float4* d_pts = (float4*)(base + b_kp);
  int32_t* d_idx = (int32_t*)(base + b_kp + b_pts);
  float4* d_xyz = (float4*)(base + b_kp + b_pts + b_idx);
  int32_t* d_n = (int32_t*)(base + b_kp + 2 * b_pts + b_idx);
  std::vector<float> pts((size_t)n_kp * 4, 0.f);
  for (int32_t i = 0; i < n_kp; ++i) {
    const float x = kp_xy[2 * i], y = kp_xy[2 * i + 1];
    if (x >= (float)cols || x < 0.f || y >= (float)rows || y < 0.f || std::isnan(x) || std::isnan(y)) continue;
    memcpy(&pts[(size_t)i * 4], cloud + 4 * ((size_t)(int)y * (size_t)cols + (size_t)(int)x), 16);  // :877
  }
  HIP_TRY(ctx, hipMemcpyAsync(d_kp, kp_xy, (size_t)n_kp * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_pts, pts.data(), (size_t)n_kp * 16, hipMemcpyHostToDevice, ctx->stream));
  launch_project_cloud(d_kp, n_kp, d_pts, true, rows, cols, maximum_depth, max_keypoints, d_idx, d_xyz, d_n, ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(kept_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(xyz1, d_xyz, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *n_out = n;
  return RGBDFE_OK;
}

int rgbdfe_project_to_3d_cloud(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* cloud, int32_t rows,
                               int32_t cols, double maximum_depth, int32_t max_keypoints, int32_t* kept_idx, float* xyz1,
                               int32_t* n_out) {
  if (!ctx || n_kp < 0 || rows < 1 || cols < 1 || !cloud || !kept_idx || !xyz1 || !n_out || max_keypoints < 0 ||
      (n_kp > 0 && !kp_xy))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  return project_cloud_common(ctx, kp_xy, n_kp, cloud, rows, cols, maximum_depth, max_keypoints, kept_idx, xyz1, n_out);
}

// The feature path of the Node constructor that is handed the sensor's organised point cloud (node.cpp:252-369):
// detector->detect (:293) -> projectTo3D(cloud) (:308, with the maximum_depth test and the max_keypoints cut) ->
// extractor->compute (:311).  No removeDepthless, no retainBest on this path.
// Deviation D6: cv::ORB::compute drops keypoints within 31 px of the border and regroups the rest by octave, which in
// the reference leaves feature_locations_3d_ (filled BEFORE compute) out of step with the keypoints and descriptors
// (its assert at :318 fires in a debug build); here the 3-D points follow their keypoints.
int rgbdfe_detect_describe_cloud(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* cloud,
                                 int32_t rows, int32_t cols, double maximum_depth, rgbdfe_keypoint* keypoints,
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out) {
  if (!ctx || !gray || !cloud || rows < 1 || cols < 1 || !keypoints || !descriptors || !xyz1 || !n_out)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  ensure_detector(ctx);
  OrbWorkspace& orb = ctx->orb;
  const int max_kp = ctx->orb_max_keypoints;
  std::string err;
  int rc = orb.prepare(cols, rows, true, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  orb.cell_mask_nonzero.assign((size_t)orb.n_cells, mask ? 0 : 1);
  if (mask)
    for (int c = 0; c < orb.n_cells; ++c) {
      const OrbWorkspace::Cell& ce = orb.cells[c];
      char nz = 0;
      for (int y = 0; y < ce.h && !nz; ++y) {
        const uint8_t* r = mask + (size_t)(ce.y0 + y) * cols + ce.x0;
        for (int x = 0; x < ce.w; ++x)
          if (r[x]) { nz = 1; break; }
      }
      orb.cell_mask_nonzero[c] = nz;
    }
  rc = orb.upload_and_build(gray, mask, ctx->stream, err);
  std::vector<KpOut> kps;
  if (rc == RGBDFE_OK) rc = orb.grid_detect(kps, ctx->stream, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  *n_out = 0;
  const int n_det = (int)kps.size();
  std::vector<float> xy((size_t)n_det * 2), pxyz((size_t)n_det * 4);
  std::vector<int32_t> kept((size_t)std::max(n_det, 1));
  for (int i = 0; i < n_det; ++i) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
  int32_t n3 = 0;
  rc = project_cloud_common(ctx, xy.data(), n_det, cloud, rows, cols, maximum_depth, max_kp, kept.data(), pxyz.data(), &n3);
  if (rc != RGBDFE_OK) return rc;
  std::vector<KpOut> k3((size_t)n3);
  for (int i = 0; i < n3; ++i) k3[i] = kps[(size_t)kept[i]];  // feature_locations_2d after the erase / resize (:874-895)
  std::vector<uint8_t> desc;
  std::vector<int> order;
  rc = orb.compute(k3, desc, ctx->stream, err, nullptr, &order);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  const int n = (int)k3.size();
  for (int i = 0; i < n; ++i) memcpy(xyz1 + 4 * (size_t)i, &pxyz[(size_t)order[i] * 4], 16);
  kp_to_abi(k3, keypoints);
  if (!desc.empty()) memcpy(descriptors, desc.data(), desc.size());
  *n_out = n;
  return RGBDFE_OK;
}

// kp_size != nullptr: "use_feature_min_depth" (node.cpp:727-731) -- the depth of a keypoint is
// getMinDepthInNeighborhood(depth, pt, size) (misc.cpp:774-793) instead of the pixel under it
int rgbdfe_sift_node_features(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp, const float* desc_in,
                              const float* depth, int32_t rows, int32_t cols, double fx, double fy,
                              double cx, double cy, double depth_scaling, int32_t max_keypoints,
                              int32_t use_root_sift, int32_t* kept_idx, float* xyz1,
                              float* siftgpu_descriptors, float* feature_descriptors, int32_t* n_out) {
  if (!ctx || n_kp < 0 || rows < 1 || cols < 1 || !depth || !kept_idx || !xyz1 || !siftgpu_descriptors ||
      !n_out || max_keypoints < 0 || (n_kp > 0 && (!kp_xy || !desc_in)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  *n_out = 0;
  if (n_kp == 0 || max_keypoints == 0) return RGBDFE_OK;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int cap = n_kp < max_keypoints ? n_kp : max_keypoints;
  const size_t b_kp = up((size_t)n_kp * 8), b_depth = up((size_t)rows * cols * 4), b_idx = up((size_t)n_kp * 4);
  const size_t b_xyz = up((size_t)n_kp * 16), b_in = up((size_t)n_kp * 512), b_out = up((size_t)cap * 512);
  const size_t b_kp3 = kp_size ? up((size_t)n_kp * 12) : 0, b_z = kp_size ? up((size_t)n_kp * 4) : 0;
  int rc = ensure_scratch(ctx, b_kp + b_depth + b_idx + b_xyz + b_in + 2 * b_out + b_kp3 + b_z + 256);
  if (rc != RGBDFE_OK) return rc;
  char* p = (char*)ctx->d_scratch;
  float* d_kp3 = (float*)p;         p += b_kp3;
  float* d_z = (float*)p;           p += b_z;
  float* d_kp = (float*)p;          p += b_kp;
  float* d_depth = (float*)p;       p += b_depth;
  int32_t* d_idx = (int32_t*)p;     p += b_idx;
  float4* d_xyz = (float4*)p;       p += b_xyz;
  float* d_in = (float*)p;          p += b_in;
  float* d_raw = (float*)p;         p += b_out;
  float* d_feat = (float*)p;        p += b_out;
  int32_t* d_n = (int32_t*)p;
  HIP_TRY(ctx, hipMemcpyAsync(d_kp, kp_xy, (size_t)n_kp * 8, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_depth, depth, (size_t)rows * cols * 4, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_in, desc_in, (size_t)n_kp * 512, hipMemcpyHostToDevice, ctx->stream));
  std::vector<float> h3;
  if (kp_size) {
    h3.resize((size_t)n_kp * 3);
    for (int32_t i = 0; i < n_kp; ++i) { h3[3 * i] = kp_xy[2 * i]; h3[3 * i + 1] = kp_xy[2 * i + 1]; h3[3 * i + 2] = kp_size[i]; }
    HIP_TRY(ctx, hipMemcpyAsync(d_kp3, h3.data(), h3.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_min_depth(d_kp3, n_kp, d_depth, rows, cols, d_z, ctx->stream);
  }
  launch_project_to_3d(d_kp, n_kp, d_depth, rows, cols, (float)(1. / fx), (float)(1. / fy), (float)cx,
                       (float)cy, depth_scaling, max_keypoints, d_idx, d_xyz, d_n, ctx->stream, true, kp_size ? d_z : nullptr);
  launch_sift_pack(d_in, d_idx, d_n, cap, use_root_sift != 0, d_raw, feature_descriptors ? d_feat : nullptr,
                   ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(kept_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(xyz1, d_xyz, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(siftgpu_descriptors, d_raw, (size_t)n * 512, hipMemcpyDeviceToHost, ctx->stream));
    if (feature_descriptors)
      HIP_TRY(ctx, hipMemcpyAsync(feature_descriptors, d_feat, (size_t)n * 512, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *n_out = n;
  return RGBDFE_OK;
}

int rgbdfe_depth_to_mono8(rgbdfe_ctx* ctx, const void* depth, int32_t depth_is_u16, int32_t rows, int32_t cols,
                          uint8_t* mono8, float* depth_m) {
  if (!ctx || !depth || !mono8 || rows < 1 || cols < 1 || (depth_is_u16 && !depth_m))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  const size_t n = (size_t)rows * (size_t)cols;
  const size_t b_in = ((n * (depth_is_u16 ? 2 : 4)) + 255) & ~(size_t)255;
  const size_t b_m8 = (n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, b_in + b_m8 + n * 4 + 256);
  if (rc != RGBDFE_OK) return rc;
  char* p = (char*)ctx->d_scratch;
  void* d_in = p;
  uint8_t* d_m8 = (uint8_t*)(p + b_in);
  float* d_f = (float*)(p + b_in + b_m8);
  HIP_TRY(ctx, hipMemcpyAsync(d_in, depth, n * (depth_is_u16 ? 2 : 4), hipMemcpyHostToDevice, ctx->stream));
  if (depth_is_u16) launch_depth_u16((const uint16_t*)d_in, n, d_m8, d_f, ctx->stream);
  else launch_depth_to_mono8_f32((const float*)d_in, n, d_m8, ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(mono8, d_m8, n, hipMemcpyDeviceToHost, ctx->stream));
  if (depth_is_u16) HIP_TRY(ctx, hipMemcpyAsync(depth_m, d_f, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RGBDFE_OK;
}

int rgbdfe_upload_node_cloud(rgbdfe_ctx* ctx, int32_t node_id, const float* depth, int32_t rows, int32_t cols,
                             const uint8_t* rgb, int32_t rgb_channels, int32_t encoding_bgr, double fx,
                             double fy, double cx, double cy, double depth_scaling, double min_depth,
                             int32_t cloud_skip, float* cloud_out) {
  if (!ctx || !depth || rows < 1 || cols < 1 || cloud_skip < 1 || (rgb && rgb_channels != 1 && rgb_channels != 3))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  if (rows % cloud_skip != 0 || cols % cloud_skip != 0)  // misc.cpp:479-481: "will most likely crash"
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "cloud_creation_skip_step must divide the image dimensions");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  const int ch = rows / cloud_skip, cw = cols / cloud_skip;
  const size_t n = (size_t)rows * (size_t)cols;
  const size_t b_depth = (n * 4 + 255) & ~(size_t)255;
  const size_t b_rgb = rgb ? ((n * (size_t)rgb_channels + 255) & ~(size_t)255) : 0;
  int rc = ensure_scratch(ctx, b_depth + b_rgb + 256);
  if (rc != RGBDFE_OK) return rc;
  float* d_depth = (float*)ctx->d_scratch;
  uint8_t* d_rgb = rgb ? (uint8_t*)ctx->d_scratch + b_depth : nullptr;
  CloudEntry& ce = ctx->clouds[node_id];
  if (ce.d && (ce.ch != ch || ce.cw != cw)) {
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
    (void)hipFree(ce.d);
    ce.d = nullptr;
  }
  if (!ce.d) {
    if (hipMalloc((void**)&ce.d, (size_t)ch * cw * (sizeof(float4) + sizeof(float))) != hipSuccess) {
      ctx->clouds.erase(node_id);
      return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "cloud allocation failed");
    }
  }
  ce.ch = ch; ce.cw = cw; ce.cloud_skip = cloud_skip;
  ce.samples_skip = 0;  // the cached sample array belongs to the previous depth image
  ce.fx = (float)fx; ce.fy = (float)fy; ce.cx = (float)cx; ce.cy = (float)cy;  // misc.cpp:59-62
  HIP_TRY(ctx, hipMemcpyAsync(d_depth, depth, n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (rgb) HIP_TRY(ctx, hipMemcpyAsync(d_rgb, rgb, n * (size_t)rgb_channels, hipMemcpyHostToDevice, ctx->stream));
  // getCameraIntrinsicsInverseFocalLength (misc.cpp:64-69): 1./float(fx) assigned to float
  const float fxinv = (float)(1. / ce.fx), fyinv = (float)(1. / ce.fy);
  launch_create_cloud(d_depth, rows, cols, d_rgb, rgb ? rgb_channels : 1, encoding_bgr, fxinv, fyinv, ce.cx, ce.cy,
                      depth_scaling, (float)min_depth, cloud_skip, ch, cw, ce.d,
                      reinterpret_cast<float*>(ce.d + (size_t)ch * cw), ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  if (cloud_out)
    HIP_TRY(ctx, hipMemcpyAsync(cloud_out, ce.d, (size_t)ch * cw * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RGBDFE_OK;
}

int rgbdfe_release_node_cloud(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = ctx->clouds.find(node_id);
  if (it == ctx->clouds.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "no cloud for this node");
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (it->second.d) (void)hipFree(it->second.d);
  if (it->second.d_samples) (void)hipFree(it->second.d_samples);
  ctx->clouds.erase(it);
  return RGBDFE_OK;
}

int rgbdfe_observation_likelihood(rgbdfe_ctx* ctx, int32_t n, const int32_t* new_ids, const int32_t* old_ids,
                                  const float* transforms, int32_t emm_skip_step, rgbdfe_emm_counts* out) {
  if (!ctx || n < 0 || (n > 0 && (!new_ids || !old_ids || !transforms || !out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  if (n == 0) return RGBDFE_OK;
  if (emm_skip_step <= 0) {  // misc.cpp:829-832 (skip_step < 0; 0 would not terminate in the reference)
    for (int32_t i = 0; i < n; ++i) { out[i].inliers = out[i].all = 1; out[i].outliers = out[i].occluded = 0; }
    return RGBDFE_OK;
  }
  std::vector<EmmJob> jobs((size_t)n);
  int ch = 0, cw = 0, cloud_skip = 1;
  for (int32_t i = 0; i < n; ++i) {
    auto a = ctx->clouds.find(new_ids[i]);
    auto b = ctx->clouds.find(old_ids[i]);
    if (a == ctx->clouds.end() || b == ctx->clouds.end())
      return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "observation likelihood needs the clouds of both nodes");
    const CloudEntry& cn = a->second;
    const CloudEntry& co = b->second;
    if (i == 0) { ch = co.ch; cw = co.cw; cloud_skip = co.cloud_skip; }
    if (cn.ch != ch || cn.cw != cw || co.ch != ch || co.cw != cw || co.cloud_skip != cloud_skip)
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "clouds of one batch must share their dimensions");  // misc.cpp:845
    if (cn.samples_skip != emm_skip_step) {  // (re)build this node's dense sample array for this skip step
      CloudEntry& w = a->second;
      if (w.d_samples) { (void)hipFree(w.d_samples); w.d_samples = nullptr; }
      const size_t ns = (size_t)((ch + emm_skip_step - 1) / emm_skip_step) * (size_t)((cw + emm_skip_step - 1) / emm_skip_step);
      if (hipMalloc((void**)&w.d_samples, ns * sizeof(float4)) != hipSuccess)
        return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "sample array allocation failed");
      launch_decimate_cloud(w.d, ch, cw, emm_skip_step, w.d_samples, ctx->stream);
      w.samples_skip = emm_skip_step;
    }
    EmmJob& jb = jobs[(size_t)i];
    jb.new_samples = cn.d_samples;
    jb.old_z = reinterpret_cast<const float*>(co.d + (size_t)co.ch * co.cw);
    const float* T = transforms + (size_t)i * 16;  // column-major like rgbdfe_match_result.trafo
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) jb.T[r * 4 + c] = T[c * 4 + r];
    jb.fx = co.fx / cloud_skip; jb.fy = co.fy / cloud_skip;  // misc.cpp:868-871
    jb.cx = co.cx / cloud_skip; jb.cy = co.cy / cloud_skip;
  }
  if (ch <= 1 || cw <= 1) {  // misc.cpp:834-843: unstructured cloud
    for (int32_t i = 0; i < n; ++i) { out[i].inliers = out[i].all = 1; out[i].outliers = out[i].occluded = 0; }
    return RGBDFE_OK;
  }
  const size_t b_jobs = (sizeof(EmmJob) * (size_t)n + 255) & ~(size_t)255;
  int rc = ensure_scratch(ctx, b_jobs + (size_t)n * 16 + 256);
  if (rc != RGBDFE_OK) return rc;
  EmmJob* d_jobs = (EmmJob*)ctx->d_scratch;
  uint32_t* d_counts = (uint32_t*)((char*)ctx->d_scratch + b_jobs);
  HIP_TRY(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(EmmJob) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  // cdf(x, mu, sigma) = 0.5 * (1 + erf((x - mu) / (sigma * SQRT_2))), sigma = sqrt(old_sigma + new_sigma),
  // both = cloud_creation_skip_step * depth_covariance() (misc.cpp:809-812, 914-922; a18: frozen value)
  const double s1 = cloud_skip * ctx->cfg.params.depth_cov;
  const double denom = std::sqrt(s1 + s1) * 1.41421;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profiling) {
    e0 = get_event(ctx); e1 = get_event(ctx);
    (void)hipEventRecord(e0, ctx->stream);
  }
  if (!(denom > 0.0) || !std::isfinite(denom))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "depth_cov must be positive and finite for the measurement model");
  const double d_lo = division_boundary(ctx->emm_q_lo, denom), d_hi = division_boundary(ctx->emm_q_hi, denom);
  launch_emm(d_jobs, n, ch, cw, emm_skip_step, d_lo, d_hi, d_counts, ctx->stream);
  if (ctx->profiling) (void)hipEventRecord(e1, ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d_counts, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->profiling) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
      ctx->k_ms[RGBDFE_KERNEL_EMM] += ms;
      ctx->k_launches[RGBDFE_KERNEL_EMM]++;
      ctx->k_pairs[RGBDFE_KERNEL_EMM] += n;
    }
    ctx->event_pool.push_back(e0);
    ctx->event_pool.push_back(e1);
  }
  return RGBDFE_OK;
}

int rgbdfe_observation_criterion_met(uint32_t inliers, uint32_t outliers, uint32_t all, double observability_threshold,
                                     double* quality) {
  // misc.cpp:1136-1148
  if (observability_threshold < 0) return 1;
  const double q = inliers / static_cast<double>(inliers + outliers);
  if (quality) *quality = q;
  const double certainty = inliers / static_cast<double>(all);
  return (q > observability_threshold) && (certainty > 0.25) ? 1 : 0;
}

int rgbdfe_set_latency_mode(rgbdfe_ctx* ctx, int32_t max_pairs, int32_t chunk_iterations) {
  if (!ctx || max_pairs < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->latency_pairs = max_pairs;
  ctx->latency_chunk_iters = chunk_iterations;
  return RGBDFE_OK;
}

int rgbdfe_set_hamming_mode(rgbdfe_ctx* ctx, int32_t mode) {
  if (!ctx || mode < 0 || mode > 3) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "hamming mode must be 0, 1, 2 or 3");
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->hamming_mode = mode;
  return RGBDFE_OK;
}

int rgbdfe_set_profiling(rgbdfe_ctx* ctx, int enable) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->profiling = enable != 0;
  return RGBDFE_OK;
}

int rgbdfe_get_kernel_time(rgbdfe_ctx* ctx, int which, double* total_ms, int64_t* launches,
                           int64_t* pairs) {
  if (!ctx || which < 0 || which >= RGBDFE_KERNEL_COUNT) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  drain_pending(ctx);  // synchronises the stream the kernels ran on
  if (total_ms) *total_ms = ctx->k_ms[which];
  if (launches) *launches = ctx->k_launches[which];
  if (pairs) *pairs = ctx->k_pairs[which];
  return RGBDFE_OK;
}

int rgbdfe_reset_kernel_time(rgbdfe_ctx* ctx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  for (int i = 0; i < RGBDFE_KERNEL_COUNT; ++i) {
    ctx->k_ms[i] = 0;
    ctx->k_launches[i] = 0;
    ctx->k_pairs[i] = 0;
  }
  return RGBDFE_OK;
}

int rgbdfe_sizeof_match_result(void) { return (int)sizeof(rgbdfe_match_result); }
int rgbdfe_sizeof_compact_result(void) { return (int)sizeof(rgbdfe_compact_result); }
int rgbdfe_pack_compact(rgbdfe_ctx* ctx, const void* d_records, int32_t n, void* d_compact, void* stream) {
  if (!ctx || n < 0 || (n > 0 && (!d_records || !d_compact))) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad pack arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  launch_compact_pack((const rgbdfe_match_result*)d_records, (uint32_t)n, (rgbdfe_compact_result*)d_compact,
                      stream ? (hipStream_t)stream : ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  return RGBDFE_OK;
}
int rgbdfe_set_graph_capture(rgbdfe_ctx* ctx, int enable) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->use_graphs = enable != 0;
  return RGBDFE_OK;
}

int rgbdfe_pack_inliers(rgbdfe_ctx* ctx, const void* d_records, int32_t n, int32_t n_headers, void* d_stream, int32_t* d_total,
                        void* stream) {
  if (!ctx || n < 0 || n_headers < n || !d_stream || !d_total || (n > 0 && !d_records))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad pack arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  launch_pack_inliers((const rgbdfe_match_result*)d_records, (uint32_t)n, (uint32_t)n_headers, d_stream, d_total,
                      stream ? (hipStream_t)stream : ctx->stream);
  HIP_TRY(ctx, hipGetLastError());
  return RGBDFE_OK;
}
int rgbdfe_sizeof_inlier_header(void) { return (int)sizeof(rgbdfe_inlier_header); }
int rgbdfe_graph_stats(rgbdfe_ctx* ctx, int64_t* out, int32_t n_out) {
  if (!ctx || !out || n_out < 0) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  const int64_t v[RGBDFE_GRAPH_STATS] = {ctx->graph_captures, ctx->graph_launches, ctx->graph_misses, ctx->graph_plain_batches,
                                         (int64_t)ctx->graph_capture_failures, ctx->graph_launch_failures,
                                         (int64_t)ctx->graphs.size(), ctx->use_graphs ? 1 : 0};
  for (int32_t i = 0; i < n_out && i < RGBDFE_GRAPH_STATS; ++i) out[i] += v[i];
  return RGBDFE_OK;
}

int rgbdfe_abi_version(void) { return 6; }  // 6: rgbdfe_wait_host_into, rgbdfe_gather_exchanges, the inlier all-gather's stride is a capacity (>= the longest list); 5: rgbdfe_submit_pair_list_host / rgbdfe_wait_host (the refinement kernel's round-4 debug hooks are gone); 2: multi-device handles, rgbdfe_set_hamming_mode, RGBDFE_ERR_INTERNAL; 3: compact gather records, rgbdfe_sift_detect; 4: rgbdfe_graph_stats, rgbdfe_set_graph_capture, rgbdfe_pack_inliers, rgbdfe_match_pair_list_allgather_inliers


}  // namespace impl
