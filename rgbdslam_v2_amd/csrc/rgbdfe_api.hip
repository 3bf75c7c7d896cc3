// rgbdfe_api.hip -- the extern "C" layer of include/rgbdfe.h: every entry point behind an exception barrier, dispatched to the
// single-device implementation (namespace impl) or to the multi-device group.  There is no CPU fallback: without a HIP device
// every entry point reports RGBDFE_ERR_NO_DEVICE.
#include "rgbdfe_host.h"

extern "C" {

void rgbdfe_default_config(rgbdfe_config* cfg) { impl::rgbdfe_default_config(cfg); }

int rgbdfe_create(const rgbdfe_config* cfg, rgbdfe_ctx** out) {
  return guarded(nullptr, [&]() -> int { return impl::rgbdfe_create(cfg, out); });
}

int rgbdfe_create_multi(const rgbdfe_config* cfg, const int32_t* device_ids, int32_t n_devices, rgbdfe_ctx** out) {
  return guarded(nullptr, [&]() -> int { return group_create(cfg, device_ids, n_devices, out); });
}

void rgbdfe_destroy(rgbdfe_ctx* ctx) {
  if (!ctx) return;
  try {
    if (ctx->group) group_destroy(ctx);
    else impl::rgbdfe_destroy(ctx);
  } catch (...) {
  }
}

int rgbdfe_device_count(rgbdfe_ctx* ctx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return ctx->group ? (int)ctx->group->children.size() : 1;
}

rgbdfe_ctx* rgbdfe_device_context(rgbdfe_ctx* ctx, int32_t i) {
  if (!ctx) return nullptr;
  if (!ctx->group) return i == 0 ? ctx : nullptr;
  return i >= 0 && (size_t)i < ctx->group->children.size() ? ctx->group->children[(size_t)i] : nullptr;
}

int rgbdfe_group_submit_us(rgbdfe_ctx* ctx, double* us) {
  if (!ctx || !us) return RGBDFE_ERR_INVALID_ARG;
  *us = 0.0;
  if (ctx->group) {
    std::lock_guard<std::recursive_mutex> call_lock(ctx->group->mu);
    *us = ctx->group->last_submit_us;
  }
  return RGBDFE_OK;
}

const char* rgbdfe_gather_transport(rgbdfe_ctx* ctx) {
  static thread_local std::string buf;
  if (ctx && ctx->group) {
    std::lock_guard<std::recursive_mutex> call_lock(ctx->group->mu);
    buf = ctx->group->transport;
  } else buf = "none";
  return buf.c_str();
}

int rgbdfe_gather_exchanges(rgbdfe_ctx* ctx) {
  if (!ctx || !ctx->group) return 0;
  std::lock_guard<std::recursive_mutex> call_lock(ctx->group->mu);
  return ctx->group->inl_exchanges;
}

int rgbdfe_set_params(rgbdfe_ctx* ctx, const rgbdfe_params* p) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  if (RGBDFE_IS_GROUP(ctx) && p) {
    std::lock_guard<std::recursive_mutex> call_lock(ctx->group->mu);
    ctx->cfg.params = *p;
    return RGBDFE_ALL(ctx, impl::rgbdfe_set_params(c, p));
  }
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_params(c, p));
}

const char* rgbdfe_status_string(int status) { return impl::rgbdfe_status_string(status); }

// a copy per calling thread: the context's string may be rewritten by another thread at any time
const char* rgbdfe_last_error(rgbdfe_ctx* ctx) {
  static thread_local std::string buf;
  try {
    buf.clear();
    if (ctx) {
      std::lock_guard<std::mutex> g(ctx->err_mu);
      buf = ctx->last_error;
    }
    return buf.c_str();
  } catch (...) {
    return "";
  }
}

int rgbdfe_upload_node(rgbdfe_ctx* ctx, int32_t node_id, const uint8_t* desc, const float* xyz1, int32_t n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_node(c, node_id, desc, xyz1, n));
}

int rgbdfe_upload_nodes(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                        const float* const* xyz1, const int32_t* counts) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_nodes(c, n_nodes, node_ids, desc, xyz1, counts));
}

int rgbdfe_upload_node_device(rgbdfe_ctx* ctx, int32_t node_id, const void* d_desc, const void* d_xyz1, int32_t n,
                              void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_upload_node_device");
    return impl::rgbdfe_upload_node_device(ctx, node_id, d_desc, d_xyz1, n, stream);
  });
}

int rgbdfe_upload_node_keypoints(rgbdfe_ctx* ctx, int32_t node_id, const float* kp_xy, int32_t n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_node_keypoints(c, node_id, kp_xy, n));
}

int rgbdfe_release_node(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_release_node(c, node_id));
}

int rgbdfe_node_count(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    return impl::rgbdfe_node_count(RGBDFE_IS_GROUP(ctx) ? ctx->group->children[0] : ctx, node_id);
  });
}

int rgbdfe_match_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                           rgbdfe_match_result* out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_match(ctx, query_ids, train_ids, n_pairs, out, false, nullptr);
    return impl::rgbdfe_match_pair_list(ctx, query_ids, train_ids, n_pairs, out);
  });
}

int rgbdfe_match_node_pairs(rgbdfe_ctx* ctx, int32_t new_node_id, const int32_t* candidate_ids, int32_t n_pairs,
                            rgbdfe_match_result* out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (n_pairs < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
    std::vector<int32_t> q((size_t)n_pairs, new_node_id);
    if (RGBDFE_IS_GROUP(ctx)) return group_match(ctx, q.data(), candidate_ids, n_pairs, out, false, nullptr);
    return impl::rgbdfe_match_pair_list(ctx, q.data(), candidate_ids, n_pairs, out);
  });
}

int rgbdfe_match_pair_list_allgather(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                     int32_t n_pairs, void* const* d_out, int32_t* records_per_device) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_match_pair_list_allgather needs a context made by rgbdfe_create_multi");
    return group_match_allgather(ctx, query_ids, train_ids, n_pairs, d_out, records_per_device);
  });
}

int rgbdfe_match_pair_list_allgather_compact(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                             int32_t n_pairs, void* const* d_out, int32_t* records_per_device) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_match_pair_list_allgather_compact needs a context made by rgbdfe_create_multi");
    return group_match_allgather(ctx, query_ids, train_ids, n_pairs, d_out, records_per_device, true);
  });
}

int rgbdfe_pack_compact(rgbdfe_ctx* ctx, const void* d_records, int32_t n, void* d_compact, void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_pack_compact");
    return impl::rgbdfe_pack_compact(ctx, d_records, n, d_compact, stream);
  });
}

int rgbdfe_set_graph_capture(rgbdfe_ctx* ctx, int enable) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_graph_capture(c, enable));
}

int rgbdfe_pack_inliers(rgbdfe_ctx* ctx, const void* d_records, int32_t n, int32_t n_headers, void* d_stream, int32_t* d_total,
                        void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_pack_inliers");
    return impl::rgbdfe_pack_inliers(ctx, d_records, n, n_headers, d_stream, d_total, stream);
  });
}
int rgbdfe_sizeof_inlier_header(void) { return impl::rgbdfe_sizeof_inlier_header(); }

int rgbdfe_match_pair_list_allgather_edges(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                           int32_t n_pairs, void* const* d_out, int32_t* const* d_index,
                                           int32_t* edges_per_device, int32_t* stride) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG,
                  "rgbdfe_match_pair_list_allgather_edges needs a context made by rgbdfe_create_multi");
    return group_match_allgather_edges(ctx, query_ids, train_ids, n_pairs, d_out, d_index, edges_per_device, stride);
  });
}

int rgbdfe_match_pair_list_allgather_inliers(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                             int32_t n_pairs, void* const* d_out, int32_t* records_per_device,
                                             int32_t* list_entries, int64_t* stride_bytes) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG,
                  "rgbdfe_match_pair_list_allgather_inliers needs a context made by rgbdfe_create_multi");
    return group_match_allgather_inliers(ctx, query_ids, train_ids, n_pairs, d_out, records_per_device, list_entries, stride_bytes);
  });
}

int rgbdfe_submit_pair_list_host(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 void* out, size_t out_bytes, int payload, int64_t* ticket) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_submit_pair_list_host");
    return impl::rgbdfe_submit_pair_list_host(ctx, query_ids, train_ids, n_pairs, out, out_bytes, payload, ticket);
  });
}
int rgbdfe_wait_host(rgbdfe_ctx* ctx, int64_t ticket, int64_t* bytes_written) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_wait_host");
    return impl::rgbdfe_wait_host(ctx, ticket, bytes_written);
  });
}

int rgbdfe_wait_host_into(rgbdfe_ctx* ctx, int64_t ticket, void* out, size_t out_bytes, int64_t* bytes_written) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_wait_host_into");
    return impl::rgbdfe_wait_host_into(ctx, ticket, out, out_bytes, bytes_written);
  });
}

int rgbdfe_match_pair_list_device(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                  void* d_out, void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_match_pair_list_device");
    return impl::rgbdfe_match_pair_list_device(ctx, query_ids, train_ids, n_pairs, d_out, stream);
  });
}

int rgbdfe_submit_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                            void* d_out, int64_t* ticket) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_submit_pair_list");
    return impl::rgbdfe_submit_pair_list(ctx, query_ids, train_ids, n_pairs, d_out, ticket);
  });
}

int rgbdfe_wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_wait_ticket");
    return impl::rgbdfe_wait_ticket(ctx, ticket, stream);
  });
}

int rgbdfe_synchronize(rgbdfe_ctx* ctx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_synchronize(c));
}

int rgbdfe_upload_sift_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc128, const float* xyz1, int32_t n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_sift_node(c, node_id, desc128, xyz1, n));
}

int rgbdfe_match_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                rgbdfe_match_result* out, float* out_dist) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_match(ctx, query_ids, train_ids, n_pairs, out, true, out_dist);
    return impl::rgbdfe_match_sift_pair_list(ctx, query_ids, train_ids, n_pairs, out, out_dist);
  });
}

int rgbdfe_upload_float_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc, int32_t dim, const float* xyz1,
                             int32_t n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_float_node(c, node_id, desc, dim, xyz1, n));
}

int rgbdfe_match_flann_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 double nn_distance_ratio, rgbdfe_match_result* out, float* out_dist) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!(nn_distance_ratio > 0.0)) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "nn_distance_ratio must be > 0");
    if (RGBDFE_IS_GROUP(ctx)) {
      if (n_pairs < 0 || (n_pairs > 0 && (!query_ids || !train_ids || !out)))
        return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
      Group& g = *ctx->group;
      const int G = (int)g.children.size();
      return group_run(ctx, [&](int i) -> int {
        std::vector<int32_t> qs, ts;
        for (int32_t k = i; k < n_pairs; k += G) { qs.push_back(query_ids[k]); ts.push_back(train_ids[k]); }
        if (qs.empty()) return RGBDFE_OK;
        return impl::rgbdfe_match_sift_pair_list(g.children[(size_t)i], qs.data(), ts.data(), (int32_t)qs.size(), out + i,
                                                 out_dist ? out_dist + (size_t)i * RGBDFE_MAX_MATCHES : nullptr, G, 2,
                                                 nn_distance_ratio);
      });
    }
    return impl::rgbdfe_match_sift_pair_list(ctx, query_ids, train_ids, n_pairs, out, out_dist, 1, 2, nn_distance_ratio);
  });
}

int rgbdfe_submit_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 void* d_out, void* d_out_dist, int64_t* ticket) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (RGBDFE_IS_GROUP(ctx)) return group_only_single(ctx, "rgbdfe_submit_sift_pair_list");
    return impl::rgbdfe_submit_sift_pair_list(ctx, query_ids, train_ids, n_pairs, d_out, d_out_dist, ticket);
  });
}

int rgbdfe_sift_match_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* match_q, int32_t* match_t,
                            float* match_dist, int32_t* n_matches) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_match_nodes(c, query_id, train_id, match_q, match_t, match_dist, n_matches));
}

int rgbdfe_detector_configure(rgbdfe_ctx* ctx, int32_t max_keypoints, int32_t grid_resolution,
                              int32_t adjuster_max_iterations) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_detector_configure(c, max_keypoints, grid_resolution, adjuster_max_iterations));
}

int rgbdfe_detector_thresholds(rgbdfe_ctx* ctx, double* thresholds, int32_t* n_cells) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_detector_thresholds(c, thresholds, n_cells));
}

int rgbdfe_orb_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols,
                      int32_t fast_threshold, rgbdfe_keypoint* keypoints, int32_t capacity, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_orb_detect(c, gray, mask, rows, cols, fast_threshold, keypoints, capacity, n_out));
}

int rgbdfe_orb_compute(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints,
                       int32_t n, uint8_t* descriptors, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_orb_compute(c, gray, rows, cols, keypoints, n, descriptors, n_out));
}

int rgbdfe_detect_describe(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* depth, int32_t rows,
                           int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling,
                           rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_detect_describe(c, gray, mask, depth, rows, cols, fx, fy, cx, cy, depth_scaling,
                                                        keypoints, descriptors, xyz1, n_out));
}

int rgbdfe_set_feature_min_depth(rgbdfe_ctx* ctx, int32_t on) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_feature_min_depth(c, on));
}

int rgbdfe_project_to_3d_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp,
                                   const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                   double cy, double depth_scaling, int32_t max_keypoints, int32_t* kept_idx,
                                   float* xyz1, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_project_to_3d_min_depth(c, kp_xy, kp_size, n_kp, depth, rows, cols, fx, fy, cx, cy,
                                                                depth_scaling, max_keypoints, kept_idx, xyz1, n_out));
}

int rgbdfe_detect_describe_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                 const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                 double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_detect_describe_batch(c, n_frames, gray, mask, depth, rows, cols, fx, fy, cx, cy,
                                                              depth_scaling, out_stride, keypoints, descriptors, xyz1,
                                                              n_out));
}

int rgbdfe_detect_describe_batch_nodes(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                       const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                       double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                       uint8_t* descriptors, float* xyz1, int32_t* n_out, const int32_t* node_ids) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  if (!node_ids) return guarded(ctx, [&]() -> int { return fail(ctx, RGBDFE_ERR_INVALID_ARG, "node_ids is required"); });
  if (!RGBDFE_IS_GROUP(ctx))
    return RGBDFE_FIRST(ctx, impl::rgbdfe_detect_describe_batch(c, n_frames, gray, mask, depth, rows, cols, fx, fy, cx, cy,
                                                                depth_scaling, out_stride, keypoints, descriptors, xyz1, n_out,
                                                                node_ids));
  // several devices behind the handle: the frames are processed on the first one, the nodes go to all of them from the host
  // outputs (every device holds every node)
  int rc = RGBDFE_FIRST(ctx, impl::rgbdfe_detect_describe_batch(c, n_frames, gray, mask, depth, rows, cols, fx, fy, cx, cy,
                                                                depth_scaling, out_stride, keypoints, descriptors, xyz1, n_out));
  if (rc != RGBDFE_OK) return rc;
  std::vector<int32_t> ids, cnt;
  std::vector<const uint8_t*> dp;
  std::vector<const float*> xp;
  for (int32_t f = 0; f < n_frames; ++f)
    if (node_ids[f] >= 0) {
      ids.push_back(node_ids[f]); cnt.push_back(n_out[f]);
      dp.push_back(descriptors + (size_t)f * out_stride * 32); xp.push_back(xyz1 + (size_t)f * out_stride * 4);
    }
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_nodes(c, (int32_t)ids.size(), ids.data(), dp.data(), xp.data(), cnt.data()));
}

int rgbdfe_sift_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols,
                       int32_t max_keypoints, rgbdfe_keypoint* keypoints, float* desc128, int32_t capacity, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_detect(c, gray, mask, rows, cols, max_keypoints, keypoints, desc128, capacity,
                                                    n_out));
}

int rgbdfe_sift_describe(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints, int32_t n,
                         float* desc128) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_describe(c, gray, rows, cols, keypoints, n, desc128));
}

int rgbdfe_sift_detect_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, int32_t rows, int32_t cols,
                             int32_t max_keypoints, int32_t out_stride, rgbdfe_keypoint* keypoints, float* desc128,
                             int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_detect_batch(c, n_frames, gray, rows, cols, max_keypoints, out_stride, keypoints,
                                                          desc128, n_out));
}

int rgbdfe_sift_debug_plane(rgbdfe_ctx* ctx, int32_t octave, int32_t level, float* out, int32_t capacity_floats, int32_t* w,
                            int32_t* h) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_debug_plane(c, octave, level, out, capacity_floats, w, h));
}

int rgbdfe_sift_debug_candidates(rgbdfe_ctx* ctx, int32_t octave, int32_t dog_level, float* out, int32_t capacity_rows,
                                 int32_t* n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_debug_candidates(c, octave, dog_level, out, capacity_rows, n));
}

int rgbdfe_sift_geometry(rgbdfe_ctx* ctx, int32_t* octave_min, int32_t* octave_num, int32_t* levels, int32_t* dog_levels) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_geometry(c, octave_min, octave_num, levels, dog_levels));
}

int rgbdfe_hamming_nn_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* out_hd, int32_t* out_idx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_hamming_nn_nodes(c, query_id, train_id, out_hd, out_idx));
}

int rgbdfe_place_recognition(rgbdfe_ctx* ctx, int32_t query_id, const int32_t* candidate_ids, int32_t n_candidates,
                             int32_t k_neighbours, int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                             int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_place_recognition(c, query_id, candidate_ids, n_candidates, k_neighbours, max_hd,
                                                          max_out, out_ids, out_scores, n_out));
}

int rgbdfe_place_recognition_batch(rgbdfe_ctx* ctx, const int32_t* query_ids, int32_t n_queries,
                                   const int32_t* candidate_offsets, const int32_t* candidate_ids, int32_t k_neighbours,
                                   int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores,
                                   int32_t* out_counts) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_place_recognition_batch(c, query_ids, n_queries, candidate_offsets, candidate_ids,
                                                                k_neighbours, max_hd, max_out, out_ids, out_scores,
                                                                out_counts));
}

int rgbdfe_hamming_nn_host(rgbdfe_ctx* ctx, const uint8_t* qdesc, int32_t nq, const uint8_t* tdesc, int32_t nt,
                           int32_t* out_hd, int32_t* out_idx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_hamming_nn_host(c, qdesc, nq, tdesc, nt, out_hd, out_idx));
}

int rgbdfe_project_to_3d(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* depth, int32_t rows,
                         int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling,
                         int32_t max_keypoints, int32_t* kept_idx, float* xyz1, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_project_to_3d(c, kp_xy, n_kp, depth, rows, cols, fx, fy, cx, cy, depth_scaling,
                                                      max_keypoints, kept_idx, xyz1, n_out));
}

int rgbdfe_project_to_3d_cloud(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* cloud, int32_t rows,
                               int32_t cols, double maximum_depth, int32_t max_keypoints, int32_t* kept_idx, float* xyz1,
                               int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_project_to_3d_cloud(c, kp_xy, n_kp, cloud, rows, cols, maximum_depth,
                                                            max_keypoints, kept_idx, xyz1, n_out));
}

int rgbdfe_detect_describe_cloud(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* cloud,
                                 int32_t rows, int32_t cols, double maximum_depth, rgbdfe_keypoint* keypoints,
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_detect_describe_cloud(c, gray, mask, cloud, rows, cols, maximum_depth, keypoints,
                                                              descriptors, xyz1, n_out));
}

int rgbdfe_sift_node_features(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* desc_in,
                              const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                              double depth_scaling, int32_t max_keypoints, int32_t use_root_sift, int32_t* kept_idx,
                              float* xyz1, float* siftgpu_descriptors, float* feature_descriptors, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_node_features(c, kp_xy, nullptr, n_kp, desc_in, depth, rows, cols, fx, fy, cx, cy,
                                                           depth_scaling, max_keypoints, use_root_sift, kept_idx, xyz1,
                                                           siftgpu_descriptors, feature_descriptors, n_out));
}

int rgbdfe_sift_node_features_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp,
                                        const float* desc_in, const float* depth, int32_t rows, int32_t cols, double fx,
                                        double fy, double cx, double cy, double depth_scaling, int32_t max_keypoints,
                                        int32_t use_root_sift, int32_t* kept_idx, float* xyz1, float* siftgpu_descriptors,
                                        float* feature_descriptors, int32_t* n_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  if (n_kp > 0 && !kp_size) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "kp_size is required");
  return RGBDFE_FIRST(ctx, impl::rgbdfe_sift_node_features(c, kp_xy, kp_size, n_kp, desc_in, depth, rows, cols, fx, fy, cx,
                                                           cy, depth_scaling, max_keypoints, use_root_sift, kept_idx, xyz1,
                                                           siftgpu_descriptors, feature_descriptors, n_out));
}

int rgbdfe_host_register(rgbdfe_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!ptr || bytes == 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, RGBDFE_ERR_HIP, "hipHostRegister failed (already registered, or not host memory)");
    }
    return RGBDFE_OK;
  });
}

int rgbdfe_host_unregister(rgbdfe_ctx* ctx, void* ptr) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!ptr) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
    if (hipHostUnregister(ptr) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, RGBDFE_ERR_HIP, "hipHostUnregister failed (not a registered range)");
    }
    return RGBDFE_OK;
  });
}

int rgbdfe_depth_to_mono8(rgbdfe_ctx* ctx, const void* depth, int32_t depth_is_u16, int32_t rows, int32_t cols,
                          uint8_t* mono8, float* depth_m) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_FIRST(ctx, impl::rgbdfe_depth_to_mono8(c, depth, depth_is_u16, rows, cols, mono8, depth_m));
}

int rgbdfe_upload_node_cloud(rgbdfe_ctx* ctx, int32_t node_id, const float* depth, int32_t rows, int32_t cols,
                             const uint8_t* rgb, int32_t rgb_channels, int32_t encoding_bgr, double fx, double fy,
                             double cx, double cy, double depth_scaling, double min_depth, int32_t cloud_skip,
                             float* cloud_out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  // replicated like the node features, so that the measurement-model jobs can be sharded; the copy comes from device 0
  return RGBDFE_ALL(ctx, impl::rgbdfe_upload_node_cloud(c, node_id, depth, rows, cols, rgb, rgb_channels, encoding_bgr, fx,
                                                        fy, cx, cy, depth_scaling, min_depth, cloud_skip,
                                                        (!RGBDFE_IS_GROUP(ctx) || c == ctx->group->children[0]) ? cloud_out
                                                                                                               : nullptr));
}

int rgbdfe_release_node_cloud(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_release_node_cloud(c, node_id));
}

int rgbdfe_observation_likelihood(rgbdfe_ctx* ctx, int32_t n, const int32_t* new_ids, const int32_t* old_ids,
                                  const float* transforms, int32_t emm_skip_step, rgbdfe_emm_counts* out) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx))
      return impl::rgbdfe_observation_likelihood(ctx, n, new_ids, old_ids, transforms, emm_skip_step, out);
    if (n < 0 || (n > 0 && (!new_ids || !old_ids || !transforms || !out)))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
    Group& g = *ctx->group;
    const int G = (int)g.children.size();
    return group_run(ctx, [&](int i) -> int {  // job k -> device k mod G
      std::vector<int32_t> a, b;
      std::vector<float> T;
      for (int32_t k = i; k < n; k += G) {
        a.push_back(new_ids[k]); b.push_back(old_ids[k]);
        T.insert(T.end(), transforms + (size_t)k * 16, transforms + (size_t)k * 16 + 16);
      }
      if (a.empty()) return RGBDFE_OK;
      std::vector<rgbdfe_emm_counts> o(a.size());
      const int rc = impl::rgbdfe_observation_likelihood(g.children[(size_t)i], (int32_t)a.size(), a.data(), b.data(),
                                                         T.data(), emm_skip_step, o.data());
      if (rc != RGBDFE_OK) return rc;
      for (size_t m = 0; m < o.size(); ++m) out[(size_t)i + m * (size_t)G] = o[m];
      return RGBDFE_OK;
    });
  });
}

int rgbdfe_observation_criterion_met(uint32_t inliers, uint32_t outliers, uint32_t all, double observability_threshold,
                                     double* quality) {
  return impl::rgbdfe_observation_criterion_met(inliers, outliers, all, observability_threshold, quality);
}

int rgbdfe_set_latency_mode(rgbdfe_ctx* ctx, int32_t max_pairs, int32_t chunk_iterations) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_latency_mode(c, max_pairs, chunk_iterations));
}

int rgbdfe_set_hamming_mode(rgbdfe_ctx* ctx, int32_t mode) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_hamming_mode(c, mode));
}

int rgbdfe_set_profiling(rgbdfe_ctx* ctx, int enable) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_set_profiling(c, enable));
}

// group: the sum over the devices
int rgbdfe_get_kernel_time(rgbdfe_ctx* ctx, int which, double* total_ms, int64_t* launches, int64_t* pairs) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    if (!RGBDFE_IS_GROUP(ctx)) return impl::rgbdfe_get_kernel_time(ctx, which, total_ms, launches, pairs);
    double ms = 0; int64_t nl = 0, np = 0;
    for (rgbdfe_ctx* c : ctx->group->children) {
      double a = 0; int64_t b = 0, d = 0;
      const int rc = impl::rgbdfe_get_kernel_time(c, which, &a, &b, &d);
      if (rc != RGBDFE_OK) return rc;
      ms += a; nl += b; np += d;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = nl;
    if (pairs) *pairs = np;
    return RGBDFE_OK;
  });
}

int rgbdfe_reset_kernel_time(rgbdfe_ctx* ctx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return RGBDFE_ALL(ctx, impl::rgbdfe_reset_kernel_time(c));
}

// group: the sum over the devices
int rgbdfe_graph_stats(rgbdfe_ctx* ctx, int64_t* out, int32_t n_out) {
  if (!ctx || !out || n_out < 0) return RGBDFE_ERR_INVALID_ARG;
  return guarded(ctx, [&]() -> int {
    for (int32_t i = 0; i < n_out; ++i) out[i] = 0;
    if (!RGBDFE_IS_GROUP(ctx)) return impl::rgbdfe_graph_stats(ctx, out, n_out);
    for (rgbdfe_ctx* c : ctx->group->children) {
      const int rc = impl::rgbdfe_graph_stats(c, out, n_out);
      if (rc != RGBDFE_OK) return rc;
    }
    return RGBDFE_OK;
  });
}

int rgbdfe_sizeof_match_result(void) { return impl::rgbdfe_sizeof_match_result(); }
int rgbdfe_sizeof_compact_result(void) { return impl::rgbdfe_sizeof_compact_result(); }
int rgbdfe_abi_version(void) { return impl::rgbdfe_abi_version(); }

}  // extern "C"

