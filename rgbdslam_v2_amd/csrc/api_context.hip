// api_context.hip -- context lifetime, parameters, node residency (rgbdfe_create ... rgbdfe_node_count)
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include "rgbdfe_host.h"

namespace impl {


void rgbdfe_default_config(rgbdfe_config* cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof(*cfg));
  cfg->device_id = 0;
  cfg->max_nodes = 256;
  cfg->max_keypoints = 1024;
  cfg->max_pairs_per_batch = 4096;
  cfg->params.max_matches = 300;           // parameter_server.cpp:86
  cfg->params.min_matches = 20;            // parameter_server.cpp:85
  cfg->params.ransac_iterations = 200;     // parameter_server.cpp:101
  cfg->params.max_dist_for_inliers = 3.0f; // parameter_server.cpp:100
  cfg->params.depth_cov = 1e-4;            // (sigma_depth=0.01 * (1 m)^2)^2, misc2.h:20-35
  cfg->params.seed = 20260923u;
}

int rgbdfe_create(const rgbdfe_config* cfg, rgbdfe_ctx** out) {
  if (!cfg || !out) return RGBDFE_ERR_INVALID_ARG;
  *out = nullptr;
  if (cfg->max_nodes < 1 || cfg->max_keypoints < 1 || cfg->max_keypoints > RGBDFE_MAX_KEYPOINTS ||
      cfg->max_pairs_per_batch < 1)
    return RGBDFE_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RGBDFE_ERR_NO_DEVICE;
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return RGBDFE_ERR_NO_DEVICE;
  rgbdfe_ctx* ctx = new rgbdfe_ctx();
  ctx->cfg = *cfg;
  int rc = validate_params(ctx, cfg->params);
  if (rc != RGBDFE_OK) { delete ctx; return rc; }
  fill_ransac_const(ctx);
  if (const char* sf = getenv("RGBDFE_SIFT_FAST_KEYS")) ctx->sift_fast = atoi(sf) != 0;
  if (const char* hm = getenv("RGBDFE_HAMMING_MODE")) ctx->hamming_mode = atoi(hm) < 0 || atoi(hm) > 3 ? RGBDFE_HAMMING_MODE_DEFAULT : atoi(hm);
  if (const char* gr = getenv("RGBDFE_GRAPHS")) ctx->use_graphs = atoi(gr) != 0;   // (rgbdfe_set_graph_capture overrides)
  auto bail = [&](int code) { ::rgbdfe_destroy(ctx); return code; };   // (the C entry point: it owns the teardown order)
  if (hipSetDevice(cfg->device_id) != hipSuccess) return bail(RGBDFE_ERR_NO_DEVICE);
  (void)ransac_split_init();  // kernel attributes of the RANSAC refinement kernel: once, outside any stream capture
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(RGBDFE_ERR_HIP);
  if (hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming) != hipSuccess) return bail(RGBDFE_ERR_HIP);
  const size_t rows = (size_t)cfg->max_nodes * (size_t)cfg->max_keypoints + 16;  // +pad: prefetch overrun
  if (hipMalloc((void**)&ctx->d_desc, rows * 32) != hipSuccess) return bail(RGBDFE_ERR_OUT_OF_MEMORY);
  if (hipMalloc((void**)&ctx->d_xyz, rows * 16) != hipSuccess) return bail(RGBDFE_ERR_OUT_OF_MEMORY);
  // max_keypoints is rounded up to whole 32-row tiles per slot in the expanded slab
  if (hipMalloc((void**)&ctx->d_desc4, hamming_mfma_slab_bytes((uint32_t)cfg->max_nodes, (uint32_t)cfg->max_keypoints)) != hipSuccess)
    return bail(RGBDFE_ERR_OUT_OF_MEMORY);
  // The zero-fills run on the context's own stream and the call waits for THAT stream (a NULL-stream hipMemset is not
  // ordered before the context's non-blocking streams without a device-wide synchronisation -- and hipDeviceSynchronize()
  // invalidates a hipGraph capture another context's thread may have open).
  if (hipMemsetAsync(ctx->d_desc, 0, rows * 32, ctx->stream) != hipSuccess) return bail(RGBDFE_ERR_HIP);
  if (hipMemsetAsync(ctx->d_xyz, 0, rows * 16, ctx->stream) != hipSuccess) return bail(RGBDFE_ERR_HIP);
  if (hipMemsetAsync(ctx->d_desc4, 0, hamming_mfma_slab_bytes((uint32_t)cfg->max_nodes, (uint32_t)cfg->max_keypoints), ctx->stream) != hipSuccess)
    return bail(RGBDFE_ERR_HIP);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(RGBDFE_ERR_HIP);
  const size_t np = (size_t)cfg->max_pairs_per_batch;
  for (auto& sl : ctx->ring) {
    if (hipMalloc((void**)&sl.d_work, np * sizeof(PairWork)) != hipSuccess) return bail(RGBDFE_ERR_OUT_OF_MEMORY);
    if (hipHostMalloc((void**)&sl.h_work, np * sizeof(PairWork), hipHostMallocDefault) != hipSuccess)
      return bail(RGBDFE_ERR_OUT_OF_MEMORY);
    if (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) return bail(RGBDFE_ERR_HIP);
  }
  for (auto& ln : ctx->lanes) {
    if (hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) != hipSuccess) return bail(RGBDFE_ERR_HIP);
    if (hipMalloc((void**)&ln.d_keys, np * (size_t)cfg->max_keypoints * 4) != hipSuccess)
      return bail(RGBDFE_ERR_OUT_OF_MEMORY);
    if (hipMalloc((void**)&ln.d_results, np * sizeof(rgbdfe_match_result)) != hipSuccess)
      return bail(RGBDFE_ERR_OUT_OF_MEMORY);
    if (hipMalloc((void**)&ln.d_prep, np * sizeof(PairPrep)) != hipSuccess) return bail(RGBDFE_ERR_OUT_OF_MEMORY);
  }
  ctx->emm_q_lo = erf_boundary(0.001);
  ctx->emm_q_hi = erf_boundary(0.999);
  ctx->free_slots.reserve(cfg->max_nodes);
  for (int32_t s = cfg->max_nodes - 1; s >= 0; --s) ctx->free_slots.push_back((uint32_t)s);
  *out = ctx;
  return RGBDFE_OK;
}

void rgbdfe_destroy(rgbdfe_ctx* ctx) {
  if (!ctx) return;
  // every stream this context has work on -- not the device: another context's thread may be capturing a hipGraph
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& ln : ctx->lanes) if (ln.stream) (void)hipStreamSynchronize(ln.stream);
  for (hipStream_t st : {ctx->orb_upload_stream, ctx->orb_compute_stream, ctx->sift_stream1, ctx->sift_stream2, ctx->sift_stream3})
    if (st) (void)hipStreamSynchronize(st);
  drain_pending(ctx);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& ge : ctx->graphs) { (void)hipGraphExecDestroy(ge.exec); (void)hipGraphDestroy(ge.graph); }
  if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
  if (ctx->upload_stage) (void)hipHostFree(ctx->upload_stage);
  if (ctx->sift_stream1) (void)hipStreamDestroy(ctx->sift_stream1);
  if (ctx->sift_stream2) (void)hipStreamDestroy(ctx->sift_stream2);
  if (ctx->sift_stream3) (void)hipStreamDestroy(ctx->sift_stream3);
  ctx->graphs.clear();
  if (ctx->d_desc) (void)hipFree(ctx->d_desc);
  if (ctx->d_xyz) (void)hipFree(ctx->d_xyz);
  for (auto& sl : ctx->ring) {
    if (sl.d_work) (void)hipFree(sl.d_work);
    if (sl.h_work) (void)hipHostFree(sl.h_work);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  for (auto& kv : ctx->clouds) {
    if (kv.second.d) (void)hipFree(kv.second.d);
    if (kv.second.d_samples) (void)hipFree(kv.second.d_samples);
  }
  if (ctx->d_sift_bf16) (void)hipFree(ctx->d_sift_bf16);
  if (ctx->d_sift_f32) (void)hipFree(ctx->d_sift_f32);
  for (auto& ln : ctx->lanes) {
    if (ln.d_row_part) (void)hipFree(ln.d_row_part);
    if (ln.d_col_part) (void)hipFree(ln.d_col_part);
    if (ln.d_col_blocks) (void)hipFree(ln.d_col_blocks);
    if (ln.d_sm_q) (void)hipFree(ln.d_sm_q);
    if (ln.d_sm_t) (void)hipFree(ln.d_sm_t);
    if (ln.d_sm_d) (void)hipFree(ln.d_sm_d);
    if (ln.d_sm_n) (void)hipFree(ln.d_sm_n);
    if (ln.d_all_dist) (void)hipFree(ln.d_all_dist);
    if (ln.d_recs) (void)hipFree(ln.d_recs);
    if (ln.d_walk) (void)hipFree(ln.d_walk);
    if (ln.d_prep) (void)hipFree(ln.d_prep);
    if (ln.d_ec) (void)hipFree(ln.d_ec);
    if (ln.d_keys) (void)hipFree(ln.d_keys);
    if (ln.d_results) (void)hipFree(ln.d_results);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
  }
  if (ctx->h_results) (void)hipHostFree(ctx->h_results);
  for (int li = 0; li < rgbdfe_ctx::kLanes; ++li) {
    if (ctx->host_jobs[li].copied) (void)hipEventDestroy(ctx->host_jobs[li].copied);
    if (ctx->h_stage[li]) (void)hipHostFree(ctx->h_stage[li]);
    if (ctx->d_inl_stream[li]) (void)hipFree(ctx->d_inl_stream[li]);
    if (ctx->d_inl_total[li]) (void)hipFree(ctx->d_inl_total[li]);
    if (ctx->h_inl_total[li]) (void)hipHostFree(ctx->h_inl_total[li]);
  }
  if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
  if (ctx->nodes_ready_ev) (void)hipEventDestroy(ctx->nodes_ready_ev);
  if (ctx->d_desc4) (void)hipFree(ctx->d_desc4);
  if (ctx->d_kp2d) (void)hipFree(ctx->d_kp2d);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  for (hipEvent_t e : ctx->orb_upload_done) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->orb_describe_done) if (e) (void)hipEventDestroy(e);
  if (ctx->orb_upload_stream) (void)hipStreamDestroy(ctx->orb_upload_stream);
  if (ctx->orb_compute_stream) (void)hipStreamDestroy(ctx->orb_compute_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int rgbdfe_set_params(rgbdfe_ctx* ctx, const rgbdfe_params* p) {
  if (!ctx || !p) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  int rc = validate_params(ctx, *p);
  if (rc != RGBDFE_OK) return rc;
  ctx->cfg.params = *p;
  fill_ransac_const(ctx);
  return RGBDFE_OK;
}

const char* rgbdfe_status_string(int status) {
  switch (status) {
    case RGBDFE_OK: return "ok";
    case RGBDFE_ERR_INVALID_ARG: return "invalid argument";
    case RGBDFE_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case RGBDFE_ERR_HIP: return "HIP runtime error";
    case RGBDFE_ERR_UNKNOWN_NODE: return "unknown node id";
    case RGBDFE_ERR_CAPACITY: return "capacity exceeded";
    case RGBDFE_ERR_OUT_OF_MEMORY: return "out of memory";
    case RGBDFE_ERR_INTERNAL: return "internal error (exception caught at the ABI)";
    default: return "unknown status";
  }
}

const char* rgbdfe_last_error(rgbdfe_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

static int upload_common(rgbdfe_ctx* ctx, int32_t node_id, const void* desc, const void* xyz1,
                         int32_t n, hipMemcpyKind kind, hipStream_t stream, bool sync) {
  if (!ctx || n < 0 || (n > 0 && (!desc || !xyz1))) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments");
  if (n > ctx->cfg.max_keypoints) return fail(ctx, RGBDFE_ERR_CAPACITY, "node has more rows than max_keypoints");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  uint32_t slot;
  auto it = ctx->nodes.find(node_id);
  if (it != ctx->nodes.end()) {
    slot = it->second.slot;  // overwrite in place: wait for batches that may still read it
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  } else {
    if (ctx->free_slots.empty()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
    slot = ctx->free_slots.back();
    ctx->free_slots.pop_back();
  }
  const size_t row0 = (size_t)slot * (size_t)ctx->cfg.max_keypoints;
  if (n > 0) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_desc + row0 * 8, desc, (size_t)n * 32, kind, stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_xyz + row0, xyz1, (size_t)n * 16, kind, stream));
  }
  if (n > 0) {
    // the MFMA Hamming kernel reads the descriptors in their expanded (fp4 operand) form: built here, once per node
    launch_hamming_expand(ctx->d_desc + row0 * 8, ctx->d_desc4, slot, (uint32_t)ctx->cfg.max_keypoints, (uint32_t)n, stream);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (sync) HIP_TRY(ctx, hipStreamSynchronize(stream));
  else {
    // Ordering contract of rgbdfe_upload_node_device with a caller stream: the copies are enqueued on that stream and
    // every batch submitted afterwards (on the context's internal streams) waits for them through this event.
    if (!ctx->nodes_ready_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->nodes_ready_ev, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->nodes_ready_ev, stream));
    ctx->nodes_ready = ctx->nodes_ready_ev;
  }
  ctx->nodes[node_id] = NodeEntry{slot, (uint32_t)n, 0u, 0u};
  return RGBDFE_OK;
}

int rgbdfe_upload_node(rgbdfe_ctx* ctx, int32_t node_id, const uint8_t* desc, const float* xyz1,
                       int32_t n) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  return upload_common(ctx, node_id, desc, xyz1, n, hipMemcpyHostToDevice, ctx->stream, true);
}

// Many nodes in one call (an offline run hands over the nodes of a stretch of frames): every node's rows go through one
// pinned staging buffer, the copies and expansion kernels of all nodes are enqueued back to back and the host waits once --
// a single rgbdfe_upload_node is two pageable copies, a launch and a synchronisation (~65 us), here a node costs its
// three enqueues.  All-or-nothing on argument / capacity errors (checked before anything is copied).
int rgbdfe_upload_nodes(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                        const float* const* xyz1, const int32_t* counts) {
  if (!ctx || n_nodes < 0 || (n_nodes > 0 && (!node_ids || !desc || !xyz1 || !counts)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  return upload_nodes_locked(ctx, n_nodes, node_ids, desc, xyz1, counts);
}

int upload_nodes_locked(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                               const float* const* xyz1, const int32_t* counts) {
  size_t rows = 0, fresh = 0;
  bool overwrite = false;
  std::unordered_set<int32_t> seen;
  seen.reserve((size_t)n_nodes * 2);
  for (int32_t i = 0; i < n_nodes; ++i) {
    if (counts[i] < 0 || (counts[i] > 0 && (!desc[i] || !xyz1[i]))) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments");
    if (counts[i] > ctx->cfg.max_keypoints) return fail(ctx, RGBDFE_ERR_CAPACITY, "node has more rows than max_keypoints");
    if (!seen.insert(node_ids[i]).second) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "a node id appears twice in one upload");
    if (ctx->nodes.count(node_ids[i])) overwrite = true; else ++fresh;
    rows += (size_t)counts[i];
  }
  if (fresh > ctx->free_slots.size()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
  if (overwrite)  // nodes rewritten in place: wait for batches that may still read them
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  if (rows * 48 > ctx->upload_stage_bytes) {
    if (ctx->upload_stage) (void)hipHostFree(ctx->upload_stage);
    ctx->upload_stage = nullptr; ctx->upload_stage_bytes = 0;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->upload_stage, rows * 48 * 2, hipHostMallocDefault));
    ctx->upload_stage_bytes = rows * 48 * 2;
  }
  uint8_t* stage = ctx->upload_stage;
  // A node is registered BEFORE its copies are enqueued, so a failing enqueue leaves no slot unaccounted for: the node is
  // resident with whatever reached it (the caller gets the error and uploads it again or releases it).  Whatever the
  // outcome, the copies out of the pinned stage have ended when this returns -- the next call overwrites the stage.
  hipError_t err = hipSuccess;
  for (int32_t i = 0; i < n_nodes && err == hipSuccess; ++i) {
    const int32_t n = counts[i];
    uint32_t slot;
    auto it = ctx->nodes.find(node_ids[i]);
    if (it != ctx->nodes.end()) slot = it->second.slot;
    else { slot = ctx->free_slots.back(); ctx->free_slots.pop_back(); }
    ctx->nodes[node_ids[i]] = NodeEntry{slot, (uint32_t)n, 0u, 0u};
    const size_t row0 = (size_t)slot * (size_t)ctx->cfg.max_keypoints;
    if (n > 0) {
      memcpy(stage, desc[i], (size_t)n * 32);
      memcpy(stage + (size_t)n * 32, xyz1[i], (size_t)n * 16);
      err = hipMemcpyAsync(ctx->d_desc + row0 * 8, stage, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream);
      if (err == hipSuccess)
        err = hipMemcpyAsync(ctx->d_xyz + row0, stage + (size_t)n * 32, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
      if (err == hipSuccess) {
        launch_hamming_expand(ctx->d_desc + row0 * 8, ctx->d_desc4, slot, (uint32_t)ctx->cfg.max_keypoints, (uint32_t)n, ctx->stream);
        err = hipGetLastError();
      }
      stage += (size_t)n * 48;
    }
  }
  const hipError_t sync_err = hipStreamSynchronize(ctx->stream);
  if (err == hipSuccess) err = sync_err;
  if (err != hipSuccess) return fail(ctx, RGBDFE_ERR_HIP, std::string("rgbdfe_upload_nodes: ") + hipGetErrorString(err));
  return RGBDFE_OK;
}

int rgbdfe_upload_node_device(rgbdfe_ctx* ctx, int32_t node_id, const void* d_desc,
                              const void* d_xyz1, int32_t n, void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  return upload_common(ctx, node_id, d_desc, d_xyz1, n, hipMemcpyDeviceToDevice, s, stream == nullptr);
}

int rgbdfe_upload_node_keypoints(rgbdfe_ctx* ctx, int32_t node_id, const float* kp_xy, int32_t n) {
  if (!ctx || n < 0 || (n > 0 && !kp_xy)) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad keypoint upload arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  auto it = ctx->nodes.find(node_id);
  if (it == ctx->nodes.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "keypoints of a node that is not resident");
  if ((uint32_t)n != it->second.n) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "keypoint count differs from the node's rows");
  if (!ctx->d_kp2d) {
    const size_t rows = (size_t)ctx->cfg.max_nodes * (size_t)ctx->cfg.max_keypoints;
    if (hipMalloc((void**)&ctx->d_kp2d, rows * 8) != hipSuccess) return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "keypoint slab");
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_kp2d, 0, rows * 8, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (this stream only: see rgbdfe_create)
  }
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));  // batches in flight may read the slot
  if (n > 0)
  {   // (not a NULL-stream copy: refused while another thread of the process has a stream capture open, see orb_host.hip)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_kp2d + (size_t)it->second.slot * (size_t)ctx->cfg.max_keypoints * 2, kp_xy,
                                (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  it->second.flags |= kNodeHasKeypoints;  // every upload into the slot builds a fresh NodeEntry, i.e. clears it
  return RGBDFE_OK;
}

int rgbdfe_release_node(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = ctx->nodes.find(node_id);
  if (it == ctx->nodes.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "release of unknown node");
  // batches in flight may still read this slot
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  ctx->free_slots.push_back(it->second.slot);
  ctx->nodes.erase(it);
  auto ci = ctx->clouds.find(node_id);
  if (ci != ctx->clouds.end()) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ci->second.d) (void)hipFree(ci->second.d);
    if (ci->second.d_samples) (void)hipFree(ci->second.d_samples);
    ctx->clouds.erase(ci);
  }
  return RGBDFE_OK;
}

int rgbdfe_node_count(rgbdfe_ctx* ctx, int32_t node_id) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  auto it = ctx->nodes.find(node_id);
  if (it == ctx->nodes.end()) return RGBDFE_ERR_UNKNOWN_NODE;
  return (int)it->second.n;
}


}  // namespace impl
