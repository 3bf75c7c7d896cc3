// api_pairs.hip -- the pair op: ORB / SIFT / FLANN batches, synchronous, ticketed and host-output forms
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include "rgbdfe_host.h"

namespace impl {

// out_stride (in records): the multi-device group hands every device the interleaved positions of its shard
int rgbdfe_match_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                           int32_t n_pairs, rgbdfe_match_result* out, int64_t out_stride) {
  if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!query_ids || !train_ids || !out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  // A large request is cut into pieces that alternate between the context's internal streams, so that the Hamming
  // kernel of piece k+1 fills the SIMDs the RANSAC tail of piece k leaves idle (as bench.py does across steps).
  // Results are downloaded into a pinned staging buffer -- a download into the caller's pageable memory would block
  // this thread until its stream has drained and serialise the pieces -- and copied out at the end.  Results do not
  // depend on the batch composition.
  const int32_t cap = ctx->cfg.max_pairs_per_batch;
  if (!ctx->h_results && n_pairs > 0) {
    if (hipHostMalloc((void**)&ctx->h_results, sizeof(rgbdfe_match_result) * (size_t)cap, hipHostMallocDefault) != hipSuccess)
      return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "pinned result staging allocation failed");
  }
  for (int32_t super = 0; super < n_pairs; super += cap) {
    const int32_t m = (n_pairs - super) < cap ? (n_pairs - super) : cap;
    // one piece per lane: a RANSAC launch lasts at least as long as its slowest pair (~7 ms), so finer pieces that
    // queue behind each other on a lane only add up (measured: 4 pieces 19.8 ms, 2 pieces 17.2 ms per 4000 pairs)
    const int32_t parts = m >= 512 ? rgbdfe_ctx::kLanes : 1;
    const int32_t piece = (m + parts - 1) / parts;
    for (int32_t off = 0; off < m; off += piece) {
      const int32_t n = (m - off) < piece ? (m - off) : piece;
      int li = 0;
      int rc = enqueue_pairs(ctx, query_ids + super + off, train_ids + super + off, n, nullptr, nullptr, nullptr, &li);
      if (rc != RGBDFE_OK) return rc;
      // stream order makes the lane's device staging buffer safe to reuse two pieces later
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results + off, ctx->lanes[li].d_results, sizeof(rgbdfe_match_result) * (size_t)n,
                                  hipMemcpyDeviceToHost, ctx->lanes[li].stream));
    }
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
    if (out_stride == 1) memcpy(out + super, ctx->h_results, sizeof(rgbdfe_match_result) * (size_t)m);
    else
      for (int32_t i = 0; i < m; ++i) out[(int64_t)(super + i) * out_stride] = ctx->h_results[i];
  }
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  if (ctx->profiling) drain_pending(ctx);
  return RGBDFE_OK;
}

int rgbdfe_match_node_pairs(rgbdfe_ctx* ctx, int32_t new_node_id, const int32_t* candidate_ids,
                            int32_t n_pairs, rgbdfe_match_result* out) {
  if (!ctx || n_pairs < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
  std::vector<int32_t> q((size_t)n_pairs, new_node_id);
  return impl::rgbdfe_match_pair_list(ctx, q.data(), candidate_ids, n_pairs, out);
}

int rgbdfe_match_pair_list_device(rgbdfe_ctx* ctx, const int32_t* query_ids,
                                  const int32_t* train_ids, int32_t n_pairs, void* d_out,
                                  void* stream) {
  if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!query_ids || !train_ids || !d_out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  // in-order semantics on the caller's stream: the batch starts after everything already
  // enqueued on `stream`, and `stream` continues after the batch.
  hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_in, s));
  int64_t ticket = 0;
  int rc = enqueue_pairs(ctx, query_ids, train_ids, n_pairs, (rgbdfe_match_result*)d_out, ctx->ev_in,
                         &ticket, nullptr);
  if (rc != RGBDFE_OK) return rc;
  return wait_ticket(ctx, ticket, s);
}

int rgbdfe_submit_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                            int32_t n_pairs, void* d_out, int64_t* ticket) {
  if (!ctx || n_pairs < 0 || !ticket || (n_pairs > 0 && (!query_ids || !train_ids || !d_out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad submit arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  return enqueue_pairs(ctx, query_ids, train_ids, n_pairs, (rgbdfe_match_result*)d_out, nullptr, ticket,
                       nullptr);
}

int rgbdfe_wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, void* stream) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  return wait_ticket(ctx, ticket, (hipStream_t)stream);
}

// is [p, p + bytes) host memory the device can copy into asynchronously (hipHostMalloc / hipHostRegister)?  Both ends of
// the range are asked about: a registration shorter than the buffer must not make an async DMA write pageable memory.
static bool is_pinned_host(const void* p, size_t bytes) {
  if (bytes == 0) return false;
  for (const void* q : {p, (const void*)((const uint8_t*)p + bytes - 1)}) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
  }
  return true;
}

// The asynchronous form of rgbdfe_match_pair_list: results in HOST memory, the download of batch k behind batch k on its lane
// while batch k+1 computes on the other lane (the reference's consumer reads the results on the host:
// graph_manager.cpp:409-419, 554-560).
int rgbdfe_submit_pair_list_host(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs,
                                 void* out, size_t out_bytes, int payload, int64_t* ticket) {
  if (!ctx || n_pairs < 0 || !ticket || (n_pairs > 0 && (!query_ids || !train_ids || !out)) ||
      (payload != RGBDFE_HOST_RECORDS && payload != RGBDFE_HOST_INLIERS))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad host submit arguments");
  const size_t rec = sizeof(rgbdfe_match_result), hdr = sizeof(rgbdfe_inlier_header);
  const size_t need = payload == RGBDFE_HOST_RECORDS ? rec * (size_t)n_pairs : hdr * (size_t)n_pairs;  // (+ the list block)
  if (out_bytes < need) return fail(ctx, RGBDFE_ERR_CAPACITY, "host output buffer too small");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  const int li = (int)(ctx->next_ticket % rgbdfe_ctx::kLanes);   // the lane enqueue_pairs will take
  rgbdfe_ctx::HostJob& job = ctx->host_jobs[li];
  if (job.pending) return fail(ctx, RGBDFE_ERR_CAPACITY, "rgbdfe_submit_pair_list_host: wait for an earlier ticket first (one job per lane)");
  const size_t cap = (size_t)ctx->cfg.max_pairs_per_batch;
  const size_t stream_cap = cap * (hdr + 4 * (size_t)RGBDFE_MAX_MATCHES);
  if (!job.copied) HIP_TRY(ctx, hipEventCreateWithFlags(&job.copied, hipEventDisableTiming));
  if (!ctx->h_stage[li] &&
      hipHostMalloc((void**)&ctx->h_stage[li], stream_cap > rec * cap ? stream_cap : rec * cap, hipHostMallocDefault) != hipSuccess)
    return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "pinned result staging allocation failed");
  if (payload == RGBDFE_HOST_INLIERS && !ctx->d_inl_stream[li]) {
    if (hipMalloc((void**)&ctx->d_inl_stream[li], stream_cap) != hipSuccess ||
        hipMalloc((void**)&ctx->d_inl_total[li], sizeof(int32_t)) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_inl_total[li], sizeof(int32_t), hipHostMallocDefault) != hipSuccess)
      return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "inlier stream buffers");
  }
  int lane_used = 0;
  const int rc = enqueue_pairs(ctx, query_ids, train_ids, n_pairs, nullptr, nullptr, ticket, &lane_used);
  if (rc != RGBDFE_OK) return rc;
  if (lane_used != li) return fail(ctx, RGBDFE_ERR_INTERNAL, "host submit: lane bookkeeping out of step");
  hipStream_t st = ctx->lanes[li].stream;
  // (the whole caller buffer: the inlier stream's list block lands behind the headers, its length is known only later)
  job.direct = n_pairs > 0 && is_pinned_host(out, out_bytes);
  if (payload == RGBDFE_HOST_RECORDS) {
    if (n_pairs > 0)
      HIP_TRY(ctx, hipMemcpyAsync(job.direct ? out : (void*)ctx->h_stage[li], ctx->lanes[li].d_results, rec * (size_t)n_pairs,
                                  hipMemcpyDeviceToHost, st));
  } else if (n_pairs > 0) {
    launch_pack_inliers(ctx->lanes[li].d_results, (uint32_t)n_pairs, (uint32_t)n_pairs, ctx->d_inl_stream[li],
                        ctx->d_inl_total[li], st);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_inl_total[li], ctx->d_inl_total[li], sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(job.direct ? out : (void*)ctx->h_stage[li], ctx->d_inl_stream[li], hdr * (size_t)n_pairs,
                                hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipEventRecord(job.copied, st));
  job.pending = true;
  job.waiting = false;
  job.payload = payload;
  job.ticket = *ticket;
  job.n = n_pairs;
  job.out = out;
  job.out_bytes = out_bytes;
  return RGBDFE_OK;
}

// replace: collect the job into [new_out, new_out + new_bytes) instead of the buffer named at submit (new_out == nullptr:
// drop the job).  The headers of a job that went straight into a pinned buffer are fetched again, through the stage.
static int wait_host_impl(rgbdfe_ctx* ctx, int64_t ticket, bool replace, void* new_out, size_t new_bytes, int64_t* bytes_written) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  rgbdfe_ctx::HostJob job;
  int li = -1;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    for (int k = 0; k < rgbdfe_ctx::kLanes; ++k)
      if (ctx->host_jobs[k].pending && ctx->host_jobs[k].ticket == ticket) li = k;
    if (li < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_wait_host: no host job with this ticket");
    // one waiter per job: a second one would copy out of a staging buffer the lane's next job may already be filling
    if (ctx->host_jobs[li].waiting) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_wait_host: another thread is waiting for this ticket");
    if (replace && new_out == nullptr) {   // the caller gives the job up
      ctx->host_jobs[li].pending = false;
      if (bytes_written) *bytes_written = 0;
      return RGBDFE_OK;
    }
    ctx->host_jobs[li].waiting = true;
    job = ctx->host_jobs[li];
  }
  bool refetch = false;
  if (replace) {
    refetch = job.direct;
    job.direct = false;
    job.out = new_out;
    job.out_bytes = new_bytes;
  }
  // (the context is not locked while this thread waits and copies: another thread may submit the next batch meanwhile)
  hipError_t e = hipSetDevice(ctx->cfg.device_id);
  if (e == hipSuccess) e = hipEventSynchronize(job.copied);
  const size_t rec = sizeof(rgbdfe_match_result), hdr = sizeof(rgbdfe_inlier_header);
  size_t written = 0;
  int rc = RGBDFE_OK;
  if (e == hipSuccess && job.n > 0 && refetch) {   // (the first download went into the buffer that is being replaced)
    e = hipMemcpyAsync(ctx->h_stage[li], job.payload == RGBDFE_HOST_RECORDS ? (const void*)ctx->lanes[li].d_results : (const void*)ctx->d_inl_stream[li],
                       (job.payload == RGBDFE_HOST_RECORDS ? rec : hdr) * (size_t)job.n, hipMemcpyDeviceToHost, ctx->lanes[li].stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->lanes[li].stream);
  }
  if (e == hipSuccess && job.n > 0) {
    if (job.payload == RGBDFE_HOST_RECORDS) {
      written = rec * (size_t)job.n;
      if (written > job.out_bytes) rc = RGBDFE_ERR_CAPACITY;
      else if (!job.direct) memcpy(job.out, ctx->h_stage[li], written);
    } else {
      const size_t list_bytes = 4 * (size_t)(*ctx->h_inl_total[li] > 0 ? *ctx->h_inl_total[li] : 0);
      written = hdr * (size_t)job.n + list_bytes;
      if (written > job.out_bytes) {
        rc = RGBDFE_ERR_CAPACITY;
      } else {
        if (!job.direct) memcpy(job.out, ctx->h_stage[li], hdr * (size_t)job.n);
        // the list block: its length is known only now (a second, short download on the lane's stream)
        if (list_bytes > 0) {
          uint8_t* dst = job.direct ? (uint8_t*)job.out + hdr * (size_t)job.n : ctx->h_stage[li] + hdr * (size_t)job.n;
          e = hipMemcpyAsync(dst, ctx->d_inl_stream[li] + hdr * (size_t)job.n, list_bytes, hipMemcpyDeviceToHost, ctx->lanes[li].stream);
          if (e == hipSuccess) e = hipStreamSynchronize(ctx->lanes[li].stream);
          if (e == hipSuccess && !job.direct) memcpy((uint8_t*)job.out + hdr * (size_t)job.n, dst, list_bytes);
        }
      }
    }
  }
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->host_jobs[li].waiting = false;
    // a payload that did not fit: the job stays (its results are still on the device) -- *bytes_written is what the buffer
    // has to hold; rgbdfe_wait_host_into() collects it with a larger one, or drops it
    if (!(rc == RGBDFE_ERR_CAPACITY && e == hipSuccess)) ctx->host_jobs[li].pending = false;
  }
  if (bytes_written) *bytes_written = (int64_t)written;
  if (e != hipSuccess) return fail(ctx, RGBDFE_ERR_HIP, std::string("rgbdfe_wait_host: ") + hipGetErrorString(e));
  if (rc != RGBDFE_OK) return fail(ctx, rc, "rgbdfe_wait_host: the payload does not fit the caller's buffer (the job stays: rgbdfe_wait_host_into)");
  return RGBDFE_OK;
}

int rgbdfe_wait_host(rgbdfe_ctx* ctx, int64_t ticket, int64_t* bytes_written) {
  return wait_host_impl(ctx, ticket, false, nullptr, 0, bytes_written);
}
int rgbdfe_wait_host_into(rgbdfe_ctx* ctx, int64_t ticket, void* out, size_t out_bytes, int64_t* bytes_written) {
  return wait_host_impl(ctx, ticket, true, out, out_bytes, bytes_written);
}

int rgbdfe_synchronize(rgbdfe_ctx* ctx) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  drain_pending(ctx);  // synchronises every lane
  return RGBDFE_OK;
}


static int ensure_sift(rgbdfe_ctx* ctx) {
  if (ctx->sift_ready) return RGBDFE_OK;
  const size_t rows = (size_t)ctx->cfg.max_nodes * (size_t)ctx->cfg.max_keypoints + 16;
  const size_t np = (size_t)ctx->cfg.max_pairs_per_batch, mk = (size_t)ctx->cfg.max_keypoints;
  // + 384 rows: sift_top2_fast_kernel prefetches whole 128-row tiles without clamping the row (up to one tile past the
  // node's last one); what lies beyond a node's rows is finite (zeros or older quantised values) and masked
  const size_t bf16_rows = rows + 384;
  if (hipMalloc((void**)&ctx->d_sift_bf16, bf16_rows * 128 * 2) != hipSuccess ||
      hipMalloc((void**)&ctx->d_sift_f32, rows * 128 * 4) != hipSuccess)
    return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "SIFT node slabs");
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_sift_bf16, 0, bf16_rows * 128 * 2, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_sift_f32, 0, rows * 128 * 4, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (this stream only: see rgbdfe_create)
  for (auto& ln : ctx->lanes) {
    if (hipMalloc((void**)&ln.d_row_part, np * mk * 3 * 4) != hipSuccess ||
        hipMalloc((void**)&ln.d_col_part, np * mk * 3 * 4) != hipSuccess ||
        hipMalloc((void**)&ln.d_col_blocks, np * sift_col_block_bytes_per_pair()) != hipSuccess ||
        hipMalloc((void**)&ln.d_sm_q, np * mk * 2) != hipSuccess ||
        hipMalloc((void**)&ln.d_sm_t, np * mk * 2) != hipSuccess ||
        hipMalloc((void**)&ln.d_sm_d, np * mk * 4) != hipSuccess ||
        hipMalloc((void**)&ln.d_sm_n, np * 4) != hipSuccess ||
        hipMalloc((void**)&ln.d_all_dist, np * RGBDFE_MAX_MATCHES * 4) != hipSuccess)
      return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "SIFT batch scratch");
  }
  ctx->sift_ready = true;
  return RGBDFE_OK;
}

int rgbdfe_upload_sift_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc128,
                            const float* xyz1, int32_t n) {
  if (!ctx || n < 0 || (n > 0 && (!desc128 || !xyz1))) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments");
  if (n > ctx->cfg.max_keypoints) return fail(ctx, RGBDFE_ERR_CAPACITY, "node has more rows than max_keypoints");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  int rc = ensure_sift(ctx);
  if (rc != RGBDFE_OK) return rc;
  uint32_t slot;
  auto it = ctx->nodes.find(node_id);
  if (it != ctx->nodes.end()) {
    slot = it->second.slot;
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  } else {
    if (ctx->free_slots.empty()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
    slot = ctx->free_slots.back();
    ctx->free_slots.pop_back();
  }
  const size_t row0 = (size_t)slot * (size_t)ctx->cfg.max_keypoints;
  if (n > 0) {
    float* df = ctx->d_sift_f32 + row0 * 128;
    HIP_TRY(ctx, hipMemcpyAsync(df, desc128, (size_t)n * 128 * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_sift_quantise(df, ctx->d_sift_bf16 + row0 * 128, (size_t)n * 128, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_xyz + row0, xyz1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // quantised squared norms (SiftMatchCU.cpp:96-99's u8 values): decides whether pairs of this node may use the fast keys
  uint32_t flags = 1u;
  for (int32_t r = 0; r < n && flags; ++r) {
    uint64_t sq = 0;
    const float* d = desc128 + (size_t)r * 128;
    for (int k = 0; k < 128; ++k) {
      const float prod = 512 * d[k];
      const unsigned char u = (unsigned char)(int)((double)prod + 0.5);
      sq += (uint64_t)u * u;
    }
    if (sq >= (1ull << 19)) flags = 0u;  // strict: dot <= sqrt(sq1 * sq2) < 2^19
  }
  ctx->nodes[node_id] = NodeEntry{slot, (uint32_t)n, 1u, flags};
  return RGBDFE_OK;
}

// Float descriptors as Node::feature_descriptors_ holds them for the FLANN branch (N x dim CV_32F, node.cpp:610-667;
// SURF 64-d, SIFT 128-d, RootSIFT-normalised when use_root_sift): kind 2, rows zero-padded to 128 floats.
int rgbdfe_upload_float_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc, int32_t dim, const float* xyz1,
                             int32_t n) {
  if (!ctx || n < 0 || dim < 4 || dim > 128 || dim % 4 != 0 || (n > 0 && (!desc || !xyz1)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments (dim must be a multiple of 4 in [4, 128])");
  if (n > ctx->cfg.max_keypoints) return fail(ctx, RGBDFE_ERR_CAPACITY, "node has more rows than max_keypoints");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  int rc = ensure_sift(ctx);
  if (rc != RGBDFE_OK) return rc;
  uint32_t slot;
  auto it = ctx->nodes.find(node_id);
  if (it != ctx->nodes.end()) {
    slot = it->second.slot;
    for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  } else {
    if (ctx->free_slots.empty()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
    slot = ctx->free_slots.back();
    ctx->free_slots.pop_back();
  }
  const size_t row0 = (size_t)slot * (size_t)ctx->cfg.max_keypoints;
  if (n > 0) {
    float* df = ctx->d_sift_f32 + row0 * 128;
    if (dim == 128) {
      HIP_TRY(ctx, hipMemcpyAsync(df, desc, (size_t)n * 128 * 4, hipMemcpyHostToDevice, ctx->stream));
    } else {
      HIP_TRY(ctx, hipMemsetAsync(df, 0, (size_t)n * 128 * 4, ctx->stream));
      HIP_TRY(ctx, hipMemcpy2DAsync(df, 128 * 4, desc, (size_t)dim * 4, (size_t)dim * 4, (size_t)n, hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_xyz + row0, xyz1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->nodes[node_id] = NodeEntry{slot, (uint32_t)n, 2u, 0u};
  return RGBDFE_OK;
}

int rgbdfe_match_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                int32_t n_pairs, rgbdfe_match_result* out, float* out_dist, int64_t out_stride,
                                int matcher, double flann_ratio) {
  if (out_stride != 1 && n_pairs > 0 && out) {  // multi-device shard: dense call, then the interleaved placement
    std::vector<rgbdfe_match_result> tmp((size_t)n_pairs);
    std::vector<float> tmpd(out_dist ? (size_t)n_pairs * RGBDFE_MAX_MATCHES : 0);
    int rc = impl::rgbdfe_match_sift_pair_list(ctx, query_ids, train_ids, n_pairs, tmp.data(), out_dist ? tmpd.data() : nullptr, 1,
                                               matcher, flann_ratio);
    if (rc != RGBDFE_OK) return rc;
    for (int32_t i = 0; i < n_pairs; ++i) {
      out[(int64_t)i * out_stride] = tmp[(size_t)i];
      if (out_dist)
        memcpy(out_dist + (int64_t)i * out_stride * RGBDFE_MAX_MATCHES, tmpd.data() + (size_t)i * RGBDFE_MAX_MATCHES,
               sizeof(float) * RGBDFE_MAX_MATCHES);
    }
    return RGBDFE_OK;
  }
  if (!ctx || n_pairs < 0 || (n_pairs > 0 && (!query_ids || !train_ids || !out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  if (!ctx->sift_ready) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "no SIFT node has been uploaded");
  const int32_t cap = ctx->cfg.max_pairs_per_batch;
  int chunk = 0;
  for (int32_t off = 0; off < n_pairs; off += cap, ++chunk) {
    const int32_t n = (n_pairs - off) < cap ? (n_pairs - off) : cap;
    const int li_next = (int)(ctx->next_ticket % rgbdfe_ctx::kLanes);
    if (chunk >= rgbdfe_ctx::kLanes) HIP_TRY(ctx, hipStreamSynchronize(ctx->lanes[li_next].stream));
    int li = 0;
    int rc = enqueue_pairs(ctx, query_ids + off, train_ids + off, n, nullptr, nullptr, nullptr, &li, matcher, nullptr, flann_ratio);
    if (rc != RGBDFE_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out + off, ctx->lanes[li].d_results, sizeof(rgbdfe_match_result) * (size_t)n,
                                hipMemcpyDeviceToHost, ctx->lanes[li].stream));
    if (out_dist)
      HIP_TRY(ctx, hipMemcpyAsync(out_dist + (size_t)off * RGBDFE_MAX_MATCHES, ctx->lanes[li].d_all_dist,
                                  sizeof(float) * RGBDFE_MAX_MATCHES * (size_t)n, hipMemcpyDeviceToHost,
                                  ctx->lanes[li].stream));
  }
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  if (ctx->profiling) drain_pending(ctx);
  return RGBDFE_OK;
}

int rgbdfe_submit_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                                 int32_t n_pairs, void* d_out, void* d_out_dist, int64_t* ticket) {
  if (!ctx || n_pairs < 0 || !ticket || (n_pairs > 0 && (!query_ids || !train_ids || !d_out)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad submit arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  if (!ctx->sift_ready) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "no SIFT node has been uploaded");
  return enqueue_pairs(ctx, query_ids, train_ids, n_pairs, (rgbdfe_match_result*)d_out, nullptr, ticket,
                       nullptr, 1, (float*)d_out_dist);
}

int rgbdfe_sift_match_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* match_q,
                            int32_t* match_t, float* match_dist, int32_t* n_matches) {
  if (!ctx || !match_q || !match_t || !match_dist || !n_matches) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  *n_matches = 0;
  auto q = ctx->nodes.find(query_id);
  auto t = ctx->nodes.find(train_id);
  if (q == ctx->nodes.end() || t == ctx->nodes.end()) return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "node not resident");
  if (q->second.kind != 1u || t->second.kind != 1u) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "not SIFT nodes");
  for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  for (auto& sl : ctx->ring) sl.pending = false;
  rgbdfe_ctx::Slot& slot = ctx->ring[0];
  rgbdfe_ctx::Lane& lane = ctx->lanes[0];
  PairWork& w = slot.h_work[0];
  w.q_slot = q->second.slot; w.t_slot = t->second.slot;
  w.nq = q->second.n; w.nt = t->second.n;
  w.uid = pair_uid(query_id, train_id); w.qid = query_id; w.tid = train_id;
  w.pad = sift_fast_keys(ctx, q->second, t->second);
  const uint32_t mk = (uint32_t)ctx->cfg.max_keypoints;
  HIP_TRY(ctx, hipMemcpyAsync(slot.d_work, slot.h_work, sizeof(PairWork), hipMemcpyHostToDevice, lane.stream));
  launch_sift_dot(ctx->d_sift_bf16, slot.d_work, mk, 1u, w.nq, w.nt, w.pad ? 1u : 2u, lane.d_row_part, lane.d_col_part,
                  lane.d_col_blocks, lane.stream);
  launch_sift_finish(ctx->d_sift_f32, slot.d_work, mk, 1u, lane.d_row_part, lane.d_col_part, lane.d_col_blocks, lane.d_sm_q,
                     lane.d_sm_t, lane.d_sm_d, lane.d_sm_n, lane.stream);
  HIP_TRY(ctx, hipGetLastError());
  int32_t n = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&n, lane.d_sm_n, 4, hipMemcpyDeviceToHost, lane.stream));
  HIP_TRY(ctx, hipStreamSynchronize(lane.stream));
  if (n > 0) {
    std::vector<uint16_t> hq(n), ht(n);
    HIP_TRY(ctx, hipMemcpyAsync(hq.data(), lane.d_sm_q, (size_t)n * 2, hipMemcpyDeviceToHost, lane.stream));
    HIP_TRY(ctx, hipMemcpyAsync(ht.data(), lane.d_sm_t, (size_t)n * 2, hipMemcpyDeviceToHost, lane.stream));
    HIP_TRY(ctx, hipMemcpyAsync(match_dist, lane.d_sm_d, (size_t)n * 4, hipMemcpyDeviceToHost, lane.stream));
    HIP_TRY(ctx, hipStreamSynchronize(lane.stream));
    for (int i = 0; i < n; ++i) { match_q[i] = hq[i]; match_t[i] = ht[i]; }
  }
  *n_matches = n;
  return RGBDFE_OK;
}




}  // namespace impl
