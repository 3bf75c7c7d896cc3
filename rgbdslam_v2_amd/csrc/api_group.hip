// api_group.hip -- several devices behind ONE handle (rgbdfe_create_multi): node replication, pair sharding, gathers
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include "rgbdfe_host.h"


namespace rgbdfe_host {

void worker_main(Worker* w) {
  std::unique_lock<std::mutex> lk(w->m);
  for (;;) {
    w->cv.wait(lk, [&] { return w->has_job || w->quit; });
    if (w->quit) return;
    std::function<int()> job = std::move(w->job);
    lk.unlock();
    int rc;
    try {
      rc = job();
    } catch (const std::bad_alloc&) {
      rc = RGBDFE_ERR_OUT_OF_MEMORY;
    } catch (...) {
      rc = RGBDFE_ERR_INTERNAL;
    }
    lk.lock();
    w->rc = rc;
    w->has_job = false;
    w->cv.notify_all();
  }
}

// run fn(i) for every device on that device's host thread; returns the first error
int group_run(rgbdfe_ctx* gctx, const std::function<int(int)>& fn) {
  Group& g = *gctx->group;
  std::lock_guard<std::recursive_mutex> call_lock(g.mu);
  const int G = (int)g.children.size();
  for (int i = 0; i < G; ++i) {
    Worker& w = *g.workers[(size_t)i];
    std::lock_guard<std::mutex> lk(w.m);
    w.job = [&fn, i] { return fn(i); };
    w.has_job = true;
    w.cv.notify_all();
  }
  int first = RGBDFE_OK;
  for (int i = 0; i < G; ++i) {
    Worker& w = *g.workers[(size_t)i];
    std::unique_lock<std::mutex> lk(w.m);
    w.cv.wait(lk, [&] { return !w.has_job; });
    if (w.rc != RGBDFE_OK && first == RGBDFE_OK) {
      first = w.rc;
      std::string msg;
      {
        std::lock_guard<std::mutex> e(g.children[(size_t)i]->err_mu);
        msg = g.children[(size_t)i]->last_error;
      }
      fail(gctx, first, "device " + std::to_string(g.device_ids[(size_t)i]) + ": " + msg);
    }
  }
  return first;
}

void group_destroy(rgbdfe_ctx* gctx) {
  Group* g = gctx->group;
  if (g) {
    for (auto& w : g->workers) {
      if (!w) continue;
      {
        std::lock_guard<std::mutex> lk(w->m);
        w->quit = true;
        w->cv.notify_all();
      }
      if (w->th.joinable()) w->th.join();
    }
    if (g->rccl_ok)
      for (void* c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    for (size_t i = 0; i < g->children.size(); ++i) {
      if (g->children[i]) (void)hipSetDevice(g->device_ids[i]);
      if (i < g->gather_streams.size() && g->gather_streams[i]) (void)hipStreamDestroy(g->gather_streams[i]);
      if (i < g->gather_events.size() && g->gather_events[i]) (void)hipEventDestroy(g->gather_events[i]);
      if (i < g->edge_recs.size()) {
        if (i < g->inl_stream.size() && g->inl_stream[i]) (void)hipFree(g->inl_stream[i]);
      if (g->edge_recs[i]) (void)hipFree(g->edge_recs[i]);
        if (g->edge_idx[i]) (void)hipFree(g->edge_idx[i]);
        if (g->edge_dst[i]) (void)hipFree(g->edge_dst[i]);
        if (g->edge_cnt[i]) (void)hipFree(g->edge_cnt[i]);
        if (g->edge_cnt_host[i]) (void)hipHostFree(g->edge_cnt_host[i]);
      }
      if (g->children[i]) impl::rgbdfe_destroy(g->children[i]);
    }
    delete g;
  }
  delete gctx;
}

int group_create(const rgbdfe_config* cfg, const int32_t* device_ids, int32_t n, rgbdfe_ctx** out) {
  if (!cfg || !device_ids || !out || n < 1 || n > 64) return RGBDFE_ERR_INVALID_ARG;
  *out = nullptr;
  rgbdfe_ctx* gctx = new rgbdfe_ctx();
  gctx->cfg = *cfg;
  gctx->group = new Group();
  Group& g = *gctx->group;
  for (int32_t i = 0; i < n; ++i) {
    rgbdfe_config c = *cfg;
    c.device_id = device_ids[i];
    rgbdfe_ctx* child = nullptr;
    const int rc = impl::rgbdfe_create(&c, &child);
    if (rc != RGBDFE_OK) {
      group_destroy(gctx);
      return rc;
    }
    g.children.push_back(child);
    g.device_ids.push_back(device_ids[i]);
  }
  g.gather_streams.assign((size_t)n, nullptr);
  g.gather_events.assign((size_t)n, nullptr);
  for (int32_t i = 0; i < n; ++i) {
    if (hipSetDevice(device_ids[i]) != hipSuccess ||
        hipStreamCreateWithFlags(&g.gather_streams[(size_t)i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g.gather_events[(size_t)i], hipEventDisableTiming) != hipSuccess) {
      group_destroy(gctx);
      return RGBDFE_ERR_HIP;
    }
  }
  for (int32_t i = 0; i < n; ++i) {
    g.workers.emplace_back(new Worker());
    Worker* w = g.workers.back().get();
    w->th = std::thread(worker_main, w);
  }
  *out = gctx;
  return RGBDFE_OK;
}

// sharded host-output match: device i computes pairs i, i+G, ... and writes them to out[i], out[i+G], ...
// ORB shards that fit one batch are ENQUEUED first (group_submit: by the calling thread device after device up to two
// devices, by the devices' host threads all at once from three on), then collected; everything else goes through the
// per-device worker threads.
int group_match(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, rgbdfe_match_result* out, bool sift,
                float* out_dist) {
  if (n < 0 || (n > 0 && (!q || !t || !out))) return fail(gctx, RGBDFE_ERR_INVALID_ARG, "bad match arguments");
  Group& g = *gctx->group;
  std::lock_guard<std::recursive_mutex> call_lock(g.mu);
  const int G = (int)g.children.size();
  const int32_t per = (n + G - 1) / G;
  if (!sift && n > 0 && per <= gctx->cfg.max_pairs_per_batch) {
    std::vector<std::vector<int32_t>> qs((size_t)G), ts((size_t)G);
    for (int32_t k = 0; k < n; ++k) { qs[(size_t)(k % G)].push_back(q[k]); ts[(size_t)(k % G)].push_back(t[k]); }
    std::vector<int> lane((size_t)G, -1);
    const double t0 = orb_now_us();
    int first = group_submit(gctx, [&](int i) -> int {
      rgbdfe_ctx* c = g.children[(size_t)i];
      const int32_t ni = (int32_t)qs[(size_t)i].size();
      if (ni == 0) return RGBDFE_OK;
      std::lock_guard<std::mutex> lk(c->mu);
      int r = RGBDFE_OK;
      if (hipSetDevice(c->cfg.device_id) != hipSuccess) r = fail(c, RGBDFE_ERR_HIP, "hipSetDevice");
      if (r == RGBDFE_OK && !c->h_results &&
          hipHostMalloc((void**)&c->h_results, sizeof(rgbdfe_match_result) * (size_t)c->cfg.max_pairs_per_batch,
                        hipHostMallocDefault) != hipSuccess)
        r = fail(c, RGBDFE_ERR_OUT_OF_MEMORY, "pinned result staging allocation failed");
      int li = 0;
      if (r == RGBDFE_OK) r = enqueue_pairs(c, qs[(size_t)i].data(), ts[(size_t)i].data(), ni, nullptr, nullptr, nullptr, &li);
      if (r == RGBDFE_OK &&
          hipMemcpyAsync(c->h_results, c->lanes[li].d_results, sizeof(rgbdfe_match_result) * (size_t)ni, hipMemcpyDeviceToHost,
                         c->lanes[li].stream) != hipSuccess)
        r = fail(c, RGBDFE_ERR_HIP, "result read-back");
      if (r == RGBDFE_OK) lane[(size_t)i] = li;
      return r;
    });
    g.last_submit_us = orb_now_us() - t0;
    for (int i = 0; i < G; ++i) {   // collect (also after an error: nothing may stay in flight behind the caller's back)
      if (lane[(size_t)i] < 0) continue;
      rgbdfe_ctx* c = g.children[(size_t)i];
      std::lock_guard<std::mutex> lk(c->mu);
      (void)hipSetDevice(c->cfg.device_id);
      if (hipStreamSynchronize(c->lanes[lane[(size_t)i]].stream) != hipSuccess) {
        if (first == RGBDFE_OK) first = fail(gctx, RGBDFE_ERR_HIP, "device " + std::to_string(g.device_ids[(size_t)i]) + ": synchronize");
        continue;
      }
      const int32_t ni = (int32_t)qs[(size_t)i].size();
      for (int32_t m = 0; m < ni; ++m) out[(size_t)i + (size_t)m * (size_t)G] = c->h_results[m];
    }
    return first;
  }
  return group_run(gctx, [&](int i) -> int {
    std::vector<int32_t> qs, ts;
    for (int32_t k = i; k < n; k += G) { qs.push_back(q[k]); ts.push_back(t[k]); }
    if (qs.empty()) return RGBDFE_OK;
    if (sift)
      return impl::rgbdfe_match_sift_pair_list(g.children[(size_t)i], qs.data(), ts.data(), (int32_t)qs.size(), out + i,
                                               out_dist ? out_dist + (size_t)i * RGBDFE_MAX_MATCHES : nullptr, G);
    return impl::rgbdfe_match_pair_list(g.children[(size_t)i], qs.data(), ts.data(), (int32_t)qs.size(), out + i, G);
  });
}

// The enqueues of a sharded batch, fn(i) per device.  From three devices on every device's enqueues run on that device's own
// host thread, all at once: measured with one GPU listed eight times (tools/bench_group_submit.py, ROCm 7.0) the calling
// thread alone takes 48 - 57 us per device, 380 - 450 us for eight, the eight threads together 142 - 146 us (round 2's runtime
// serialised concurrent enqueues on its locks; this one does not).  One or two devices: the calling thread (the hand-off to
// a thread costs what a device's enqueues cost).  RGBDFE_GROUP_SUBMIT=serial / threads forces one or the other.
int group_submit(rgbdfe_ctx* gctx, const std::function<int(int)>& fn) {
  static const char* mode = getenv("RGBDFE_GROUP_SUBMIT");
  const bool threads = mode ? std::string(mode) == "threads" : gctx->group->children.size() >= 3;
  return threads ? group_run(gctx, fn) : group_each(gctx, fn);
}

// run fn(i) for every device on THIS thread, device after device: for work that only enqueues (returns the first error)
int group_each(rgbdfe_ctx* gctx, const std::function<int(int)>& fn) {
  Group& g = *gctx->group;
  int first = RGBDFE_OK;
  for (int i = 0; i < (int)g.children.size(); ++i) {
    const int r = fn(i);
    if (r != RGBDFE_OK && first == RGBDFE_OK) {
      first = r;
      std::string msg; { std::lock_guard<std::mutex> e(g.children[(size_t)i]->err_mu); msg = g.children[(size_t)i]->last_error; }
      fail(gctx, first, "device " + std::to_string(g.device_ids[(size_t)i]) + ": " + msg);
    }
  }
  return first;
}

bool group_setup_rccl(rgbdfe_ctx* gctx) {
  Group& g = *gctx->group;
  if (g.rccl_tried) return g.rccl_ok;
  g.rccl_tried = true;
  const char* force = getenv("RGBDFE_GATHER");
  if (force && std::string(force) == "p2p") return false;
  std::vector<int> sorted = g.device_ids;
  std::sort(sorted.begin(), sorted.end());
  if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return false;  // RCCL: one rank per device
  if (!g.rccl.load()) return false;
  g.comms.assign(g.device_ids.size(), nullptr);
  if (g.rccl.CommInitAll(g.comms.data(), (int)g.device_ids.size(), g.device_ids.data()) != 0) {
    g.comms.clear();
    return false;
  }
  g.rccl_ok = true;
  return true;
}

// per-device scratch of `per` full records (+ index / scan buffers): the edges-only and the compact gathers stage there
int group_ensure_edge_scratch(rgbdfe_ctx* gctx, int32_t per) {
  Group& g = *gctx->group;
  const int G = (int)g.children.size();
  const size_t rec = sizeof(rgbdfe_match_result);
  if (g.edge_cap >= per) return RGBDFE_OK;
  g.edge_recs.resize((size_t)G, nullptr); g.edge_idx.resize((size_t)G, nullptr); g.edge_dst.resize((size_t)G, nullptr);
  g.edge_cnt.resize((size_t)G, nullptr); g.edge_cnt_host.resize((size_t)G, nullptr);
  for (int i = 0; i < G; ++i) {
    HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
    if (g.edge_recs[(size_t)i]) { (void)hipFree(g.edge_recs[(size_t)i]); (void)hipFree(g.edge_idx[(size_t)i]); (void)hipFree(g.edge_dst[(size_t)i]); }
    g.edge_recs[(size_t)i] = nullptr; g.edge_idx[(size_t)i] = nullptr; g.edge_dst[(size_t)i] = nullptr;
    g.edge_cap = 0;
    HIP_TRY(gctx, hipMalloc((void**)&g.edge_recs[(size_t)i], rec * (size_t)per));
    HIP_TRY(gctx, hipMalloc((void**)&g.edge_idx[(size_t)i], sizeof(int32_t) * (size_t)per));
    HIP_TRY(gctx, hipMalloc((void**)&g.edge_dst[(size_t)i], sizeof(int32_t) * (size_t)per));
    if (g.inl_stream.size() < (size_t)G) g.inl_stream.resize((size_t)G, nullptr);
    if (g.inl_stream[(size_t)i]) { (void)hipFree(g.inl_stream[(size_t)i]); g.inl_stream[(size_t)i] = nullptr; }   // (allocated on first use)
    if (!g.edge_cnt[(size_t)i]) {
      HIP_TRY(gctx, hipMalloc((void**)&g.edge_cnt[(size_t)i], sizeof(int32_t)));
      HIP_TRY(gctx, hipHostMalloc((void**)&g.edge_cnt_host[(size_t)i], sizeof(int32_t), hipHostMallocDefault));
    }
  }
  g.edge_cap = per;
  return RGBDFE_OK;
}

// Results of all pairs on every device.  d_out[i]: device-i buffer of G * per records, per = ceil(n / G);
// pair k ends up at [(k % G) * per + k / G] of every buffer; unused tail records are filled with 0xFF (ids -1).
// compact: d_out holds rgbdfe_compact_result (144 B) instead of full records; the shard is computed into the device's
// record scratch and packed into its segment.
int group_match_allgather(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out,
                          int32_t* records_per_device, bool compact) {
  if (n < 0 || !d_out || (n > 0 && (!q || !t))) return fail(gctx, RGBDFE_ERR_INVALID_ARG, "bad allgather arguments");
  Group& g = *gctx->group;
  std::lock_guard<std::recursive_mutex> call_lock(g.mu);
  const int G = (int)g.children.size();
  const int32_t per = (n + G - 1) / G;
  if (records_per_device) *records_per_device = per;
  if (per == 0) return RGBDFE_OK;
  for (int i = 0; i < G; ++i)
    if (!d_out[i]) return fail(gctx, RGBDFE_ERR_INVALID_ARG, "allgather: a device buffer is NULL");
  if (per > gctx->cfg.max_pairs_per_batch)
    return fail(gctx, RGBDFE_ERR_CAPACITY, "allgather: the shard of a device exceeds max_pairs_per_batch");
  const size_t rec = compact ? sizeof(rgbdfe_compact_result) : sizeof(rgbdfe_match_result);
  if (compact) { const int rce = group_ensure_edge_scratch(gctx, per); if (rce != RGBDFE_OK) return rce; }
  // 1. every device computes its shard into its own segment of its own buffer (group_submit)
  const double t_sub0 = orb_now_us();
  int rc = group_submit(gctx, [&](int i) -> int {
    rgbdfe_ctx* c = g.children[(size_t)i];
    std::vector<int32_t> qs, ts;
    for (int32_t k = i; k < n; k += G) { qs.push_back(q[k]); ts.push_back(t[k]); }
    char* seg_bytes = (char*)d_out[i] + (size_t)i * per * rec;
    rgbdfe_match_result* seg = compact ? g.edge_recs[(size_t)i] : (rgbdfe_match_result*)seg_bytes;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      HIP_TRY(c, hipSetDevice(c->cfg.device_id));
      HIP_TRY(c, hipMemsetAsync(seg_bytes, 0xFF, rec * (size_t)per, g.gather_streams[(size_t)i]));
      HIP_TRY(c, hipEventRecord(g.gather_events[(size_t)i], g.gather_streams[(size_t)i]));
    }
    int64_t ticket = 0;
    int r = RGBDFE_OK;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      r = enqueue_pairs(c, qs.data(), ts.data(), (int32_t)qs.size(), seg, g.gather_events[(size_t)i], &ticket, nullptr);
      if (r == RGBDFE_OK) r = wait_ticket(c, ticket, g.gather_streams[(size_t)i]);
      if (r == RGBDFE_OK && compact) {
        launch_compact_pack(seg, (uint32_t)qs.size(), (rgbdfe_compact_result*)seg_bytes, g.gather_streams[(size_t)i]);
        if (hipGetLastError() != hipSuccess) r = fail(c, RGBDFE_ERR_HIP, "compact_pack_kernel launch");
      }
    }
    return r;
  });
  g.last_submit_us = orb_now_us() - t_sub0;
  if (rc != RGBDFE_OK) {   // nothing may stay in flight behind the caller's back
    for (int i = 0; i < G; ++i) { (void)hipSetDevice(g.device_ids[(size_t)i]); (void)hipStreamSynchronize(g.gather_streams[(size_t)i]); }
    return rc;
  }
  // 2. the exchange
  if (G == 1 && !group_setup_rccl(gctx)) {
    g.transport = "none (one device)";
    HIP_TRY(gctx, hipSetDevice(g.device_ids[0]));
    HIP_TRY(gctx, hipStreamSynchronize(g.gather_streams[0]));
    return RGBDFE_OK;
  }
  if (group_setup_rccl(gctx)) {
    g.transport = "rccl";
    // one thread issues the grouped collective: ncclGroupStart/End makes the per-device calls one operation
    if (g.rccl.GroupStart() != 0) return fail(gctx, RGBDFE_ERR_HIP, "ncclGroupStart failed");
    int nrc = 0;
    for (int i = 0; i < G && nrc == 0; ++i) {
      const char* base = (const char*)d_out[i];
      nrc = g.rccl.AllGather(base + (size_t)i * per * rec, d_out[i], (size_t)per * rec, kNcclChar, g.comms[(size_t)i],
                             g.gather_streams[(size_t)i]);
    }
    const int erc = g.rccl.GroupEnd();
    if (nrc != 0 || erc != 0)
      return fail(gctx, RGBDFE_ERR_HIP, std::string("ncclAllGather: ") +
                                            (g.rccl.GetErrorString ? g.rccl.GetErrorString(nrc ? nrc : erc) : "error"));
  } else {
    g.transport = "p2p";
    // every device pushes its segment into every other buffer once its own batch has finished
    for (int i = 0; i < G; ++i) {
      HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
      const char* src = (const char*)d_out[i] + (size_t)i * per * rec;
      for (int j = 0; j < G; ++j) {
        if (j == i || d_out[j] == d_out[i]) continue;
        char* dst = (char*)d_out[j] + (size_t)i * per * rec;
        HIP_TRY(gctx, hipMemcpyPeerAsync(dst, g.device_ids[(size_t)j], src, g.device_ids[(size_t)i], (size_t)per * rec,
                                         g.gather_streams[(size_t)i]));
      }
    }
  }
  for (int i = 0; i < G; ++i) {
    HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
    HIP_TRY(gctx, hipStreamSynchronize(g.gather_streams[(size_t)i]));
  }
  return RGBDFE_OK;
}

// All-gather of the ACCEPTED edges only (an all-pairs loop-closure sweep rejects most pairs and their
// records need not travel): every device compacts its shard (stable: shard order), the host learns the counts, the
// exchange moves `stride` = the largest count records per device instead of ceil(n / G).  On return d_out[j] holds, for
// every device i, its counts[i] accepted records at [i * stride, i * stride + counts[i]) and d_index[j] (optional) their
// positions in the caller's pair list.  Buffers are sized for the worst case: G * ceil(n / G) records / indices.
int group_match_allgather_edges(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out,
                                int32_t* const* d_index, int32_t* counts, int32_t* stride_out) {
  if (n < 0 || !d_out || !counts || !stride_out || (n > 0 && (!q || !t)))
    return fail(gctx, RGBDFE_ERR_INVALID_ARG, "bad allgather arguments");
  Group& g = *gctx->group;
  std::lock_guard<std::recursive_mutex> call_lock(g.mu);
  const int G = (int)g.children.size();
  const int32_t per = (n + G - 1) / G;
  *stride_out = 0;
  for (int i = 0; i < G; ++i) counts[i] = 0;
  if (per == 0) return RGBDFE_OK;
  for (int i = 0; i < G; ++i)
    if (!d_out[i] || (d_index && !d_index[i])) return fail(gctx, RGBDFE_ERR_INVALID_ARG, "allgather: a device buffer is NULL");
  if (per > gctx->cfg.max_pairs_per_batch)
    return fail(gctx, RGBDFE_ERR_CAPACITY, "allgather: the shard of a device exceeds max_pairs_per_batch");
  const size_t rec = sizeof(rgbdfe_match_result);
  { const int rce = group_ensure_edge_scratch(gctx, per); if (rce != RGBDFE_OK) return rce; }
  // 1. every device: its shard into its own segment of its own buffer, then the accepted records, compacted, into scratch
  int rc = group_run(gctx, [&](int i) -> int {
    rgbdfe_ctx* c = g.children[(size_t)i];
    std::vector<int32_t> qs, ts;
    for (int32_t k = i; k < n; k += G) { qs.push_back(q[k]); ts.push_back(t[k]); }
    rgbdfe_match_result* seg = (rgbdfe_match_result*)d_out[i] + (size_t)i * per;
    hipStream_t gs = g.gather_streams[(size_t)i];
    int64_t ticket = 0;
    int r = RGBDFE_OK;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      HIP_TRY(c, hipSetDevice(c->cfg.device_id));
      HIP_TRY(c, hipEventRecord(g.gather_events[(size_t)i], gs));
      r = enqueue_pairs(c, qs.data(), ts.data(), (int32_t)qs.size(), seg, g.gather_events[(size_t)i], &ticket, nullptr);
      if (r == RGBDFE_OK) r = wait_ticket(c, ticket, gs);
    }
    if (r != RGBDFE_OK) return r;
    launch_compact_edges(seg, (uint32_t)qs.size(), g.edge_recs[(size_t)i], g.edge_idx[(size_t)i], G, i, g.edge_dst[(size_t)i],
                         g.edge_cnt[(size_t)i], gs);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(g.edge_cnt_host[(size_t)i], g.edge_cnt[(size_t)i], sizeof(int32_t), hipMemcpyDeviceToHost, gs));
    HIP_TRY(c, hipStreamSynchronize(gs));
    return RGBDFE_OK;
  });
  if (rc != RGBDFE_OK) return rc;
  int32_t stride = 0;
  for (int i = 0; i < G; ++i) {
    counts[i] = *g.edge_cnt_host[(size_t)i];
    stride = std::max(stride, counts[i]);
  }
  *stride_out = stride;
  if (stride == 0) { g.transport = "none (no edges)"; return RGBDFE_OK; }
  // 2. the exchange: `stride` records (and indices) per device
  if (group_setup_rccl(gctx)) {
    g.transport = "rccl";
    if (g.rccl.GroupStart() != 0) return fail(gctx, RGBDFE_ERR_HIP, "ncclGroupStart failed");
    int nrc = 0;
    for (int i = 0; i < G && nrc == 0; ++i) {
      nrc = g.rccl.AllGather(g.edge_recs[(size_t)i], d_out[i], (size_t)stride * rec, kNcclChar, g.comms[(size_t)i],
                             g.gather_streams[(size_t)i]);
      if (nrc == 0 && d_index)
        nrc = g.rccl.AllGather(g.edge_idx[(size_t)i], d_index[i], (size_t)stride * sizeof(int32_t), kNcclChar,
                               g.comms[(size_t)i], g.gather_streams[(size_t)i]);
    }
    const int erc = g.rccl.GroupEnd();
    if (nrc != 0 || erc != 0)
      return fail(gctx, RGBDFE_ERR_HIP, std::string("ncclAllGather: ") +
                                            (g.rccl.GetErrorString ? g.rccl.GetErrorString(nrc ? nrc : erc) : "error"));
  } else {
    g.transport = G == 1 ? "none (one device)" : "p2p";
    for (int i = 0; i < G; ++i) {
      HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
      for (int j = 0; j < G; ++j) {
        char* dst = (char*)d_out[j] + (size_t)i * stride * rec;
        HIP_TRY(gctx, hipMemcpyPeerAsync(dst, g.device_ids[(size_t)j], g.edge_recs[(size_t)i], g.device_ids[(size_t)i],
                                         (size_t)counts[i] * rec, g.gather_streams[(size_t)i]));
        if (d_index)
          HIP_TRY(gctx, hipMemcpyPeerAsync(d_index[j] + (size_t)i * stride, g.device_ids[(size_t)j], g.edge_idx[(size_t)i],
                                           g.device_ids[(size_t)i], (size_t)counts[i] * sizeof(int32_t),
                                           g.gather_streams[(size_t)i]));
      }
    }
  }
  for (int i = 0; i < G; ++i) {
    HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
    HIP_TRY(gctx, hipStreamSynchronize(g.gather_streams[(size_t)i]));
  }
  return RGBDFE_OK;
}

// All-gather of the INLIER FORM of the results (include/rgbdfe.h: rgbdfe_inlier_header): what GraphManager reads of a
// MatchingResult -- edge, rmse, counts and the inlier matches' (queryIdx, trainIdx) -- ~260 bytes per pair at configs[1] instead
// of 1744.  Every device packs its shard (per headers + its lists) into scratch and the streams are exchanged at a stride of
// per * 104 + 4 * C bytes per device, C >= the longest list.  On return d_out[j] holds device i's stream at byte offset
// i * stride: pair k of the caller's list = header k / G of device k mod G.
//
// ONE exchange, no host read in front of it: C is fixed BEFORE the devices have counted their lists -- the
// longest list the group has seen so far plus a quarter -- so matching, packing and the collective go onto every device's
// stream back to back, and the host reads the counts once, behind all of it (they are what the call returns).  Only the
// first call of a group, and a call whose lists outgrow what was seen (more than a quarter longer than any before), pay a
// second exchange -- of the streams that are still in scratch -- at the exact size; `inl_exchanges` says which it was
// (rgbdfe_gather_exchanges).
int group_match_allgather_inliers(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out,
                                  int32_t* records_per_device, int32_t* totals, int64_t* stride_bytes) {
  if (n < 0 || !d_out || !totals || !stride_bytes || (n > 0 && (!q || !t)))
    return fail(gctx, RGBDFE_ERR_INVALID_ARG, "bad allgather arguments");
  Group& g = *gctx->group;
  std::lock_guard<std::recursive_mutex> call_lock(g.mu);
  const int G = (int)g.children.size();
  const int32_t per = (n + G - 1) / G;
  if (records_per_device) *records_per_device = per;
  *stride_bytes = 0;
  g.inl_exchanges = 0;
  for (int i = 0; i < G; ++i) totals[i] = 0;
  if (per == 0) return RGBDFE_OK;
  for (int i = 0; i < G; ++i)
    if (!d_out[i]) return fail(gctx, RGBDFE_ERR_INVALID_ARG, "allgather: a device buffer is NULL");
  if (per > gctx->cfg.max_pairs_per_batch)
    return fail(gctx, RGBDFE_ERR_CAPACITY, "allgather: the shard of a device exceeds max_pairs_per_batch");
  { const int rce = group_ensure_edge_scratch(gctx, per); if (rce != RGBDFE_OK) return rce; }
  const size_t hdr_bytes = (size_t)per * sizeof(rgbdfe_inlier_header);
  if (g.inl_stream.size() < (size_t)G) g.inl_stream.resize((size_t)G, nullptr);
  for (int i = 0; i < G; ++i)
    if (!g.inl_stream[(size_t)i]) {
      HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
      HIP_TRY(gctx, hipMalloc((void**)&g.inl_stream[(size_t)i], (size_t)g.edge_cap * (sizeof(rgbdfe_inlier_header) + 4 * RGBDFE_MAX_MATCHES)));
    }
  const size_t worst = (size_t)per * RGBDFE_MAX_MATCHES;          // (the caller's buffers and the scratch hold this much)
  const size_t cap_known = std::min(g.inl_cap_entries, worst);    // 0: nothing seen yet -- the counts are read first
  // 1. every device: its shard into scratch records, then the inlier stream and its length (to pinned memory)
  int rc = group_run(gctx, [&](int i) -> int {
    rgbdfe_ctx* c = g.children[(size_t)i];
    std::vector<int32_t> qs, ts;
    for (int32_t k = i; k < n; k += G) { qs.push_back(q[k]); ts.push_back(t[k]); }
    rgbdfe_match_result* seg = g.edge_recs[(size_t)i];
    hipStream_t gs = g.gather_streams[(size_t)i];
    int64_t ticket = 0;
    int r = RGBDFE_OK;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      HIP_TRY(c, hipSetDevice(c->cfg.device_id));
      HIP_TRY(c, hipEventRecord(g.gather_events[(size_t)i], gs));
      if (!qs.empty()) {
        r = enqueue_pairs(c, qs.data(), ts.data(), (int32_t)qs.size(), seg, g.gather_events[(size_t)i], &ticket, nullptr);
        if (r == RGBDFE_OK) r = wait_ticket(c, ticket, gs);
      }
    }
    if (r != RGBDFE_OK) return r;
    launch_pack_inliers(seg, (uint32_t)qs.size(), (uint32_t)per, g.inl_stream[(size_t)i], g.edge_cnt[(size_t)i], gs);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(g.edge_cnt_host[(size_t)i], g.edge_cnt[(size_t)i], sizeof(int32_t), hipMemcpyDeviceToHost, gs));
    if (cap_known == 0) HIP_TRY(c, hipStreamSynchronize(gs));
    return RGBDFE_OK;
  });
  if (rc != RGBDFE_OK) return rc;
  // the exchange at `entries` list entries per device, enqueued behind the packing on every device's gather stream
  auto exchange = [&](size_t entries) -> int {
    const size_t stride = hdr_bytes + entries * 4;
    ++g.inl_exchanges;
    if (group_setup_rccl(gctx)) {
      g.transport = "rccl";
      if (g.rccl.GroupStart() != 0) return fail(gctx, RGBDFE_ERR_HIP, "ncclGroupStart failed");
      int nrc = 0;
      for (int i = 0; i < G && nrc == 0; ++i)
        nrc = g.rccl.AllGather(g.inl_stream[(size_t)i], d_out[i], stride, kNcclChar, g.comms[(size_t)i], g.gather_streams[(size_t)i]);
      const int erc = g.rccl.GroupEnd();
      if (nrc != 0 || erc != 0)
        return fail(gctx, RGBDFE_ERR_HIP, std::string("ncclAllGather: ") +
                                              (g.rccl.GetErrorString ? g.rccl.GetErrorString(nrc ? nrc : erc) : "error"));
    } else {
      g.transport = G == 1 ? "none (one device)" : "p2p";
      for (int i = 0; i < G; ++i) {
        HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
        for (int j = 0; j < G; ++j)
          HIP_TRY(gctx, hipMemcpyPeerAsync((char*)d_out[j] + (size_t)i * stride, g.device_ids[(size_t)j], g.inl_stream[(size_t)i],
                                           g.device_ids[(size_t)i], stride, g.gather_streams[(size_t)i]));
      }
    }
    for (int i = 0; i < G; ++i) {
      HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
      HIP_TRY(gctx, hipStreamSynchronize(g.gather_streams[(size_t)i]));
    }
    return RGBDFE_OK;
  };
  auto read_totals = [&]() -> size_t {
    int32_t longest = 0;
    for (int i = 0; i < G; ++i) { totals[i] = *g.edge_cnt_host[(size_t)i]; longest = std::max(longest, totals[i]); }
    return (size_t)longest;
  };
  // 2. the exchange: at the capacity known from earlier calls (the counts arrive behind it), or -- nothing known, or a list
  //    has outgrown it -- at the longest list of this call
  size_t entries = cap_known;
  if (cap_known != 0) {
    rc = exchange(cap_known);
    if (rc != RGBDFE_OK) return rc;
  }
  const size_t longest = read_totals();
  if (cap_known == 0 || longest > cap_known) {
    entries = longest;
    rc = exchange(longest);
    if (rc != RGBDFE_OK) return rc;
  }
  *stride_bytes = (int64_t)(hdr_bytes + entries * 4);
  g.inl_cap_entries = std::max(g.inl_cap_entries, longest + longest / 4 + 64);
  return RGBDFE_OK;
}


int group_only_single(rgbdfe_ctx* ctx, const char* what) {
  return fail(ctx, RGBDFE_ERR_INVALID_ARG,
              std::string(what) + " takes device pointers: call it on one device's context (rgbdfe_device_context)");
}

}  // namespace rgbdfe_host

