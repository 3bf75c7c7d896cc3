// api_batches.hip -- the batch machinery behind every pair entry point: RANSAC constants, scratch, events and stage times, the record / replay plan, the Hamming launch, the hipGraph cache, enqueue_pairs / wait_ticket
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include "rgbdfe_host.h"

namespace rgbdfe_host {


int fail(rgbdfe_ctx* ctx, int code, const std::string& msg) {
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->err_mu);
    ctx->last_error = msg;
  }
  return code;
}


void fill_ransac_const(rgbdfe_ctx* ctx) {
  const rgbdfe_params& p = ctx->cfg.params;
  RansacConst& rc = ctx->rc;
  rc.max_matches = p.max_matches;
  rc.min_matches = p.min_matches;
  rc.ransac_iterations = p.ransac_iterations;
  rc.max_dist_m = (float)(double)p.max_dist_for_inliers;       // node.cpp:1105
  rc.sq_max_dist = (double)(rc.max_dist_m * rc.max_dist_m);    // node.cpp:1152 (float product)
  rc.depth_cov = p.depth_cov;
  // misc.cpp:702-709
  const double cam_angle_x = 58.0 / 180.0 * M_PI;
  const double cam_angle_y = 45.0 / 180.0 * M_PI;
  const double cam_resol_x = 640;
  const double cam_resol_y = 480;
  const double sx = 3 * tan(cam_angle_x / cam_resol_x);
  const double sy = 3 * tan(cam_angle_y / cam_resol_y);
  rc.raster_cov_x = sx * sx;
  rc.raster_cov_y = sy * sy;
  rc.seed = p.seed;
  rc.g2o_iterations = (int32_t)p.g2o_iterations;
}

int validate_params(rgbdfe_ctx* ctx, const rgbdfe_params& p) {
  if (p.max_matches < 1 || p.max_matches > RGBDFE_MAX_MATCHES)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "max_matches must be in [1, RGBDFE_MAX_MATCHES]");
  if (p.min_matches < 0 || p.ransac_iterations < 0)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "min_matches / ransac_iterations must be >= 0");
  if (!(p.max_dist_for_inliers > 0.f))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "max_dist_for_inliers must be > 0");
  if (p.g2o_iterations > 1000u) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "g2o_iterations must be <= 1000");
  if (p.g2o_iterations > 0u && !(p.depth_cov > 0.0))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "the g2o refinement needs depth_cov > 0");
  return RGBDFE_OK;
}

int ensure_scratch(rgbdfe_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return RGBDFE_OK;
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  ctx->d_scratch = nullptr;
  ctx->scratch_bytes = 0;
  HIP_TRY(ctx, hipMalloc(&ctx->d_scratch, bytes));
  ctx->scratch_bytes = bytes;
  return RGBDFE_OK;
}

hipEvent_t get_event(rgbdfe_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// fold finished timing records into the totals
void drain_pending(rgbdfe_ctx* ctx) {
  for (auto& ln : ctx->lanes)
    if (ln.stream) (void)hipStreamSynchronize(ln.stream);
  for (auto& p : ctx->pending) {
    auto add = [&](int which, hipEvent_t e0, hipEvent_t e1) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
        ctx->k_ms[which] += ms;
        ctx->k_launches[which]++;
        ctx->k_pairs[which] += p.pairs;
      }
    };
    if (p.sift) {
      add(RGBDFE_KERNEL_SIFT_DOT, p.a, p.b);
      add(RGBDFE_KERNEL_SIFT_FINISH, p.b, p.c);
      add(RGBDFE_KERNEL_RANSAC, p.c, p.d);
      ctx->event_pool.push_back(p.d);
    } else {
      add(RGBDFE_KERNEL_HAMMING, p.a, p.b);
      add(RGBDFE_KERNEL_RANSAC, p.b, p.c);
    }
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
    ctx->event_pool.push_back(p.c);
  }
  ctx->pending.clear();
}

// Decides whether a batch of n pairs takes the record / replay latency path and makes sure the lane's record buffer
// is large enough (falls back to the one-wave-per-pair kernel when it cannot be allocated).
// the error pool of the select+RANSAC launches of a lane: one region per wave of the largest grid
int ensure_ec_pool(rgbdfe_ctx* ctx, rgbdfe_ctx::Lane& lane, size_t regions, hipStream_t stream) {
  if (regions <= lane.ec_regions) return RGBDFE_OK;
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  if (lane.d_ec) (void)hipFree(lane.d_ec);
  lane.d_ec = nullptr;
  lane.ec_regions = 0;
  HIP_TRY(ctx, hipMalloc((void**)&lane.d_ec, regions * select_ransac_ec_region_bytes()));
  lane.ec_regions = regions;
  return RGBDFE_OK;
}
int want_latency_path(rgbdfe_ctx* ctx, rgbdfe_ctx::Lane& lane, int32_t n, hipStream_t stream, bool* use, int* chunk_out,
                      PhasePlan* plan) {
  const size_t need_recs = (size_t)n * (size_t)(ctx->rc.ransac_iterations > 0 ? ctx->rc.ransac_iterations : 0);
  const bool force_phases = ctx->latency_chunk_iters < 0;  // testing aid: the phased schedule for any batch size
  const int chunk_cfg = force_phases ? -ctx->latency_chunk_iters : ctx->latency_chunk_iters;
  // automatic: small batches want many short waves (latency), large ones long waves (a wave refills its 7 slots from
  // its own share of iterations, so longer shares keep the batched rounds fuller); tools/bench_batch_sweep.py
  int chunk = chunk_cfg > 0 ? chunk_cfg : (n <= 64 ? 4 : (n <= 640 ? 7 : (n <= 1280 ? 14 : 28)));
  // every recording wave owns a region of the error pool: keep the largest grid (a phase is at most all iterations)
  // within kMaxEcRegions by recording more iterations per wave
  // (a batch of more than kMaxEcRegions pairs cannot get below one region per pair: it takes the one-wave kernel)
  const int I_all = ctx->rc.ransac_iterations > 0 ? ctx->rc.ransac_iterations : 0;
  const bool too_many_pairs = (size_t)n > kMaxEcRegions;
  while (!too_many_pairs && chunk < I_all && (size_t)n * (size_t)((I_all + chunk - 1) / chunk) > kMaxEcRegions) ++chunk;
  bool latency = !too_many_pairs && n <= ctx->latency_pairs && ctx->rc.ransac_iterations >= 2 * chunk &&
                 need_recs <= ((size_t)1 << 24);  // 1.7 GB of records per lane at most
  *chunk_out = chunk;
  if (latency && need_recs > lane.recs_capacity) {
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (lane.d_recs) (void)hipFree(lane.d_recs);
    lane.d_recs = nullptr;
    lane.recs_capacity = 0;
    if (hipMalloc((void**)&lane.d_recs, need_recs * (sizeof(IterRec) + sizeof(IterSum)) +  // records + summaries
                                            ransac_split_mask_bytes(need_recs, (size_t)ctx->cfg.max_pairs_per_batch)) == hipSuccess)  // + viable-iteration masks
      lane.recs_capacity = need_recs;
    else latency = false;
  }
  if (latency && !lane.d_walk &&
      hipMalloc((void**)&lane.d_walk, sizeof(WalkState) * ((size_t)ctx->cfg.max_pairs_per_batch + 1)) != hipSuccess) {  // + the batch's counters
    lane.d_walk = nullptr;
    latency = false;
  }
  // Up to 256 pairs one phase (full speculation, lowest latency); above, four phases so that recording stops
  // where the reference's bookkeeping stops iterating.
  const int I = ctx->rc.ransac_iterations;
  static const int env_phases = getenv("RGBDFE_PHASES") ? atoi(getenv("RGBDFE_PHASES")) : 0;  // experiments only
  // (round 6, measured and kept as a switch only -- RGBDFE_MID_PLAN: a phased plan's unit is a whole pair whose windows follow each
  // other inside one workgroup; with 257 .. 1280 pairs, fewer than the launch has unit buffers, a HARD pair -- one that needs
  // nearly all of its iterations and passes the pre-screen with most of them, 0.002 z^2 -- is one workgroup's chain of 160
  // refinements while others idle: 512 such pairs take 2.7 ms phased and 1.8 ms with full speculation in shares (= 1).  But an
  // EASY pair, the usual case between neighbouring frames, ends its loop inside the first 14 iterations and speculation records
  // all 200: the front-end sub-record's 2030 pairs, matched as two halves, take 6.3 ms instead of 3.3.  Two windows, [0, 14) and
  // the rest (= 2), help neither.  Default 0: phased from 257 pairs on, as before.)
  static const int mid_plan = getenv("RGBDFE_MID_PLAN") ? atoi(getenv("RGBDFE_MID_PLAN")) : 0;
  const bool mid = n > 256 && n <= 1280 && !force_phases && env_phases == 0;
  if ((n <= 256 && !force_phases) || env_phases == 1 || (mid && mid_plan == 1)) { plan->n_phases = 1; plan->ends[0] = I; }
  else if (mid && mid_plan == 2 && I > 14) { plan->n_phases = 2; plan->ends[0] = 14; plan->ends[1] = I; }
  else if (env_phases == 2) { plan->n_phases = 2; plan->ends[0] = ((I * 7 / 20) / 7) * 7 > 0 ? ((I * 7 / 20) / 7) * 7 : I; plan->ends[1] = I; if (plan->ends[0] >= I) plan->n_phases = 1; }
  else {
    const int cand[4] = {14, ((I * 7 / 20) / 7) * 7, ((I * 14 / 20) / 7) * 7, I};
    int k = 0, last = 0;
    for (int c : cand) { const int e = c > I ? I : c; if (e > last) { plan->ends[k++] = e; last = e; } }
    if (k == 0) plan->ends[k++] = I;  // ransac_iterations == 0: the replay alone writes the results
    plan->n_phases = k;
  }
  *use = latency;
  size_t regions = (size_t)n;
  if (latency) {
    int begin = 0;
    for (int p = 0; p < plan->n_phases; ++p) {
      size_t chunks = (size_t)((plan->ends[p] - begin + chunk - 1) / chunk);
      if (plan->n_phases > 2 && p == 1)  // launch_record_replay: the second phase covers all that is left, two sub-grids
        chunks += (size_t)((I - begin + 63) / 64);
      if ((size_t)n * chunks > regions) regions = (size_t)n * chunks;
      begin = plan->ends[p];
    }
  }
  // recording grids: 8 segments x ceil(n / 8) pairs x shares per pair (one region per launched wave)
  regions = regions / (size_t)(n > 0 ? n : 1) * (((size_t)n + 7) / 8 * 8);
  return ensure_ec_pool(ctx, lane, regions + 8, stream);
}

// The Hamming stage of an ORB batch: the fp4 MFMA kernel by default, the popcount kernel when asked for
// (rgbdfe_set_hamming_mode) or when the row index does not fit the MFMA kernel's 15 key bits.  Same keys either way.
bool hamming_on_mfma(const rgbdfe_ctx* ctx) { return ctx->hamming_mode != 0 && (uint32_t)ctx->cfg.max_keypoints <= 32768u; }

HammingGeometry hamming_geometry(const rgbdfe_ctx* ctx, uint32_t n, uint32_t max_nq, uint32_t max_nt) {
  const uint32_t cap = (uint32_t)ctx->cfg.max_pairs_per_batch;
  return hamming_on_mfma(ctx) ? hamming_mfma_geometry(n, max_nq, max_nt, cap) : hamming_nn_geometry(n, max_nq, max_nt, cap);
}

uint32_t launch_hamming(rgbdfe_ctx* ctx, const PairWork* d_work, uint32_t* d_keys, uint32_t n, HammingGeometry geom,
                        hipStream_t stream) {
  const uint32_t mk = (uint32_t)ctx->cfg.max_keypoints;
  if (hamming_on_mfma(ctx))
    return launch_hamming_mfma(ctx->d_desc4, d_work, d_keys, mk, n, geom, ctx->hamming_mode, stream);
  return launch_hamming_nn(ctx->d_desc, d_work, d_keys, mk, n, geom, stream);
}

uint32_t launch_hamming(rgbdfe_ctx* ctx, const PairWork* d_work, uint32_t* d_keys, uint32_t n, uint32_t max_nq,
                        uint32_t max_nt, hipStream_t stream) {
  return launch_hamming(ctx, d_work, d_keys, n, hamming_geometry(ctx, n, max_nq, max_nt), stream);
}

// Build the PairWork list (host) and enqueue H2D + both kernels on the next lane.
// Results land in d_out (device memory; nullptr = the lane's own staging buffer).
// Returns the batch's ticket.  Caller holds the lock.
// matcher: 0 = ORB (Hamming), 1 = SIFTGPU (u8 dot products on the MFMA), 2 = FLANN branch (exact L2 knn-2 + ratio test)
bool capture_stream_ready(rgbdfe_ctx* ctx) {
  if (ctx->capture_stream) return true;
  if (hipStreamCreateWithFlags(&ctx->capture_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->capture_stream = nullptr; }
  return ctx->capture_stream != nullptr;
}

int enqueue_pairs(rgbdfe_ctx* ctx, const int32_t* qids, const int32_t* tids, int32_t n, rgbdfe_match_result* d_out, hipEvent_t wait_for, int64_t* ticket_out, int* lane_out, int matcher, float* d_out_dist, double flann_ratio) {


  const bool sift = matcher != 0;  // both float matchers feed the (queryIdx, trainIdx, distance) list path
  if (n > ctx->cfg.max_pairs_per_batch)
    return fail(ctx, RGBDFE_ERR_CAPACITY, "n_pairs exceeds max_pairs_per_batch");
  const int64_t ticket = ctx->next_ticket;
  rgbdfe_ctx::Slot& slot = ctx->ring[ticket % rgbdfe_ctx::kRing];
  const int li = (int)(ticket % rgbdfe_ctx::kLanes);
  rgbdfe_ctx::Lane& lane = ctx->lanes[li];
  hipStream_t stream = lane.stream;
  if (slot.pending) {
    HIP_TRY(ctx, hipEventSynchronize(slot.done));
    slot.pending = false;
  }
  uint32_t max_nq = 0, max_nt = 0, sift_kinds = 0;
  for (int32_t i = 0; i < n; ++i) {
    auto q = ctx->nodes.find(qids[i]);
    auto t = ctx->nodes.find(tids[i]);
    if (q == ctx->nodes.end() || t == ctx->nodes.end())
      return fail(ctx, RGBDFE_ERR_UNKNOWN_NODE, "pair references a node that is not resident");
    if (q->second.kind != (uint32_t)matcher || t->second.kind != (uint32_t)matcher)
      return fail(ctx, RGBDFE_ERR_INVALID_ARG, "node descriptor kind does not fit this matcher");
    PairWork& w = slot.h_work[i];
    w.q_slot = q->second.slot;
    w.t_slot = t->second.slot;
    w.nq = q->second.n;
    w.nt = t->second.n;
    w.uid = pair_uid(qids[i], tids[i]);
    w.qid = qids[i];
    w.tid = tids[i];
    // SIFT fast keys: dot products < 2^19 (Cauchy-Schwarz over the two nodes' norms) and at most 32 column tiles
    w.pad = matcher == 1 ? sift_fast_keys(ctx, q->second, t->second) : 0u;
    sift_kinds |= w.pad ? 1u : 2u;
    // the g2o refinement reads the nodes' own feature_locations_2d_ (node.cpp:1222-1268): never a slot's previous occupant
    if (ctx->rc.g2o_iterations > 0 && !(q->second.flags & t->second.flags & kNodeHasKeypoints))
      return fail(ctx, RGBDFE_ERR_INVALID_ARG,
                  "g2o_iterations > 0: a node of the batch has no keypoints (rgbdfe_upload_node_keypoints after every upload of it)");
    if (w.nq > max_nq) max_nq = w.nq;
    if (w.nt > max_nt) max_nt = w.nt;
  }
  if (ctx->rc.g2o_iterations > 0 && !ctx->d_kp2d)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "g2o_iterations > 0 needs the nodes' keypoints (rgbdfe_upload_node_keypoints)");
  // Everything that can fail without leaving work behind (scratch allocations, the schedule) comes first; the ticket
  // is committed only once the batch is on its stream.
  // A batch whose record / replay scratch would be too large (pairs x iterations records, one error-pool region per
  // recording wave) is run as several pieces, one after the other on the same stream with the same scratch: every piece
  // takes the record / replay schedule.  (The one-wave-per-pair kernel runs only when it is asked for,
  // rgbdfe_set_latency_mode(ctx, 0, 0).)
  int32_t piece = n;
  if (n > 0 && ctx->latency_pairs != 0) {
    const size_t I = (size_t)(ctx->rc.ransac_iterations > 0 ? ctx->rc.ransac_iterations : 1);
    size_t fit = (((size_t)1 << 24) / I) < kMaxEcRegions ? (((size_t)1 << 24) / I) : kMaxEcRegions;
    if (fit > 65535) fit = 65535;  // (also the limit of a grid's y extent, which the SIFT kernels index pairs with)
    if (fit < 1) fit = 1;
    if ((size_t)n > fit && (size_t)n <= (size_t)ctx->latency_pairs) piece = (int32_t)fit;
  }
  if (sift && piece > 65535) return fail(ctx, RGBDFE_ERR_CAPACITY, "a one-wave SIFT batch holds at most 65535 pairs");
  bool latency = false;
  int chunk = 7;
  PhasePlan pp{};
  if (n > 0) {
    int rcl = want_latency_path(ctx, lane, piece, stream, &latency, &chunk, &pp);
    if (rcl != RGBDFE_OK) return rcl;
  }
  if (wait_for) HIP_TRY(ctx, hipStreamWaitEvent(stream, wait_for, 0));
  if (ctx->nodes_ready) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->nodes_ready, 0));  // rgbdfe_upload_node_device
  hipError_t launch_err = hipSuccess;
  if (n > 0) {
    if (!d_out) d_out = lane.d_results;
    // hipGraph form of the whole chain: ORB batches that run as one piece, no per-stage timing events, no refinement
    const bool graphable = ctx->use_graphs && !sift && !ctx->profiling && piece >= n && ctx->rc.g2o_iterations == 0;
    // node sizes enter the launches only through the Hamming stage's geometry (query blocks and train splits per pair)
    const HammingGeometry geom = sift ? HammingGeometry{0, 1}
                                      : hamming_geometry(ctx, (uint32_t)(piece < n ? piece : n), max_nq, max_nt);
    rgbdfe_ctx::GraphEntry* ge = nullptr;
    bool capturing = false;
    if (graphable) {
      rgbdfe_ctx::GraphKey key;
      memset(&key, 0, sizeof(key));
      key.n = n; key.qblocks = geom.qblocks; key.tsplit = geom.tsplit; key.slot = (int32_t)(ticket % rgbdfe_ctx::kRing);
      key.latency = latency ? 1 : 0; key.chunk = chunk; key.hamming_mode = ctx->hamming_mode; key.n_phases = pp.n_phases;
      for (int i = 0; i < 4; ++i) key.ends[i] = i < pp.n_phases ? pp.ends[i] : 0;
      memcpy(&key.rc, &ctx->rc, sizeof(RansacConst));
      key.d_out = d_out; key.d_recs = lane.d_recs; key.d_ec = lane.d_ec; key.d_walk = lane.d_walk; key.d_keys = lane.d_keys;
      size_t gi = 0;
      for (; gi < ctx->graphs.size(); ++gi)
        if (memcmp(&ctx->graphs[gi].key, &key, sizeof(key)) == 0) { ge = &ctx->graphs[gi]; break; }
      if (ge) {
        ge->used = ++ctx->graph_clock;
        const hipError_t le = hipGraphLaunch(ge->exec, stream);
        if (le == hipSuccess) {
          ctx->graph_launches++;
          ctx->graph_miss_run = 0;
        } else {  // an executable graph that does not launch is dropped; this batch goes out as plain launches
          (void)hipGetLastError();
          (void)hipGraphExecDestroy(ge->exec); (void)hipGraphDestroy(ge->graph);
          ctx->graphs.erase(ctx->graphs.begin() + (long)gi);
          ctx->graph_launch_failures++;
          ge = nullptr;
        }
      } else {
        // A capture costs more than the ~12 enqueues it replaces: it pays only for shapes that come back.  After
        // kGraphMissRun misses in a row (a caller whose batch shape or output buffer changes every time) batches go out
        // as plain launches, and only every kGraphRetry-th miss is captured, until a shape hits again.
        ctx->graph_misses++;
        const bool try_capture = ctx->graph_miss_run < rgbdfe_ctx::kGraphMissRun ||
                                 ctx->graph_miss_run % rgbdfe_ctx::kGraphRetry == 0;
        ctx->graph_miss_run++;
        if (try_capture && capture_stream_ready(ctx) &&
            hipStreamBeginCapture(ctx->capture_stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
          // Captured on a stream of its own, in relaxed mode: other host threads may be waiting on `stream` for an earlier
          // batch (hipStreamSynchronize on a capturing stream is an error) or be inside the HIP runtime for unrelated work
          // (any capture that is not relaxed makes their hipMalloc / synchronous copies fail for its duration).
          capturing = true;
          if (ctx->graphs.size() >= 48) {  // drop the least recently used shape
            size_t lru = 0;
            for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].used < ctx->graphs[lru].used) lru = i;
            (void)hipGraphExecDestroy(ctx->graphs[lru].exec); (void)hipGraphDestroy(ctx->graphs[lru].graph);
            ctx->graphs.erase(ctx->graphs.begin() + (long)lru);
          }
          ctx->graphs.push_back(rgbdfe_ctx::GraphEntry{key, nullptr, nullptr, ++ctx->graph_clock});
        } else {
          (void)hipGetLastError();  // no capture: plain launches
          ctx->graph_plain_batches++;
        }
      }
    }
    // (at most twice: a capture that another thread's HIP call invalidated -- relaxed mode keeps THEM from failing, but a
    // device-wide synchronisation elsewhere in the process still breaks the capture -- is dropped and the batch issued plainly)
    rgbdfe_ctx::Pending pend{};
    pend.sift = sift;
    for (int attempt = 0; attempt < 2; ++attempt) {
      hipStream_t const ls = capturing ? ctx->capture_stream : stream;   // where this batch's operations are issued
      if (!ge) {
        const hipError_t me = hipMemcpyAsync(slot.d_work, slot.h_work, sizeof(PairWork) * (size_t)n, hipMemcpyHostToDevice, ls);
        if (me != hipSuccess && launch_err == hipSuccess) launch_err = me;
      }
      if (ctx->profiling) {
        pend.a = get_event(ctx);
        pend.b = get_event(ctx);
        pend.c = get_event(ctx);
        if (sift) pend.d = get_event(ctx);
        pend.pairs = n;
        (void)hipEventRecord(pend.a, ls);
      }
      const uint32_t mk = (uint32_t)ctx->cfg.max_keypoints;
      for (int32_t off = 0; off < n && !ge; off += piece) {
        const int32_t m = (n - off) < piece ? (n - off) : piece;
        const bool first = off == 0, last = off + m >= n;
        if (!first) {  // the schedule of a shorter last piece (its scratch needs are covered by the first one's)
          int rcl = want_latency_path(ctx, lane, m, ls, &latency, &chunk, &pp);
          if (rcl != RGBDFE_OK) { launch_err = hipErrorOutOfMemory; break; }
        }
        const PairWork* d_work = slot.d_work + off;
        rgbdfe_match_result* d_res = d_out + off;
        if (!sift) {
          const uint32_t planes = launch_hamming(ctx, d_work, lane.d_keys, (uint32_t)m, first ? geom : hamming_geometry(ctx, (uint32_t)m, max_nq, max_nt), ls);
          if (ctx->profiling && first) (void)hipEventRecord(pend.b, ls);
          if (latency)
            launch_select_ransac_latency(ctx->d_xyz, d_work, lane.d_keys, planes, d_res, mk, (uint32_t)m, ctx->rc,
                                         lane.d_prep, lane.d_recs, lane.d_walk, lane.d_ec, chunk, pp.ends, pp.n_phases, ls);
          else
            launch_select_ransac(ctx->d_xyz, d_work, lane.d_keys, planes, d_res, mk, (uint32_t)m, ctx->rc, lane.d_prep,
                                 lane.d_ec, ls);
          if (ctx->rc.g2o_iterations > 0)
            launch_g2o_refine(d_work, d_res, (uint32_t)m, ctx->rc, lane.d_prep, ctx->d_kp2d, mk, lane.d_ec, ls);
          if (ctx->profiling && last) (void)hipEventRecord(pend.c, ls);
        } else {
          float* d_dist = (d_out_dist ? d_out_dist : lane.d_all_dist) + (size_t)off * RGBDFE_MAX_MATCHES;  // (the lane's buffer holds max_pairs rows)
          if (matcher == 2) {
            launch_l2_knn2(ctx->d_sift_f32, d_work, mk, (uint32_t)m, max_nq, lane.d_row_part, ls);
            if (ctx->profiling && first) (void)hipEventRecord(pend.b, ls);
            launch_l2_ratio(d_work, mk, (uint32_t)m, lane.d_row_part, lane.d_col_part, flann_ratio, lane.d_sm_q,
                            lane.d_sm_t, lane.d_sm_d, lane.d_sm_n, ls);
          } else {
            launch_sift_dot(ctx->d_sift_bf16, d_work, mk, (uint32_t)m, max_nq, max_nt, sift_kinds, lane.d_row_part,
                            lane.d_col_part, lane.d_col_blocks, ls);
            if (ctx->profiling && first) (void)hipEventRecord(pend.b, ls);
            launch_sift_finish(ctx->d_sift_f32, d_work, mk, (uint32_t)m, lane.d_row_part, lane.d_col_part, lane.d_col_blocks,
                               lane.d_sm_q, lane.d_sm_t, lane.d_sm_d, lane.d_sm_n, ls);
          }
          if (ctx->profiling && first) (void)hipEventRecord(pend.c, ls);
          if (latency)
            launch_select_ransac_sift_latency(ctx->d_xyz, d_work, lane.d_sm_q, lane.d_sm_t, lane.d_sm_d, lane.d_sm_n,
                                              d_dist, d_res, mk, (uint32_t)m, ctx->rc,
                                              lane.d_prep, lane.d_recs, lane.d_walk, lane.d_ec, chunk, pp.ends, pp.n_phases, ls);
          else
            launch_select_ransac_sift(ctx->d_xyz, d_work, lane.d_sm_q, lane.d_sm_t, lane.d_sm_d,
                                      lane.d_sm_n, d_dist, d_res, mk,
                                      (uint32_t)m, ctx->rc, lane.d_prep, lane.d_ec, ls);
          if (ctx->rc.g2o_iterations > 0)
            launch_g2o_refine(d_work, d_res, (uint32_t)m, ctx->rc, lane.d_prep, ctx->d_kp2d, mk, lane.d_ec, ls);
          if (ctx->profiling && last) (void)hipEventRecord(pend.d, ls);
        }
      }
      if (capturing) {  // close the capture, keep the executable graph, run it
        rgbdfe_ctx::GraphEntry& e = ctx->graphs.back();
        hipError_t ce = hipStreamEndCapture(ctx->capture_stream, &e.graph);
        if (ce == hipSuccess) ce = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
        if (ce == hipSuccess && launch_err == hipSuccess) {
          ctx->graph_captures++;
          ce = hipGraphLaunch(e.exec, stream);
          ctx->graph_launches++;
          if (ce != hipSuccess) launch_err = ce;
        } else {
          if (e.exec) (void)hipGraphExecDestroy(e.exec);
          if (e.graph) (void)hipGraphDestroy(e.graph);
          ctx->graphs.pop_back();
          (void)hipGetLastError();
          capturing = false;
          launch_err = hipSuccess;
          ctx->graph_capture_failures++;
          continue;   // once more, plain launches on `stream`
        }
      }
      break;
    }
    if (launch_err == hipSuccess) launch_err = hipGetLastError();
    if (ctx->profiling) {
      if (launch_err == hipSuccess) ctx->pending.push_back(pend);
      else {  // a batch that did not launch has no timing record: the events go back to the pool
        ctx->event_pool.push_back(pend.a); ctx->event_pool.push_back(pend.b); ctx->event_pool.push_back(pend.c);
        if (sift) ctx->event_pool.push_back(pend.d);
      }
    }
  }
  // whatever was enqueued is on `stream`: the slot's event covers it whether or not every launch succeeded
  ctx->next_ticket++;
  slot.ticket = ticket;
  slot.failed = launch_err != hipSuccess;
  HIP_TRY(ctx, hipEventRecord(slot.done, stream));
  slot.pending = true;
  if (launch_err != hipSuccess)
    return fail(ctx, RGBDFE_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(launch_err));
  if (ticket_out) *ticket_out = ticket;
  if (lane_out) *lane_out = li;
  return RGBDFE_OK;
}

// Make `stream` (or the host when stream == nullptr) wait for the batch with this ticket.
int wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, hipStream_t stream) {
  if (ticket <= 0 || ticket >= ctx->next_ticket) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "unknown ticket");
  rgbdfe_ctx::Slot& slot = ctx->ring[ticket % rgbdfe_ctx::kRing];
  if (slot.ticket == ticket && slot.failed) return fail(ctx, RGBDFE_ERR_HIP, "the batch with this ticket failed to launch");
  if (slot.ticket != ticket || !slot.pending) return RGBDFE_OK;  // slot reused => that batch has completed
  if (stream) {
    HIP_TRY(ctx, hipStreamWaitEvent(stream, slot.done, 0));
  } else {
    HIP_TRY(ctx, hipEventSynchronize(slot.done));
    slot.pending = false;
  }
  return RGBDFE_OK;
}


}  // namespace rgbdfe_host
