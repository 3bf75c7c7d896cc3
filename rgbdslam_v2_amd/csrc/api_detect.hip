// api_detect.hip -- the feature path: ORB detect / describe (single calls, super-frames), SIFT extraction
// (one of the host-side translation units of librgbdfe.so; shared declarations: rgbdfe_host.h)
#include <chrono>

#include "rgbdfe_host.h"

namespace impl {

// ---------------------------------------------------------------------------------------------
// per-frame feature path
// ---------------------------------------------------------------------------------------------
void ensure_detector(rgbdfe_ctx* ctx) {
  static const bool lookahead_env = !(getenv("RGBDFE_DETECT_LOOKAHEAD") && atoi(getenv("RGBDFE_DETECT_LOOKAHEAD")) == 0);
  ctx->orb.lookahead = lookahead_env;  // A/B switch: one device pass per adjuster iteration when 0
  if (ctx->orb_max_keypoints == 0) {
    ctx->orb_max_keypoints = 600;  // parameter_server.cpp:83
    ctx->orb.reset_detector(600, 3, 5);  // detector_grid_resolution 3, adjuster_max_iterations 5 (:87,:89)
  }
}

int rgbdfe_detector_configure(rgbdfe_ctx* ctx, int32_t max_keypoints, int32_t grid_resolution,
                              int32_t adjuster_max_iterations) {
  if (!ctx || max_keypoints < 1 || grid_resolution < 1 || grid_resolution > 8 || adjuster_max_iterations < 1)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad detector configuration");
  std::lock_guard<std::mutex> g(ctx->mu);
  ctx->orb_max_keypoints = max_keypoints;
  ctx->orb.reset_detector(max_keypoints, grid_resolution, adjuster_max_iterations);
  return RGBDFE_OK;
}

int rgbdfe_detector_thresholds(rgbdfe_ctx* ctx, double* thresholds, int32_t* n_cells) {
  if (!ctx || !thresholds || !n_cells) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  ensure_detector(ctx);
  *n_cells = ctx->orb.grid * ctx->orb.grid;
  for (int i = 0; i < *n_cells; ++i) thresholds[i] = ctx->orb.thresh[i];
  return RGBDFE_OK;
}

void kp_to_abi(const std::vector<KpOut>& v, rgbdfe_keypoint* out) {
  for (size_t i = 0; i < v.size(); ++i) {
    out[i].x = v[i].x; out[i].y = v[i].y; out[i].size = v[i].size; out[i].angle = v[i].angle;
    out[i].response = v[i].response; out[i].octave = v[i].octave;
  }
}

int rgbdfe_orb_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols,
                      int32_t fast_threshold, rgbdfe_keypoint* keypoints, int32_t capacity, int32_t* n_out) {
  if (!ctx || !gray || rows < 1 || cols < 1 || !keypoints || !n_out || capacity < 0)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  ensure_detector(ctx);
  std::string err;
  int rc = ctx->orb.prepare(cols, rows, false, err);
  if (rc == RGBDFE_OK) rc = ctx->orb.upload_and_build(gray, mask, ctx->stream, err);
  std::vector<std::vector<KpOut>> out(1);
  if (rc == RGBDFE_OK) rc = ctx->orb.detect_pass({1}, {fast_threshold}, out, ctx->stream, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  if ((int)out[0].size() > capacity) out[0].resize((size_t)capacity);
  kp_to_abi(out[0], keypoints);
  *n_out = (int32_t)out[0].size();
  return RGBDFE_OK;
}

int rgbdfe_orb_compute(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols,
                       rgbdfe_keypoint* keypoints, int32_t n, uint8_t* descriptors, int32_t* n_out) {
  if (!ctx || !gray || rows < 1 || cols < 1 || n < 0 || (n > 0 && (!keypoints || !descriptors)) || !n_out)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  ensure_detector(ctx);
  std::string err;
  int rc = ctx->orb.prepare(cols, rows, false, err);
  if (rc == RGBDFE_OK) rc = ctx->orb.upload_and_build(gray, nullptr, ctx->stream, err);
  std::vector<KpOut> kps((size_t)n);
  for (int i = 0; i < n; ++i)
    kps[i] = KpOut{keypoints[i].x, keypoints[i].y, keypoints[i].size, keypoints[i].angle, keypoints[i].response,
                   keypoints[i].octave};
  std::vector<uint8_t> desc;
  if (rc == RGBDFE_OK) rc = ctx->orb.compute(kps, desc, ctx->stream, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  kp_to_abi(kps, keypoints);
  if (!desc.empty()) memcpy(descriptors, desc.data(), desc.size());
  *n_out = (int32_t)kps.size();
  return RGBDFE_OK;
}

// SiftGPUWrapper::detect (src/sift_gpu_wrapper.cpp:113-167): SIFT keypoints + 128-d descriptors of one mono8 image.  The
// mask is accepted and ignored, as the reference ignores it.  Keypoints as the wrapper builds them (:156-160):
// pt = SiftGPU's (x, y), size = 12 * scale, angle = orientation in degrees; response and octave stay 0.
int rgbdfe_sift_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* /*mask*/, int32_t rows, int32_t cols,
                       int32_t max_keypoints, rgbdfe_keypoint* keypoints, float* desc128, int32_t capacity, int32_t* n_out) {
  if (!ctx || !gray || rows < 1 || cols < 1 || !n_out || capacity < 0 || (capacity > 0 && (!keypoints || !desc128)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  *n_out = 0;
  std::vector<SiftKey> keys;
  const float* desc = nullptr;
  std::string err;
  const int rc = ctx->sift.run(gray, rows, cols, max_keypoints, keys, desc, ctx->stream, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  *n_out = (int32_t)keys.size();
  if ((int32_t)keys.size() > capacity) return fail(ctx, RGBDFE_ERR_CAPACITY, "more SIFT features than the output arrays hold");
  for (size_t i = 0; i < keys.size(); ++i) {
    keypoints[i].x = keys[i].x;
    keypoints[i].y = keys[i].y;
    keypoints[i].size = (float)(12.0 * keys[i].s);
    keypoints[i].angle = (float)(keys[i].o * 180.0 / 3.1415927);
    keypoints[i].response = 0.f;
    keypoints[i].octave = 0;
  }
  if (!keys.empty()) memcpy(desc128, desc, keys.size() * 128 * sizeof(float));
  return RGBDFE_OK;
}

// SiftGPUWrapper::detect with a non-empty keypoint list (sift_gpu_wrapper.cpp:132-142, 161-165): feature_extractor_type ==
// "SIFTGPU" behind another detector (node.cpp:166-171).  The keypoints' positions, sizes and angles go through the wrapper's
// conversions (o = angle / 180 * 3.1415927, s = size / 12) and come back as the wrapper rebuilds them (12 * s, o * 180 /
// 3.1415927, response = octave = 0); desc128 gets one row per keypoint, in the callers' order.
int rgbdfe_sift_describe(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints, int32_t n,
                         float* desc128) {
  if (!ctx || !gray || rows < 1 || cols < 1 || n < 0 || (n > 0 && (!keypoints || !desc128)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  if (n == 0) return RGBDFE_OK;
  std::vector<SiftKey> keys((size_t)n);
  for (int32_t i = 0; i < n; ++i) {
    keys[(size_t)i].x = keypoints[i].x;
    keys[(size_t)i].y = keypoints[i].y;
    keys[(size_t)i].o = (float)(keypoints[i].angle / 180.0 * 3.1415927);
    keys[(size_t)i].s = (float)(keypoints[i].size / 12.0);
  }
  const float* desc = nullptr;
  std::string err;
  const int rc = ctx->sift.describe(gray, rows, cols, keys.data(), n, &desc, ctx->stream, err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  memcpy(desc128, desc, (size_t)n * 128 * sizeof(float));
  for (int32_t i = 0; i < n; ++i) {
    keypoints[i].size = (float)(12.0 * keys[(size_t)i].s);
    keypoints[i].angle = (float)(keys[(size_t)i].o * 180.0 / 3.1415927);
    keypoints[i].response = 0.f;
    keypoints[i].octave = 0;
  }
  return RGBDFE_OK;
}

// A run of frames (a recorded sequence): SiftExtractor::kMaxBatch of them share every launch of the pipeline -- the
// images are independent (SiftGPU keeps no state between them), so frame f's outputs are those of a single call.
int rgbdfe_sift_detect_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, int32_t rows, int32_t cols,
                             int32_t max_keypoints, int32_t out_stride, rgbdfe_keypoint* keypoints, float* desc128,
                             int32_t* n_out) {
  if (!ctx || n_frames < 0 || rows < 1 || cols < 1 || out_stride < 0 ||
      (n_frames > 0 && (!gray || !n_out || (out_stride > 0 && (!keypoints || !desc128)))))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  for (int32_t f = 0; f < n_frames; ++f)
    if (!gray[f]) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "null frame");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  for (int32_t f = 0; f < n_frames; ++f) n_out[f] = 0;
  bool overflow = false;
  std::string err;
  // Three extractors, three streams, a software pipeline over chunks of up to 8 frames.  A chunk passes four steps, each of
  // which waits for the launches of the one before, does the host's part and enqueues the next launches (sift_extract.hip):
  //   begin_batch           images staged + uploaded, pyramids, extremum flags, candidate lists     ~1.0 ms of device time
  //   finish_orientations   feature-count limits -> the orientation launch                          ~0.06 ms
  //   finish_descriptors    one feature per orientation -> the descriptor launch + 6 MB download    ~0.45 ms
  //   finish_outputs        keys / descriptor pointers; then the copy into the caller's arrays      host only, ~0.35 ms
  // One host thread drives all of it, so the ORDER of the steps decides what the device has to do while the host works or
  // waits: within an iteration the descriptor launch of chunk c goes out first, then chunk c + 2 is begun (its extractor's
  // device buffers were released by chunk c - 1's last wait; the results of c - 1 sit in host memory begin_batch does not
  // touch), then chunk c - 1 is copied out -- with two launch chains and a descriptor launch enqueued behind the host's
  // back -- and only then the host waits: for chunk c + 1's first half (begun a whole iteration ago) and chunk c's
  // descriptors.  Rounds 3 - 4 ran begin(c + 1) | all of finish(c) | copy-out(c) with two extractors: the device idled
  // through every copy-out and most waits (profiles/r05_logs/sift_pipeline.txt).
  constexpr int B = SiftExtractor::kMaxBatch, D = 3;   // (chunks of 4 or 6 frames measured no faster: 6820 / 6944 vs 6870 frames/s)
  const int32_t n_chunks = (n_frames + B - 1) / B;
  SiftExtractor* ex[D] = {&ctx->sift, &ctx->sift2, &ctx->sift3};
  // (the chunk streams come from the high-priority class, created back to back: queues of a pool nothing else in the
  // process is likely to use -- see create_side_stream; equal priority, so no chunk starves another)
  if (n_chunks > 1 && !ctx->sift_stream3) {
    if (!ctx->sift_stream1) HIP_TRY(ctx, create_side_stream(&ctx->sift_stream1, +1));
    if (!ctx->sift_stream2) HIP_TRY(ctx, create_side_stream(&ctx->sift_stream2, +1));
    HIP_TRY(ctx, create_side_stream(&ctx->sift_stream3, +1));
  }
  hipStream_t st[D] = {ctx->sift_stream1 ? ctx->sift_stream1 : ctx->stream, ctx->sift_stream2 ? ctx->sift_stream2 : ctx->stream,
                       ctx->sift_stream3 ? ctx->sift_stream3 : ctx->stream};
  std::vector<SiftKey> keys[D][SiftExtractor::kMaxBatch];
  const float* desc[D][SiftExtractor::kMaxBatch];
  auto count_of = [&](int32_t c) { return std::min<int32_t>(B, n_frames - c * B); };
  auto drain = [&]() { for (int i = 0; i < D; ++i) (void)hipStreamSynchronize(st[i]); };
  auto copy_out = [&](int32_t c) {
    const int32_t f0 = c * B;
    const int nf = count_of(c);
    for (int k = 0; k < nf; ++k) {
      const std::vector<SiftKey>& K = keys[c % D][k];
      const int32_t f = f0 + k;
      n_out[f] = (int32_t)K.size();
      if ((int32_t)K.size() > out_stride) { overflow = true; continue; }
      rgbdfe_keypoint* kp = keypoints + (size_t)f * out_stride;
      for (size_t i = 0; i < K.size(); ++i) {
        kp[i].x = K[i].x;
        kp[i].y = K[i].y;
        kp[i].size = (float)(12.0 * K[i].s);
        kp[i].angle = (float)(K[i].o * 180.0 / 3.1415927);
        kp[i].response = 0.f;
        kp[i].octave = 0;
      }
      if (!K.empty()) memcpy(desc128 + (size_t)f * out_stride * 128, desc[c % D][k], K.size() * 128 * sizeof(float));
    }
  };
  // RGBDFE_SIFT_TIMING=1: the calling thread's time per step, summed over the call, on stderr (how the order above was found)
  static const bool timing = getenv("RGBDFE_SIFT_TIMING") && atoi(getenv("RGBDFE_SIFT_TIMING")) != 0;
  double t_step[5] = {0, 0, 0, 0, 0};   // begin, orientations, descriptors, outputs, copy-out
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_call = timing ? now() : 0;
  const double stage0 = ctx->sift.stage_us + ctx->sift2.stage_us + ctx->sift3.stage_us;
#define SIFT_STEP(slot, expr)                                 \
  do {                                                        \
    const double t0_ = timing ? now() : 0;                    \
    const int rc_ = (expr);                                   \
    if (timing) t_step[slot] += now() - t0_;                  \
    if (rc_ != RGBDFE_OK) { drain(); return fail(ctx, rc_, err); } \
  } while (0)
  for (int32_t c = 0; c < std::min<int32_t>(2, n_chunks); ++c)
    SIFT_STEP(0, ex[c % D]->begin_batch(gray + (size_t)c * B, count_of(c), rows, cols, st[c % D], err));
  if (n_chunks > 0) SIFT_STEP(1, ex[0]->finish_orientations(max_keypoints, st[0], err));
  for (int32_t c = 0; c < n_chunks; ++c) {
    SIFT_STEP(2, ex[c % D]->finish_descriptors(st[c % D], err));
    if (c + 2 < n_chunks)
      SIFT_STEP(0, ex[(c + 2) % D]->begin_batch(gray + (size_t)(c + 2) * B, count_of(c + 2), rows, cols, st[(c + 2) % D], err));
    if (c >= 1) { const double t0 = timing ? now() : 0; copy_out(c - 1); if (timing) t_step[4] += now() - t0; }
    if (c + 1 < n_chunks) SIFT_STEP(1, ex[(c + 1) % D]->finish_orientations(max_keypoints, st[(c + 1) % D], err));
    SIFT_STEP(3, ex[c % D]->finish_outputs(keys[c % D], desc[c % D], st[c % D], err));
  }
  if (n_chunks > 0) { const double t0 = timing ? now() : 0; copy_out(n_chunks - 1); if (timing) t_step[4] += now() - t0; }
#undef SIFT_STEP
  if (timing)
    fprintf(stderr, "sift_detect_batch %d frames: %.0f us; host us in begin %.0f orientations %.0f descriptors %.0f outputs %.0f copy-out %.0f; of begin, image staging %.0f\n",
            (int)n_frames, now() - t_call, t_step[0], t_step[1], t_step[2], t_step[3], t_step[4],
            ctx->sift.stage_us + ctx->sift2.stage_us + ctx->sift3.stage_us - stage0);
  if (overflow) return fail(ctx, RGBDFE_ERR_CAPACITY, "more SIFT features in a frame than out_stride rows");
  return RGBDFE_OK;
}

// stage access for the parity tests (tests/test_gpu_sift_extract.py): a Gaussian plane / the keypoint candidates of one
// (octave, dog level) of the latest rgbdfe_sift_detect frame
int rgbdfe_sift_debug_plane(rgbdfe_ctx* ctx, int32_t octave, int32_t level, float* out, int32_t capacity_floats, int32_t* w,
                            int32_t* h) {
  if (!ctx || !out || !w || !h) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  std::vector<float> v;
  int ww = 0, hh = 0;
  const int rc = ctx->sift.debug_plane(octave, level, v, &ww, &hh, ctx->stream);
  if (rc != RGBDFE_OK) return fail(ctx, rc, "no such SIFT pyramid plane");
  *w = ww; *h = hh;
  if ((int64_t)v.size() > (int64_t)capacity_floats) return fail(ctx, RGBDFE_ERR_CAPACITY, "plane larger than the output");
  memcpy(out, v.data(), v.size() * sizeof(float));
  return RGBDFE_OK;
}

int rgbdfe_sift_debug_candidates(rgbdfe_ctx* ctx, int32_t octave, int32_t dog_level, float* out, int32_t capacity_rows,
                                 int32_t* n) {
  if (!ctx || !n) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  std::vector<float> v;
  const int rc = ctx->sift.debug_candidates(octave, dog_level, v, ctx->stream);
  if (rc != RGBDFE_OK) return fail(ctx, rc, "no such SIFT level");
  *n = (int32_t)(v.size() / 6);
  if (*n > capacity_rows) return fail(ctx, RGBDFE_ERR_CAPACITY, "more candidates than the output holds");
  if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(float));
  return RGBDFE_OK;
}

int rgbdfe_sift_geometry(rgbdfe_ctx* ctx, int32_t* octave_min, int32_t* octave_num, int32_t* levels, int32_t* dog_levels) {
  if (!ctx || !octave_min || !octave_num || !levels || !dog_levels) return RGBDFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->mu);
  *octave_min = ctx->sift.octave_min; *octave_num = ctx->sift.octave_num;
  *levels = SiftExtractor::kLevels; *dog_levels = SiftExtractor::kDogLevels;
  return RGBDFE_OK;
}

// One frame of Node::Node's feature path in three stages; the caller holds the lock.
//   detect()            detector grid + threshold adaptation (node.cpp:160) on ctx->stream: the frame's keypoints
//   describe_enqueue()  removeDepthless, retainBest, cv::ORB::compute and projectTo3D (:186-210) enqueued on a stream
//   finish()            wait for that stream, hand the results out
// A single call runs them back to back on one stream; rgbdfe_detect_describe_batch runs describe_enqueue of frame k while
// the device executes the detection pass of frame k + 1 (another stream, the other image set).
struct DetectFrame {
  rgbdfe_ctx* ctx = nullptr;
  const uint8_t* gray = nullptr; const uint8_t* mask = nullptr; const float* depth = nullptr;
  int32_t rows = 0, cols = 0;
  double fx = 0, fy = 0, cx = 0, cy = 0, depth_scaling = 1;
  rgbdfe_keypoint* keypoints = nullptr; uint8_t* descriptors = nullptr; float* xyz1 = nullptr; int32_t* n_out = nullptr;
  std::vector<KpOut> kps;
  std::vector<float> zmin;
  std::vector<uint8_t> desc;
  std::vector<int> order;  // compute(): positions, in the list handed to it, of the keypoints it keeps, in output order
  std::vector<float> xyz_in_big, xyz_out_big;
  float* xyz_out = nullptr;
  bool tm = false;
  double tq = 0;
  void lap(int slot) {
    if (!tm) return;
    const double now = orb_now_us();
    ctx->orb.timing.us[slot] += now - tq;
    tq = now;
  }

  int detect(bool uploaded, const std::function<int()>& prefetch) {
    OrbWorkspace& orb = ctx->orb;
    std::string err;
    static const bool timing_env = getenv("RGBDFE_DETECT_TIMING") && atoi(getenv("RGBDFE_DETECT_TIMING")) != 0;
    orb.timing.on = timing_env;
    tm = timing_env;
    tq = tm ? orb_now_us() : 0;
    int rc = orb.prepare(cols, rows, true, err);
    if (rc != RGBDFE_OK) return fail(ctx, rc, err);
    // hasNonZero(sub_mask) per cell (feature_adjuster.cpp:175-183)
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
    // the depth image stays on the host: removeDepthless and projectTo3D look at one pixel per keypoint
    lap(0);
    if (!uploaded) rc = orb.upload_and_build(gray, mask, ctx->stream, err, -1, /*defer_blur=*/true);
    lap(1);
    orb.before_wait = prefetch;
    const double pass_before = tm ? orb.timing.us[2] + orb.timing.us[3] + orb.timing.us[4] : 0;
    if (rc == RGBDFE_OK) rc = orb.grid_detect(kps, ctx->stream, err);  // node.cpp:160
    if (rc == RGBDFE_OK && orb.before_wait) {  // (cannot happen: a frame has at least one pass) -- never lose the hook
      std::function<int()> f = std::move(orb.before_wait);
      orb.before_wait = nullptr;
      rc = f();
    }
    orb.before_wait = nullptr;
    if (rc != RGBDFE_OK) return fail(ctx, rc, err);
    if (tm) {  // grid_detect minus its detection passes = the adjuster logic + the per-cell merge
      const double now = orb_now_us();
      orb.timing.us[5] += (now - tq) - (orb.timing.us[2] + orb.timing.us[3] + orb.timing.us[4] - pass_before);
      tq = now;
    }
    return RGBDFE_OK;
  }

  int describe_enqueue(hipStream_t st) {
    OrbWorkspace& orb = ctx->orb;
    const int max_kp = ctx->orb_max_keypoints;
    std::string err;
    int rc = RGBDFE_OK;
    if (tm) tq = orb_now_us();
    // "use_feature_min_depth" (parameter_server.cpp:90, rgbdfe_set_feature_min_depth): a keypoint's depth is the nearest
    // valid depth of its neighbourhood (getMinDepthInNeighborhood, misc.cpp:774-793) -- looked up on the device for all
    // keypoints at once (the depth image is uploaded in this mode only) and carried along with the keypoints from here on.
    const bool min_depth = ctx->feature_min_depth;
    if (min_depth && !kps.empty()) {
      const int n0 = (int)kps.size();
      const size_t b_depth = ((size_t)rows * cols * 4 + 255) & ~(size_t)255;
      const size_t b_kp = ((size_t)n0 * 12 + 255) & ~(size_t)255;
      rc = ensure_scratch(ctx, b_depth + b_kp + (size_t)n0 * 4 + 256);
      if (rc != RGBDFE_OK) return rc;
      float* d_depth = (float*)ctx->d_scratch;
      float* d_kps3 = (float*)((char*)ctx->d_scratch + b_depth);
      float* d_z = (float*)((char*)ctx->d_scratch + b_depth + b_kp);
      std::vector<float> h3((size_t)n0 * 3);
      for (int i = 0; i < n0; ++i) { h3[3 * i] = kps[i].x; h3[3 * i + 1] = kps[i].y; h3[3 * i + 2] = kps[i].size; }
      zmin.resize((size_t)n0);
      HIP_TRY(ctx, hipMemcpyAsync(d_depth, depth, (size_t)rows * cols * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(d_kps3, h3.data(), h3.size() * 4, hipMemcpyHostToDevice, st));
      launch_min_depth(d_kps3, n0, d_depth, rows, cols, d_z, st);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemcpyAsync(zmin.data(), d_z, (size_t)n0 * 4, hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    if (min_depth) {  // removeDepthless with the neighbourhood depth (node.cpp:82)
      size_t m = 0;
      for (size_t i = 0; i < kps.size(); ++i) {
        const KpOut& k = kps[i];
        if (k.x >= (float)cols || k.x < 0 || k.y >= (float)rows || k.y < 0 || std::isnan(k.x) || std::isnan(k.y)) continue;
        if (std::isnan(zmin[i])) continue;
        zmin[m] = zmin[i];
        kps[m++] = k;
      }
      kps.resize(m);
      zmin.resize(m);
    } else {  // removeDepthless (node.cpp:67-97, :186)
      // one scattered read of the 1.2 MB depth image per keypoint: issue them all before the first is needed (the loop
      // below otherwise pays a cache miss per keypoint, ~100 us per frame)
      for (const KpOut& k : kps) {
        if (!(k.x >= 0 && k.x < (float)cols && k.y >= 0 && k.y < (float)rows)) continue;
        int r = (int)roundf(k.y), c = (int)roundf(k.x);
        r = r >= rows ? rows - 1 : r;
        c = c >= cols ? cols - 1 : c;
        __builtin_prefetch(depth + (size_t)r * cols + c, 0, 1);
      }
      size_t m = 0;
      for (const KpOut& k : kps) {
        if (k.x >= (float)cols || k.x < 0 || k.y >= (float)rows || k.y < 0 || std::isnan(k.x) || std::isnan(k.y)) continue;
        int r = (int)roundf(k.y), c = (int)roundf(k.x);
        r = r >= rows ? rows - 1 : r;
        c = c >= cols ? cols - 1 : c;
        if (std::isnan(depth[(size_t)r * cols + c])) continue;
        kps[m++] = k;
      }
      kps.resize(m);
    }
    if ((int)kps.size() > max_kp) {  // retainBest(max_keypoints) + resize (node.cpp:188-191)
      // the max_kp first of the order (response descending, position ascending), in their original order: a selection
      std::vector<std::pair<float, int>> r(kps.size());
      for (size_t i = 0; i < kps.size(); ++i) r[i] = std::make_pair(kps[i].response, (int)i);
      auto before = [](const std::pair<float, int>& a, const std::pair<float, int>& b) {
        return a.first > b.first || (a.first == b.first && a.second < b.second);
      };
      std::nth_element(r.begin(), r.begin() + (max_kp - 1), r.end(), before);
      const std::pair<float, int> cut = r[(size_t)max_kp - 1];
      size_t m = 0;
      for (size_t i = 0; i < kps.size(); ++i)
        if (!before(cut, std::make_pair(kps[i].response, (int)i))) {
          if (min_depth) zmin[m] = zmin[i];
          kps[m++] = kps[i];
        }
      kps.resize(m);
      if (min_depth) zmin.resize(m);
    }
    // cv::ORB::compute (node.cpp:202) drops border keypoints and regroups the rest by octave, so projectTo3D
    // (node.cpp:210) is enqueued from inside compute_enqueue(), once the final keypoint list exists: both ride on one
    // synchronisation.  xy (2n floats) + depth.at<float>(round(y), round(x)) (n floats, node.cpp:942): 12 bytes per
    // keypoint cross PCIe instead of the 1.2 MB image.
    auto enqueue_project = [&]() -> int {
      const int n = (int)kps.size();
      if (n == 0) return RGBDFE_OK;
      float* xyz_in = orb.h_xyz_in;
      xyz_out = orb.h_xyz_out;
      if (n > orb.pin_cap) {
        xyz_in_big.resize((size_t)n * 3); xyz_out_big.resize((size_t)n * 4);
        xyz_in = xyz_in_big.data(); xyz_out = xyz_out_big.data();
      }
      for (int i = 0; i < n; ++i) {
        xyz_in[2 * i] = kps[i].x;
        xyz_in[2 * i + 1] = kps[i].y;
        if (min_depth) {  // node.cpp:940-941: the same neighbourhood depth as in removeDepthless
          xyz_in[(size_t)2 * n + i] = zmin[(size_t)order[(size_t)i]];
          continue;
        }
        int r = (int)roundf(kps[i].y), c = (int)roundf(kps[i].x);
        r = r >= rows ? rows - 1 : r;
        c = c >= cols ? cols - 1 : c;
        xyz_in[(size_t)2 * n + i] = depth[(size_t)r * cols + c];
      }
      if (hipMemcpyAsync(orb.d_kpxy, xyz_in, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)
        return RGBDFE_ERR_HIP;
      launch_project_to_3d(orb.d_kpxy, n, nullptr, rows, cols, (float)(1. / fx), (float)(1. / fy), (float)cx,
                           (float)cy, depth_scaling, max_kp, orb.d_kept, orb.d_xyz, orb.d_n_proj, st, false,
                           orb.d_kpxy + (size_t)2 * n);
      if (hipGetLastError() != hipSuccess) return RGBDFE_ERR_HIP;
      if (hipMemcpyAsync(orb.h_n_proj, orb.d_n_proj, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipMemcpyAsync(xyz_out, orb.d_xyz, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess)
        return RGBDFE_ERR_HIP;
      return RGBDFE_OK;
    };
    lap(6);
    rc = orb.compute_enqueue(kps, desc, st, err, enqueue_project, &order);
    if (rc != RGBDFE_OK) return fail(ctx, rc, err);
    return RGBDFE_OK;
  }

  int finish(hipStream_t st) {
    OrbWorkspace& orb = ctx->orb;
    std::string err;
    const int rc = orb.compute_finish(desc, st, err);
    if (rc != RGBDFE_OK) return fail(ctx, rc, err);
    if (tm) tq = orb_now_us();
    const int n = (int)kps.size();
    *n_out = 0;
    if (n > 0) {
      if (*orb.h_n_proj != n) return fail(ctx, RGBDFE_ERR_HIP, "projectTo3D dropped keypoints that removeDepthless kept");
      memcpy(xyz1, xyz_out, sizeof(float) * 4 * (size_t)n);
    }
    kp_to_abi(kps, keypoints);
    if (!desc.empty()) memcpy(descriptors, desc.data(), desc.size());
    *n_out = n;
    lap(9);
    if (tm) orb.timing.frames++;
    return RGBDFE_OK;
  }
};

int rgbdfe_detect_describe(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* depth,
                           int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                           double depth_scaling, rgbdfe_keypoint* keypoints, uint8_t* descriptors,
                           float* xyz1, int32_t* n_out) {
  if (!ctx || !gray || !depth || rows < 1 || cols < 1 || !keypoints || !descriptors || !xyz1 || !n_out)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  ensure_detector(ctx);
  DetectFrame fr;
  fr.ctx = ctx; fr.gray = gray; fr.mask = mask; fr.depth = depth; fr.rows = rows; fr.cols = cols;
  fr.fx = fx; fr.fy = fy; fr.cx = cx; fr.cy = cy; fr.depth_scaling = depth_scaling;
  fr.keypoints = keypoints; fr.descriptors = descriptors; fr.xyz1 = xyz1; fr.n_out = n_out;
  int rc = fr.detect(false, nullptr);
  if (rc == RGBDFE_OK) rc = fr.describe_enqueue(ctx->stream);
  if (rc == RGBDFE_OK) rc = fr.finish(ctx->stream);
  return rc;
}

// ---- rgbdfe_detect_describe_batch, super-frame form -----------------------------------------------------------------
constexpr int kSuperFrames = 7;   // frames per super-frame at least (large frames)
// B = 64 / grid^2 (7 for the 3 x 3 grid) frames share every launch: one upload, one pyramid chain (7 launches), one blur,
// one detection pass (FAST score -> NMS count -> scan -> emit -> measure) over 7 x 72 images, one rBRIEF launch -- a frame
// alone is 1.7 M pixels and cannot fill 256 CUs, and its 27 dependent device operations cost 5-15 us each whatever their
// size.  Frames stay sequentially dependent through the per-cell FAST thresholds: OrbWorkspace::super_detect runs the
// device pass at a floor threshold and replays the reference's adjuster over the scored corners frame by frame (identical
// keypoints: see select_pass).  The per-frame CPU work that does not touch the HIP runtime (removeDepthless, retainBest,
// cv::ORB::compute's border filter / regroup / descriptor records, the depth look-ups) runs on worker threads while the
// calling thread drives the next super-frame's pass.
namespace {

struct SuperFrameJob {  // one frame of a super-frame, between detection and copy-out
  bool deferred = false;              // the frame's keypoints are still to be selected from the pass's corners (select_frame)
  OrbWorkspace::PassView pv;
  std::vector<int> thr;               // every (frame, cell)'s final threshold of the super-frame
  std::vector<KpOut> kps;
  std::vector<int> order;
  std::vector<DescKp> dk;
  std::vector<float> xyz_in;  // 2n (x, y) then n depths
  int off = 0;                // first row of this frame in the super-frame's concatenated buffers
};

// removeDepthless (node.cpp:67-97, :186) + retainBest(max_keypoints) (:188-191) + the CPU half of cv::ORB::compute +
// projectTo3D's depth look-ups (node.cpp:942) for one frame: DetectFrame::describe_enqueue's host work, no HIP calls
void super_describe_prepare(const OrbWorkspace& orb, SuperFrameJob& j, int frame_in_super, const float* depth, int rows,
                            int cols, int max_kp) {
  std::vector<KpOut>& kps = j.kps;
  if (j.deferred) orb.select_frame(j.pv, frame_in_super, j.thr.data(), kps);
  size_t m = 0;
  for (const KpOut& k : kps) {
    if (k.x >= (float)cols || k.x < 0 || k.y >= (float)rows || k.y < 0 || std::isnan(k.x) || std::isnan(k.y)) continue;
    int r = (int)roundf(k.y), c = (int)roundf(k.x);
    r = r >= rows ? rows - 1 : r;
    c = c >= cols ? cols - 1 : c;
    if (std::isnan(depth[(size_t)r * cols + c])) continue;
    kps[m++] = k;
  }
  kps.resize(m);
  if ((int)kps.size() > max_kp) {  // the max_kp first of the order (response descending, position ascending), in place
    std::vector<std::pair<float, int>> r(kps.size());
    for (size_t i = 0; i < kps.size(); ++i) r[i] = std::make_pair(kps[i].response, (int)i);
    auto before = [](const std::pair<float, int>& a, const std::pair<float, int>& b) {
      return a.first > b.first || (a.first == b.first && a.second < b.second);
    };
    std::nth_element(r.begin(), r.begin() + (max_kp - 1), r.end(), before);
    const std::pair<float, int> cut = r[(size_t)max_kp - 1];
    m = 0;
    for (size_t i = 0; i < kps.size(); ++i)
      if (!before(cut, std::make_pair(kps[i].response, (int)i))) kps[m++] = kps[i];
    kps.resize(m);
  }
  orb.compute_prepare(kps, frame_in_super, j.order, j.dk);
  const int n = (int)kps.size();
  j.xyz_in.resize((size_t)n * 3);
  for (int i = 0; i < n; ++i) {
    j.xyz_in[(size_t)2 * i] = kps[i].x;
    j.xyz_in[(size_t)2 * i + 1] = kps[i].y;
    int r = (int)roundf(kps[i].y), c = (int)roundf(kps[i].x);
    r = r >= rows ? rows - 1 : r;
    c = c >= cols ? cols - 1 : c;
    j.xyz_in[(size_t)2 * n + i] = depth[(size_t)r * cols + c];
  }
}

int detect_describe_batch_super(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                uint8_t* descriptors, float* xyz1, int32_t* n_out, const int32_t* node_ids) {
  OrbWorkspace& orb = ctx->orb_super;
  const OrbWorkspace& one = ctx->orb;
  const int pc = one.grid * one.grid;
  // frames per super-frame: as many as the launches' tables hold, at most kSuperFrames (RGBDFE_SUPER_FRAMES: A/B runs)
  static const int super_frames_env = getenv("RGBDFE_SUPER_FRAMES") ? atoi(getenv("RGBDFE_SUPER_FRAMES")) : 0;
  // default: about 4.3 Mpixel of frames per launch, 7 to 28 frames -- 14 at 640 x 480 (measured 7 / 14 / 28 frames: 13.4 / 16.3 /
  // 16.6 k frames/s), 7 at 1280 x 960 (4.7 / 4.7 / 1.9 k: a 112-frame run is only four super-frames of 28)
  const int by_size = std::max(kSuperFrames, std::min(28, (int)(4300000.0 / ((double)rows * (double)cols) + 0.5)));
  const int B = std::max(1, std::min(std::min(kOrbCtlMax / pc, kProjectFramesMax), super_frames_env > 0 ? super_frames_env : by_size));
  // the detector object is one: its configuration and thresholds move into the super-frame workspace and back
  orb.grid = one.grid; orb.adjuster_iters = one.adjuster_iters; orb.cell_min = one.cell_min; orb.cell_max = one.cell_max;
  orb.max_total = one.max_total; orb.lookahead = one.lookahead;
  for (int i = 0; i < 64; ++i) orb.thresh[i] = one.thresh[i];
  if (const char* e = getenv("RGBDFE_SUPER_FLOOR")) orb.super_floor_factor = atof(e);  // experiments
  std::string err;
  int rc = orb.prepare(cols, rows, true, err, B);
  if (rc == RGBDFE_OK) rc = orb.ensure_alt(err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  if (!ctx->orb_upload_stream) {
    HIP_TRY(ctx, create_side_stream(&ctx->orb_upload_stream, -1));   // uploads + pyramids: behind everything else
    HIP_TRY(ctx, create_side_stream(&ctx->orb_compute_stream, +1));  // descriptions: short, the host waits for them
    for (hipEvent_t& e : ctx->orb_upload_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->orb_describe_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipStream_t up = ctx->orb_upload_stream, st2 = ctx->orb_compute_stream;
  const int max_kp = ctx->orb_max_keypoints;
  if (node_ids) {
    // all-or-nothing on capacity, like rgbdfe_upload_nodes: everything that can be refused is refused before the first
    // frame is detected.  Every fresh id is counted as needing a slot: a frame that ends up WITHOUT features still becomes an
    // (empty, n = 0) node -- a fresh id takes its slot, an existing node is rewritten to n = 0 (the end of this function) -- so
    // the count is exact.
    if (max_kp > ctx->cfg.max_keypoints)
      return fail(ctx, RGBDFE_ERR_CAPACITY, "the detector's max_keypoints exceeds the context's max_keypoints (node rows)");
    bool overwrite = false;
    std::unordered_set<int32_t> fresh_ids;
    for (int32_t f = 0; f < n_frames; ++f) {
      if (node_ids[f] < 0) continue;
      if (ctx->nodes.count(node_ids[f]) != 0) overwrite = true;
      else fresh_ids.insert(node_ids[f]);
    }
    if (fresh_ids.size() > ctx->free_slots.size()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
    if (overwrite)  // nodes rewritten in place: wait for pair batches that may still read them
      for (auto& ln : ctx->lanes) HIP_TRY(ctx, hipStreamSynchronize(ln.stream));
  }
  const int S = (n_frames + B - 1) / B;
  // D - 1 device passes are in flight ahead of the super-frame the host is replaying: D image sets / pass slots, D + 1
  // staging buffers.  A pass is a chain of ~25 dependent device operations (~1 ms from enqueue to read-back although its
  // kernels take < 0.5 ms), so one pass ahead leaves the host waiting; two hide the chain.  RGBDFE_SUPER_DEPTH=2: one ahead.
  int D = OrbWorkspace::kSets;
  if (const char* e = getenv("RGBDFE_SUPER_DEPTH")) D = std::min(std::max(atoi(e), 2), (int)OrbWorkspace::kSets);
  auto first_of = [&](int s) { return s * B; };
  auto count_of = [&](int s) { return std::min(B, n_frames - s * B); };
  // helper thread: the caller's pageable images of super-frame s -> pinned staging buffer s % NS, as soon as super-frame
  // s - NS (the buffer's previous user) has been detected (its upload from that buffer is complete then)
  const int NS = std::max(D + 1, std::min((int)OrbWorkspace::kStages,
                                          getenv("RGBDFE_SUPER_STAGES") ? atoi(getenv("RGBDFE_SUPER_STAGES")) : (int)OrbWorkspace::kStages));
  std::mutex m;
  std::condition_variable cv;
  int staged = 0, detected = 0;
  bool stop = false;
  static const int stage_threads = getenv("RGBDFE_STAGE_THREADS") ? std::max(1, atoi(getenv("RGBDFE_STAGE_THREADS"))) : 4;
  if (!ctx->stage_pool) ctx->stage_pool.reset(new TaskPool(stage_threads));
  // (the frames' CPU halves: a worker per frame up to RGBDFE_DETECT_WORKERS workers, the frames queue behind them)
  static const int max_workers = getenv("RGBDFE_DETECT_WORKERS") ? std::max(1, atoi(getenv("RGBDFE_DETECT_WORKERS"))) : 14;   // (a worker per frame of a 14-frame super-frame; 7 / 12 / 14 workers: 20.6 / 20.3 / 21.2 k frames/s, four alternating runs each)
  const int n_workers = std::min(B, max_workers);
  if (!ctx->detect_pool || ctx->detect_pool->size() != n_workers) ctx->detect_pool.reset(new TaskPool(n_workers));
  TaskPool& stage_pool = *ctx->stage_pool;
  TaskPool& pool = *ctx->detect_pool;
  pool.failed_ = false;
  std::thread helper([&]() {
    for (int s = 0; s < S; ++s) {
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || detected >= s - NS + 1; });
        if (stop) return;
      }
      // (2 W H bytes per frame through one core's memcpy would bound the whole pipeline: 75 us per 640 x 480 frame)
      for (int k = 0; k < count_of(s); ++k) {
        const int f = first_of(s) + k;
        stage_pool.submit([&orb, &gray, &mask, f, s, k, NS] { orb.stage_image_at(gray[f], mask ? mask[f] : nullptr, s % NS, k); });
      }
      stage_pool.wait_all();
      std::lock_guard<std::mutex> l(m);
      staged = s + 1;
      cv.notify_all();
    }
  });
  struct HelperJoin {
    std::thread& th; std::mutex& m; std::condition_variable& cv; bool& stop;
    ~HelperJoin() {
      { std::lock_guard<std::mutex> l(m); stop = true; }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
  } helper_join{helper, m, cv, stop};
  // how the host replays the adjuster over a super-frame's scored corners: 1 (default) from counts on the calling thread, the
  // selections themselves inside the frames' description jobs; 2 = per-cell chains + per-frame merges on the worker pool
  // while the calling thread waits; 0 = the sequential loop
  static const int replay_mode = getenv("RGBDFE_SUPER_PARALLEL_REPLAY") ? atoi(getenv("RGBDFE_SUPER_PARALLEL_REPLAY")) : 1;
  static const bool par_replay = replay_mode == 2;
  struct ParallelForGuard {  // the workspace outlives the pool
    OrbWorkspace& o;
    ~ParallelForGuard() { o.parallel_for = nullptr; }
  } pf_guard{orb};
  if (par_replay) orb.parallel_for = [&pool](int n, const std::function<void(int)>& fn) { pool.parallel_for(n, fn); };
  else orb.parallel_for = nullptr;
  std::vector<SuperFrameJob> jobs[2];
  jobs[0].resize((size_t)B); jobs[1].resize((size_t)B);
  int n_tot[2] = {0, 0};
  auto enqueue_upload = [&](int s) -> int {
    {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return staged > s; });
    }
    if (s >= D && hipStreamWaitEvent(up, ctx->orb_describe_done[s % D], 0) != hipSuccess) { err = "hipStreamWaitEvent"; return RGBDFE_ERR_HIP; }
    const int r = orb.enqueue_staged_super(count_of(s), up, err, s % D, s % NS);
    if (r != RGBDFE_OK) return r;
    if (hipEventRecord(ctx->orb_upload_done[s % D], up) != hipSuccess) { err = "hipEventRecord"; return RGBDFE_ERR_HIP; }
    return RGBDFE_OK;
  };
  // the CPU halves of super-frame s's descriptions: worker threads, no HIP calls
  auto start_prepare = [&](int s) {
    std::vector<SuperFrameJob>& J = jobs[s & 1];
    for (int k = 0; k < count_of(s); ++k) {
      const float* dp = depth[first_of(s) + k];
      SuperFrameJob* j = &J[(size_t)k];
      pool.submit([&orb, j, k, dp, rows, cols, max_kp] { super_describe_prepare(orb, *j, k, dp, rows, cols, max_kp); });
    }
  };
  // device half: one descriptor-record upload, one rBRIEF launch, one projectTo3D launch per frame, three read-backs
  auto enqueue_describe = [&](int s) -> int {
    pool.wait_all();
    if (pool.failed_) { err = "describe preparation failed"; return RGBDFE_ERR_INTERNAL; }
    std::vector<SuperFrameJob>& J = jobs[s & 1];
    const int nf = count_of(s);
    int tot = 0;
    for (int k = 0; k < nf; ++k) { J[(size_t)k].off = tot; tot += (int)J[(size_t)k].kps.size(); }
    n_tot[s & 1] = tot;
    if (hipStreamWaitEvent(st2, ctx->orb_upload_done[s % D], 0) != hipSuccess) return RGBDFE_ERR_HIP;
    if (tot > orb.pin_cap || tot > orb.kp_cap) { err = "super-frame: more keypoints than the staging buffers hold"; return RGBDFE_ERR_CAPACITY; }
    if (tot > 0) {
      for (int k = 0; k < nf; ++k) {
        const SuperFrameJob& j = J[(size_t)k];
        const size_t n = j.kps.size();
        if (n == 0) continue;
        memcpy(orb.h_desckp + j.off, j.dk.data(), sizeof(DescKp) * n);
        memcpy(orb.h_xyz_in + (size_t)3 * j.off, j.xyz_in.data(), sizeof(float) * 3 * n);
      }
      uint8_t* const pool_dev = orb.pool_set[s % D];
      uint8_t* const blur_dev = orb.blur_set[s % D];
      if (hipMemcpyAsync(orb.d_desckp, orb.h_desckp, sizeof(DescKp) * (size_t)tot, hipMemcpyHostToDevice, st2) != hipSuccess ||
          hipMemcpyAsync(orb.d_kpxy, orb.h_xyz_in, sizeof(float) * 3 * (size_t)tot, hipMemcpyHostToDevice, st2) != hipSuccess)
        return RGBDFE_ERR_HIP;
      launch_orb_brief(pool_dev, blur_dev, orb.d_frame_imgs, orb.d_desckp, tot, orb.d_desc, st2);
      ProjectFrames pf{};
      pf.n_frames = nf;
      for (int k = 0; k < nf; ++k) { pf.off[k] = J[(size_t)k].off; pf.n[k] = (int)J[(size_t)k].kps.size(); }
      launch_project_to_3d_frames(pf, orb.d_kpxy, rows, cols, (float)(1. / fx), (float)(1. / fy), (float)cx, (float)cy,
                                  depth_scaling, max_kp, orb.d_kept, orb.d_xyz, orb.d_n_proj, st2);
      if (hipGetLastError() != hipSuccess) return RGBDFE_ERR_HIP;
      // node_ids: the frames' features become resident nodes straight from the description's device buffers (no trip
      // through the host and back: what Node::Node + GraphManager::addNode + rgbdfe_upload_node would do)
      if (node_ids)
        for (int k = 0; k < nf; ++k) {
          const int32_t id = node_ids[first_of(s) + k];
          const int n = (int)J[(size_t)k].kps.size();
          if (id < 0 || n == 0) continue;
          if (n > ctx->cfg.max_keypoints) { err = "node has more rows than max_keypoints"; return RGBDFE_ERR_CAPACITY; }
          uint32_t slot;
          auto it = ctx->nodes.find(id);
          if (it != ctx->nodes.end()) slot = it->second.slot;
          else {
            if (ctx->free_slots.empty()) { err = "no free node slot (max_nodes)"; return RGBDFE_ERR_CAPACITY; }
            slot = ctx->free_slots.back();
            ctx->free_slots.pop_back();
            ctx->nodes[id] = NodeEntry{slot, 0u, 0u, 0u};   // registered before anything can fail: no slot goes missing
          }
          const size_t row0 = (size_t)slot * (size_t)ctx->cfg.max_keypoints;
          const size_t off = (size_t)J[(size_t)k].off;
          if (hipMemcpyAsync(ctx->d_desc + row0 * 8, orb.d_desc + off * 32, (size_t)n * 32, hipMemcpyDeviceToDevice, st2) != hipSuccess ||
              hipMemcpyAsync(ctx->d_xyz + row0, orb.d_xyz + off, (size_t)n * 16, hipMemcpyDeviceToDevice, st2) != hipSuccess)
            return RGBDFE_ERR_HIP;
          launch_hamming_expand(ctx->d_desc + row0 * 8, ctx->d_desc4, slot, (uint32_t)ctx->cfg.max_keypoints, (uint32_t)n, st2);
          ctx->nodes[id] = NodeEntry{slot, (uint32_t)n, 0u, 0u};
        }
      if (hipMemcpyAsync(orb.h_desc, orb.d_desc, (size_t)32 * tot, hipMemcpyDeviceToHost, st2) != hipSuccess ||
          hipMemcpyAsync(orb.h_xyz_out, orb.d_xyz, sizeof(float) * 4 * (size_t)tot, hipMemcpyDeviceToHost, st2) != hipSuccess ||
          hipMemcpyAsync(orb.h_n_proj, orb.d_n_proj, sizeof(int32_t) * (size_t)nf, hipMemcpyDeviceToHost, st2) != hipSuccess)
        return RGBDFE_ERR_HIP;
    }
    return hipEventRecord(ctx->orb_describe_done[s % D], st2) == hipSuccess ? RGBDFE_OK : RGBDFE_ERR_HIP;
  };
  auto finish = [&](int s) -> int {
    if (hipStreamSynchronize(st2) != hipSuccess) return RGBDFE_ERR_HIP;
    std::vector<SuperFrameJob>& J = jobs[s & 1];
    for (int k = 0; k < count_of(s); ++k) {
      const SuperFrameJob& j = J[(size_t)k];
      const int f = first_of(s) + k;
      const int n = (int)j.kps.size();
      n_out[f] = 0;
      if (n > 0) {
        if (orb.h_n_proj[k] != n) { err = "projectTo3D dropped keypoints that removeDepthless kept"; return RGBDFE_ERR_HIP; }
        memcpy(xyz1 + (size_t)f * out_stride * 4, orb.h_xyz_out + (size_t)4 * j.off, sizeof(float) * 4 * (size_t)n);
        memcpy(descriptors + (size_t)f * out_stride * 32, orb.h_desc + (size_t)32 * j.off, (size_t)32 * n);
      }
      kp_to_abi(j.kps, keypoints + (size_t)f * out_stride);
      n_out[f] = n;
    }
    return RGBDFE_OK;
  };
  static const bool tm = getenv("RGBDFE_DETECT_TIMING") && atoi(getenv("RGBDFE_DETECT_TIMING")) != 0;
  double t_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // mask scan, super_detect, hook: describe enqueue, hook: upload enqueue, finish, prepare start, pool wait
  double tq = tm ? orb_now_us() : 0;
  auto lap = [&](int i) { if (tm) { const double now = orb_now_us(); t_us[i] += now - tq; tq = now; } };
  orb.timing.on = tm;
  const long passes0 = orb.super_passes;
  // Software pipeline: the device pass of super-frame s + 1 is enqueued BEFORE the host replays the adjuster over
  // super-frame s (its floors come from the thresholds of that moment; a cell that falls below its floor is re-run by
  // super_replay), so the device works on s + 1 while the host selects keypoints of s; description of s - 1 and the
  // upload of s + 1 are enqueued in between, the CPU halves of the descriptions run on the worker threads.
  auto pass_enqueue = [&](int s) -> int {
    if (hipStreamWaitEvent(ctx->stream, ctx->orb_upload_done[s % D], 0) != hipSuccess) { err = "hipStreamWaitEvent"; return RGBDFE_ERR_HIP; }
    return orb.super_pass_enqueue(count_of(s), s % D, s % D, ctx->stream, err);
  };
  for (int s = 0; s < D - 1 && s < S && rc == RGBDFE_OK; ++s) {
    rc = enqueue_upload(s);
    if (rc == RGBDFE_OK) rc = pass_enqueue(s);
  }
  lap(6);
  for (int s = 0; s < S && rc == RGBDFE_OK; ++s) {
    const int nf = count_of(s);
    if (tm) tq = orb_now_us();
    if (s > 0) { rc = enqueue_describe(s - 1); if (rc != RGBDFE_OK) break; }
    lap(2);
    if (s + D - 1 < S) {
      rc = enqueue_upload(s + D - 1);
      if (rc == RGBDFE_OK) rc = pass_enqueue(s + D - 1);
      if (rc != RGBDFE_OK) break;
    }
    lap(3);
    // hasNonZero(sub_mask) per (frame, cell) (feature_adjuster.cpp:175-183)
    orb.cell_mask_nonzero.assign((size_t)orb.n_cells, 1);
    for (int k = 0; k < nf; ++k) {
      const uint8_t* mk = mask ? mask[first_of(s) + k] : nullptr;
      if (!mk) continue;
      for (int c9 = 0; c9 < pc; ++c9) {
        const OrbWorkspace::Cell& ce = orb.cells[(size_t)k * pc + c9];
        char nz = 0;
        for (int y = 0; y < ce.h && !nz; ++y) {
          const uint8_t* r = mk + (size_t)(ce.y0 + y) * cols + ce.x0;
          for (int x = 0; x < ce.w; ++x)
            if (r[x]) { nz = 1; break; }
        }
        orb.cell_mask_nonzero[(size_t)k * pc + c9] = nz;
      }
    }
    lap(0);
    std::vector<std::vector<KpOut>> kps;
    OrbWorkspace::Deferred def;
    rc = orb.super_replay(nf, s % D, s % D, kps, ctx->stream, err, replay_mode == 1 ? &def : nullptr);
    if (rc != RGBDFE_OK) break;
    {
      std::lock_guard<std::mutex> l(m);
      detected = s + 1;
    }
    cv.notify_all();
    lap(1);
    if (s > 0) { rc = finish(s - 1); if (rc != RGBDFE_OK) break; }
    lap(4);
    for (int k = 0; k < nf; ++k) {
      SuperFrameJob& j = jobs[s & 1][(size_t)k];
      j.deferred = def.valid;
      if (def.valid) { j.pv = def.pv; j.thr = def.thr_final; j.kps.clear(); }
      else j.kps.swap(kps[(size_t)k]);
    }
    start_prepare(s);
    lap(5);
  }
  if (tm) {
    fprintf(stderr, "[rgbdfe super-frame timing] depth %d, replay mode %d (sequential fallbacks: %ld); ", D, replay_mode,
            orb.replay_fallbacks);
    fprintf(stderr, "[rgbdfe super-frame timing] %d frames in %d super-frames, %ld device passes; per frame (us): mask scan %.1f, "
            "replay incl. wait for its pass %.1f, describe enqueue %.1f, upload + next pass enqueue %.1f (re-passes: enqueue %.1f, "
            "wait %.1f; selections %.1f), finish %.1f, prepare start %.1f\n", (int)n_frames, S, orb.super_passes - passes0,
            t_us[0] / n_frames, t_us[1] / n_frames, t_us[2] / n_frames, t_us[3] / n_frames, orb.timing.us[2] / n_frames,
            orb.timing.us[3] / n_frames, orb.timing.us[4] / n_frames, t_us[4] / n_frames, t_us[5] / n_frames);
    for (double& u : orb.timing.us) u = 0;
    orb.timing.frames = 0; orb.timing.passes = 0;
    tq = orb_now_us();
  }
  if (rc == RGBDFE_OK) rc = enqueue_describe(S - 1);
  if (rc == RGBDFE_OK) rc = finish(S - 1);
  pool.wait_all();
  (void)hipStreamSynchronize(st2);
  lap(7);
  if (tm) fprintf(stderr, "[rgbdfe super-frame timing] whole call (us): filling the pipeline (staging + upload + pass enqueue of the first "
                          "%d super-frames) %.0f, draining it (the last super-frame's description) %.0f\n", D - 1, t_us[6], t_us[7]);
  {
    std::lock_guard<std::mutex> l(m);
    stop = true;
    cv.notify_all();
  }
  if (helper.joinable()) helper.join();
  (void)hipStreamSynchronize(up);
  (void)hipStreamSynchronize(ctx->stream);
  orb.use_set(0);
  for (int i = 0; i < 64; ++i) ctx->orb.thresh[i] = orb.thresh[i];   // the detector's state goes back
  if (rc != RGBDFE_OK) return err.empty() ? rc : fail(ctx, rc, err);
  if (node_ids)   // frames without features: empty nodes (n = 0), as rgbdfe_upload_node(id, ..., 0) would leave them
    for (int32_t f = 0; f < n_frames; ++f)
      if (node_ids[f] >= 0 && n_out[f] == 0) {
        auto it = ctx->nodes.find(node_ids[f]);
        if (it != ctx->nodes.end()) it->second.n = 0;
        else {
          if (ctx->free_slots.empty()) return fail(ctx, RGBDFE_ERR_CAPACITY, "no free node slot (max_nodes)");
          ctx->nodes[node_ids[f]] = NodeEntry{ctx->free_slots.back(), 0u, 0u, 0u};
          ctx->free_slots.pop_back();
        }
      }
  return RGBDFE_OK;
}

}  // namespace

// A run of frames through the same detector state, in order (the per-cell thresholds of frame k+1 start from frame k's,
// as in a sequence of single calls -- same keypoints, bit for bit).  What the batch adds is overlap, three deep: frame
// k+2's images are staged, uploaded and turned into their pyramid by a helper thread (own stream, the free image set)
// while frame k+1's detection pass runs on the device and the calling thread prepares and enqueues frame k's description
// (third stream) instead of sitting in hipStreamSynchronize.  Outputs: frame f's keypoints /
// descriptors / points at offset f * out_stride (rows), n_out[f] of them.
static int detect_describe_batch_frames(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                        const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                        double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                        uint8_t* descriptors, float* xyz1, int32_t* n_out);
// node_ids (may be NULL): frame f's features also become the resident node node_ids[f] (>= 0), see
// rgbdfe_detect_describe_batch_nodes
int rgbdfe_detect_describe_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                 const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                 double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out, const int32_t* node_ids) {
  if (!ctx || n_frames < 0 || (n_frames > 0 && (!gray || !depth || !keypoints || !descriptors || !xyz1 || !n_out)) ||
      rows < 1 || cols < 1)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad arguments");
  if (node_ids)
    for (int32_t f = 0; f < n_frames; ++f)
      for (int32_t j = 0; j < f; ++j)
        if (node_ids[f] >= 0 && node_ids[j] == node_ids[f]) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "a node id appears twice");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  ensure_detector(ctx);
  if (n_frames > 0 && out_stride < ctx->orb_max_keypoints)
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "out_stride must be at least the configured max_keypoints");
  for (int32_t f = 0; f < n_frames; ++f)
    if (!gray[f] || !depth[f]) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "null frame");
  if (n_frames == 0) return RGBDFE_OK;
  {  // several frames per launch chain (above) unless switched off or the detector is in a mode only the frame path has
    static const bool super_env = !(getenv("RGBDFE_DETECT_SUPER") && atoi(getenv("RGBDFE_DETECT_SUPER")) == 0);
    if (super_env && n_frames >= 2 && !ctx->feature_min_depth && ctx->orb.grid * ctx->orb.grid * 2 <= 64)
      return detect_describe_batch_super(ctx, n_frames, gray, mask, depth, rows, cols, fx, fy, cx, cy, depth_scaling, out_stride,
                                         keypoints, descriptors, xyz1, n_out, node_ids);
  }
  const int rc_frames = detect_describe_batch_frames(ctx, n_frames, gray, mask, depth, rows, cols, fx, fy, cx, cy, depth_scaling,
                                                    out_stride, keypoints, descriptors, xyz1, n_out);
  if (rc_frames != RGBDFE_OK || !node_ids) return rc_frames;
  // the frame-by-frame pipeline hands its nodes over from the host outputs
  std::vector<int32_t> ids, cnt;
  std::vector<const uint8_t*> dp;
  std::vector<const float*> xp;
  for (int32_t f = 0; f < n_frames; ++f)
    if (node_ids[f] >= 0) {
      ids.push_back(node_ids[f]); cnt.push_back(n_out[f]);
      dp.push_back(descriptors + (size_t)f * out_stride * 32); xp.push_back(xyz1 + (size_t)f * out_stride * 4);
    }
  return upload_nodes_locked(ctx, (int32_t)ids.size(), ids.data(), dp.data(), xp.data(), cnt.data());
}

static int detect_describe_batch_frames(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask,
                                        const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                        double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints,
                                        uint8_t* descriptors, float* xyz1, int32_t* n_out) {
  OrbWorkspace& orb = ctx->orb;
  std::string err;
  int rc = orb.prepare(cols, rows, true, err);
  if (rc == RGBDFE_OK) rc = orb.ensure_alt(err);
  if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  if (!ctx->orb_upload_stream) {
    HIP_TRY(ctx, create_side_stream(&ctx->orb_upload_stream, -1));   // uploads + pyramids: behind everything else
    HIP_TRY(ctx, create_side_stream(&ctx->orb_compute_stream, +1));  // descriptions: short, the host waits for them
    for (hipEvent_t& e : ctx->orb_upload_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (hipEvent_t& e : ctx->orb_describe_done) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipStream_t up = ctx->orb_upload_stream;
  // Frame f lives in image set f & 1 (device pyramid + pinned staging buffer).  A helper thread copies the caller's
  // pageable images of frame f into the set's staging buffer as soon as frame f - 2 has been detected (its upload from that
  // buffer is complete then) -- CPU work only: two host threads inside the HIP runtime at once serialise on its locks, and
  // the calling thread's launches are the critical path.  The calling thread enqueues the device half of frame f + 2's
  // upload (one copy + the pyramid launches, on a second stream) from the hook of frame f + 1's detection pass, behind
  // frame f's description, which reads the same set.
  std::mutex m;
  std::condition_variable cv;
  int staged = 0, detected = 0;  // frames staged by the helper / detected by the caller
  bool stop = false;
  std::thread helper([&]() {
    for (int32_t f = 0; f < n_frames; ++f) {
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || detected >= f - 1; });
        if (stop) return;
      }
      orb.stage_images(gray[f], mask ? mask[f] : nullptr, f & 1);
      std::lock_guard<std::mutex> l(m);
      staged = f + 1;
      cv.notify_all();
    }
  });
  // whatever happens below (an exception on its way to the ABI barrier included): the helper is told to stop and joined
  struct HelperJoin {
    std::thread& th; std::mutex& m; std::condition_variable& cv; bool& stop;
    ~HelperJoin() {
      { std::lock_guard<std::mutex> l(m); stop = true; }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
  } helper_join{helper, m, cv, stop};
  // The calling thread: frame f + 1 is detected on ctx->stream (set (f + 1) & 1) while frame f is described on the second
  // stream (set f & 1) -- describe_enqueue(f) runs as the `before_wait` hook of frame f + 1's first detection pass, i.e.
  // its host work (removeDepthless, retainBest, the descriptor records) overlaps that pass's device time.
  hipStream_t st2 = ctx->orb_compute_stream;
  std::vector<DetectFrame> fr((size_t)2);
  auto init = [&](DetectFrame& d, int32_t f) {
    d = DetectFrame();
    d.ctx = ctx; d.gray = gray[f]; d.mask = mask ? mask[f] : nullptr; d.depth = depth[f]; d.rows = rows; d.cols = cols;
    d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy; d.depth_scaling = depth_scaling;
    d.keypoints = keypoints + (size_t)f * out_stride; d.descriptors = descriptors + (size_t)f * out_stride * 32;
    d.xyz1 = xyz1 + (size_t)f * out_stride * 4; d.n_out = n_out + f;
  };
  auto enqueue_upload = [&](int32_t f) -> int {  // the device half of frame f's upload, on `up`
    {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [&] { return staged > f; });
    }
    // the set was frame f - 2's: its description (second stream) reads the pyramid this upload overwrites
    if (f >= 2 && hipStreamWaitEvent(up, ctx->orb_describe_done[f & 1], 0) != hipSuccess) { err = "hipStreamWaitEvent"; return RGBDFE_ERR_HIP; }
    const int r = orb.enqueue_staged(mask != nullptr && mask[f] != nullptr, up, err, f & 1);
    if (r != RGBDFE_OK) return r;
    if (hipEventRecord(ctx->orb_upload_done[f & 1], up) != hipSuccess) { err = "hipEventRecord"; return RGBDFE_ERR_HIP; }
    return RGBDFE_OK;
  };
  auto wait_upload = [&](int32_t f) -> int {  // order ctx->stream behind frame f's upload and pyramid
    if (hipStreamWaitEvent(ctx->stream, ctx->orb_upload_done[f & 1], 0) != hipSuccess) { err = "hipStreamWaitEvent"; return RGBDFE_ERR_HIP; }
    return RGBDFE_OK;
  };
  auto mark_detected = [&](int32_t f) {
    std::lock_guard<std::mutex> l(m);
    detected = f + 1;
    cv.notify_all();
  };
  auto describe = [&](int32_t f) -> int {  // enqueue frame f's description on the second stream, from its own image set
    orb.use_set(f & 1);
    if (hipStreamWaitEvent(st2, ctx->orb_upload_done[f & 1], 0) != hipSuccess) return RGBDFE_ERR_HIP;
    const int r = fr[(size_t)(f & 1)].describe_enqueue(st2);
    if (r != RGBDFE_OK) return r;
    return hipEventRecord(ctx->orb_describe_done[f & 1], st2) == hipSuccess ? RGBDFE_OK : RGBDFE_ERR_HIP;
  };
  rc = enqueue_upload(0);
  if (rc == RGBDFE_OK) rc = wait_upload(0);
  if (rc == RGBDFE_OK) {
    init(fr[0], 0);
    orb.use_set(0);
    int rc_up = RGBDFE_OK;
    rc = fr[0].detect(true, n_frames > 1 ? std::function<int()>([&]() -> int { rc_up = enqueue_upload(1); return rc_up; })
                                         : std::function<int()>());
    if (rc != RGBDFE_OK && rc_up == RGBDFE_OK) err.clear();  // reported through fail()
    if (rc == RGBDFE_OK) mark_detected(0);
  }
  for (int32_t f = 0; f < n_frames && rc == RGBDFE_OK; ++f) {
    if (f + 1 < n_frames) {
      rc = wait_upload(f + 1);
      if (rc != RGBDFE_OK) break;
      init(fr[(size_t)((f + 1) & 1)], f + 1);
      orb.use_set((f + 1) & 1);
      static const bool overlap = !(getenv("RGBDFE_DETECT_OVERLAP") && atoi(getenv("RGBDFE_DETECT_OVERLAP")) == 0);  // A/B switch
      if (!overlap) {
        rc = describe(f);
        if (rc != RGBDFE_OK) { err.clear(); break; }
        orb.use_set((f + 1) & 1);
      }
      int rc_hook = RGBDFE_OK;
      bool hook_err_is_mine = false;
      rc = fr[(size_t)((f + 1) & 1)].detect(true, [&, f]() -> int {
        if (overlap) rc_hook = describe(f);
        if (rc_hook == RGBDFE_OK && f + 2 < n_frames) {
          rc_hook = enqueue_upload(f + 2);
          hook_err_is_mine = rc_hook != RGBDFE_OK;
        }
        orb.use_set((f + 1) & 1);  // the rest of the pass (a second read-back of a crowded frame) is frame f + 1's
        return rc_hook;
      });
      if (rc != RGBDFE_OK) { if (!hook_err_is_mine) err.clear(); break; }
      mark_detected(f + 1);
    } else {
      rc = describe(f);
      if (rc != RGBDFE_OK) { err.clear(); break; }
    }
    rc = fr[(size_t)(f & 1)].finish(st2);
    if (rc != RGBDFE_OK) { err.clear(); break; }
  }
  (void)hipStreamSynchronize(st2);
  {
    std::lock_guard<std::mutex> l(m);
    stop = true;
    cv.notify_all();
  }
  if (helper.joinable()) helper.join();
  (void)hipStreamSynchronize(up);
  orb.use_set(0);
  if (rc != RGBDFE_OK) return err.empty() ? rc : fail(ctx, rc, err);
  return RGBDFE_OK;
}


}  // namespace impl
