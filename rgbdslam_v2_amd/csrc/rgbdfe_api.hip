// rgbdfe_api.hip -- host side of the C ABI declared in include/rgbdfe.h.
//
// Owns the HBM-resident node slabs (descriptors + xyz1 of every node the graph keeps,
// GraphManager ownership semantics, graph_manager.h:156-161), the per-batch staging
// buffers and one HIP stream.  There is no CPU fallback: without a HIP device every
// entry point reports RGBDFE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "orb_host.h"
#include "sift_extract.h"
#include "rgbdfe_internal.h"

using namespace rgbdfe;

namespace {

// a frame's structured point cloud, resident for the environment measurement model
struct CloudEntry {
  float4* d = nullptr;  // ch x cw points, followed by the ch x cw depth plane (z only) the EMM gathers from
  int ch = 0, cw = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0;  // as getCameraIntrinsics assigns them (double -> float)
  int cloud_skip = 1;                    // cloud_creation_skip_step the cloud was built with
  float4* d_samples = nullptr;           // the points the EMM visits for emm skip step `samples_skip`, dense
  int samples_skip = 0;                  // 0: not built (invalidated by a re-upload)
};

// smallest double q with 0.5 * (1 + erf(q)) >= target under the host's libm (bisection)
inline double erf_boundary(double target) {
  double lo = -8.0, hi = 8.0;
  for (;;) {
    const double mid = lo + (hi - lo) * 0.5;
    if (!(mid > lo && mid < hi)) break;
    if (0.5 * (1 + std::erf(mid)) >= target) hi = mid; else lo = mid;
  }
  return hi;
}

// smallest double d with d / denom >= q (denom > 0): the exact pre-image of the test `d / denom < q`
inline double division_boundary(double q, double denom) {
  double c = q * denom;
  while (c / denom >= q) c = std::nextafter(c, -INFINITY);
  while (c / denom < q) c = std::nextafter(c, INFINITY);
  return c;
}

struct NodeEntry {
  uint32_t slot;
  uint32_t n;
  uint32_t kind;  // 0 = ORB (32-byte binary descriptors), 1 = SIFT (128 floats), 2 = float descriptors (FLANN branch)
  uint32_t flags = 0;  // bit 0 (SIFT): every row's quantised squared norm is < 2^19 (sift_match.hip's fast keys)
                       // bit 1: the slot holds THIS node's KeyPoint.pt (rgbdfe_upload_node_keypoints after the latest upload)
};
constexpr uint32_t kNodeHasKeypoints = 2u;

inline uint32_t mix32_host(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// RNG stream id of a pair: depends on the two node ids only, so a pair yields the same
// result in any batch, on any rank.
inline uint32_t pair_uid(int32_t qid, int32_t tid) {
  return mix32_host((uint32_t)qid * 0x9E3779B1u ^ ((uint32_t)tid + 0x7F4A7C15u));
}

}  // namespace

namespace {

class TaskPool {  // a few persistent worker threads for pure-CPU jobs
 public:
  explicit TaskPool(int n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
  }
  ~TaskPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) if (t.joinable()) t.join();
  }
  void submit(std::function<void()> f) {
    { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(f)); ++pending_; }
    cv_.notify_one();
  }
  void wait_all() {
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [&] { return pending_ == 0; });
  }
  int size() const { return (int)th_.size(); }
  // fn(0) .. fn(n - 1), the caller working too; returns when all are done (and everything else in the queue)
  void parallel_for(int n, const std::function<void(int)>& fn) {
    std::atomic<int> next{0};
    auto body = [&next, &fn, n] { for (;;) { const int i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    const int helpers = std::min(n - 1, size());
    for (int h = 0; h < helpers; ++h) submit(body);
    body();
    wait_all();
  }
 private:
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
        f = std::move(q_.front());
        q_.erase(q_.begin());
      }
      try { f(); } catch (...) { failed_ = true; }
      { std::lock_guard<std::mutex> l(m_); --pending_; }
      done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::vector<std::function<void()>> q_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  int pending_ = 0;
  bool stop_ = false;
 public:
  bool failed_ = false;
};

}  // namespace

// A stream that must run BESIDE the context's main stream gets another priority class: the runtime maps the streams of one
// priority onto a small pool of hardware queues (4 by default), and two streams that land on the same queue execute one
// after the other -- which two do depends on every stream the process created before (measured: the SIFT batch's second chunk
// stream shared the main stream's queue in a process that had run the ORB batch before, 0.24 instead of 0.18 ms per frame).
// Priority classes have their own queues.  which: -1 = the lowest, +1 = the highest priority the device offers.
static hipError_t create_side_stream(hipStream_t* s, int which) {
  int lo = 0, hi = 0;   // hipDeviceGetStreamPriorityRange: numerically lower = higher priority
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi) { (void)hipGetLastError(); return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, which > 0 ? hi : lo);
}

struct rgbdfe_ctx {
  rgbdfe_config cfg{};
  std::mutex mu;
  std::mutex err_mu;        // guards last_error only (fail() may run before `mu` is taken)
  std::string last_error;
  // multi-device group handle (rgbdfe_create_multi): the per-device contexts and one host thread per device
  struct Group* group = nullptr;
  hipStream_t stream = nullptr;
  // slabs
  uint32_t* d_desc = nullptr;  // max_nodes x max_kp x 8 dwords (+ pad rows)
  float4* d_xyz = nullptr;     // max_nodes x max_kp
  uint32_t* d_desc4 = nullptr; // max_nodes x max_kp x 32 dwords: every descriptor bit as an fp4 (+-1) operand nibble, in
                               // MFMA fragment order per tile of 32 rows (hamming_mfma.hip)
  float* d_kp2d = nullptr;     // max_nodes x max_kp x 2: KeyPoint.pt (allocated with the first rgbdfe_upload_node_keypoints)
  hipStream_t orb_upload_stream = nullptr;  // rgbdfe_detect_describe_batch: uploads of frame k+1 beside frame k
  hipStream_t orb_compute_stream = nullptr; // ... and frame k's description beside frame k+1's detection
  hipEvent_t orb_upload_done[OrbWorkspace::kSets] = {};
  hipEvent_t orb_describe_done[OrbWorkspace::kSets] = {};  // frame f's description has left its image set
  bool feature_min_depth = false;  // "use_feature_min_depth" (parameter_server.cpp:90): rgbdfe_set_feature_min_depth
  bool sift_fast = true;       // sift_match.hip's float keys where a pair qualifies (RGBDFE_SIFT_FAST_KEYS=0: never)
  int hamming_mode = RGBDFE_HAMMING_MODE_DEFAULT;   // rgbdfe.h: 0 = popcount kernel (hamming_nn.hip), 1 / 2 / 3 = fp4 MFMA kernels (hamming_mfma.hip)
  // Batches run on kLanes internal HIP streams ("lanes"), each with its own keys / results
  // staging, so that batch k+1's Hamming kernel fills the SIMDs that batch k's RANSAC tail
  // leaves idle.  The pair lists go through a ring of pinned buffers so the host can prepare
  // batch k+1 while batch k runs.
#ifndef RGBDFE_LANES
#define RGBDFE_LANES 2  // measured on the bench: 2 lanes 1.50 ms per step, 3 lanes 1.71, 4 lanes 1.50
#endif
  static constexpr int kLanes = RGBDFE_LANES;
  static constexpr int kRing = 2 * RGBDFE_LANES;
  struct Lane {
    hipStream_t stream = nullptr;
    IterRec* d_recs = nullptr;              // record / replay: per pair x RANSAC iteration outcome records
    size_t recs_capacity = 0;               // in records
    WalkState* d_walk = nullptr;            // record / replay: per pair progress (max_pairs)
    PairPrep* d_prep = nullptr;             // selected matches of every pair of the batch (max_pairs)
    double* d_ec = nullptr;                 // error pool of select+RANSAC: one region per launched wave
    size_t ec_regions = 0;
    uint32_t* d_keys = nullptr;             // max_pairs x max_kp
    rgbdfe_match_result* d_results = nullptr;  // staging for the host-output entry points
    // SIFT scratch (allocated with the first SIFT node)
    uint32_t* d_row_part = nullptr;   // max_pairs x max_kp x 3
    uint32_t* d_col_part = nullptr;   // max_pairs x max_kp x 3 (per train row)
    uint2* d_col_blocks = nullptr;    // max_pairs x sift_col_block_bytes_per_pair() (one-pass matcher: per-row-block column partials)
    uint16_t* d_sm_q = nullptr;       // max_pairs x max_kp
    uint16_t* d_sm_t = nullptr;
    float* d_sm_d = nullptr;
    int32_t* d_sm_n = nullptr;        // max_pairs
    float* d_all_dist = nullptr;      // max_pairs x RGBDFE_MAX_MATCHES
  };
  uint16_t* d_sift_bf16 = nullptr;  // max_nodes x max_kp x 128 (u8-quantised values as bf16)
  float* d_sift_f32 = nullptr;      // max_nodes x max_kp x 128 (raw descriptors)
  bool sift_ready = false;
  struct Slot {
    PairWork* h_work = nullptr;  // pinned
    PairWork* d_work = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false;
    bool failed = false;         // the batch with `ticket` did not launch completely
    int64_t ticket = 0;
  };
  Lane lanes[kLanes];
  Slot ring[kRing];
  // Batches of at most latency_pairs ORB pairs take the record / replay path (select_ransac.hip): the refinement
  // work of one pair is spread over ceil(ransac_iterations / latency_chunk_iters) waves.  0 disables it.
  // Measured (tools/bench_batch_sweep.py, bench.py --ransac-path): record / replay wins up to ~2000 pairs per batch
  // (uniform short waves fill the chip and have no straggler tail), the one-wave kernel above (it skips the
  // iterations the reference's early exits skip, and overlapped batches hide its tail).
  int32_t latency_pairs = INT32_MAX;  // record / replay for every batch size (rgbdfe_set_latency_mode)
  int32_t latency_chunk_iters = 0;  // 0 = automatic: 4 iterations per wave up to 64 pairs, 7 up to 640, 14 up to 1280, 28 above
  int64_t next_ticket = 1;
  // The launch chain of an ORB batch (pair-list upload, Hamming, pair_prep, recording / walk launches, result launch:
  // ~12 enqueues) as a hipGraph: captured once per distinct batch shape, then ONE hipGraphLaunch per batch -- what keeps
  // a single submitting thread ahead of several devices (rgbdfe_create_multi) and shortens the live-SLAM call.
  // Everything a kernel argument or a grid depends on is part of the key -- and nothing else: node sizes count only
  // through the Hamming stage's launch geometry (HammingGeometry), so frames with different keypoint counts share graphs.
  struct GraphKey {
    int32_t n; uint32_t qblocks, tsplit; int32_t slot, latency, chunk, hamming_mode, n_phases; int32_t ends[4];
    RansacConst rc;
    void* d_out; void* d_recs; void* d_ec; void* d_walk; void* d_keys;
  };
  struct GraphEntry { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; uint64_t used; };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int64_t graph_launches = 0, graph_captures = 0;
  int64_t graph_misses = 0;          // graphable batches whose shape was not cached
  int64_t graph_plain_batches = 0;   // of those: issued as plain launches without a capture attempt
  int64_t graph_launch_failures = 0; // cached executable graphs that failed to launch (dropped)
  int32_t graph_miss_run = 0;        // misses since the last hit
  static constexpr int32_t kGraphMissRun = 8, kGraphRetry = 16;
  // Off by default (round 4): while a capture is open, a device-wide synchronisation on ANY thread of the process fails with
  // hipErrorStreamCaptureUnsupported -- relaxed mode spares other threads' allocations and copies, not that -- and a
  // drop-in library may not make its host application's unrelated HIP calls fail (rgbdslam is a multi-threaded Qt / ROS
  // process).  A caller that owns every thread touching HIP turns it on: rgbdfe_set_graph_capture / RGBDFE_GRAPHS=1.
  bool use_graphs = false;
  hipStream_t capture_stream = nullptr;  // graphs are captured here, never on a stream other threads may wait on
  long graph_capture_failures = 0;       // captures another thread's HIP call invalidated (the batch then ran as plain launches)
  uint8_t* upload_stage = nullptr; size_t upload_stage_bytes = 0;  // pinned staging of rgbdfe_upload_nodes
  hipEvent_t ev_in = nullptr;  // orders a caller's stream before a lane
  hipEvent_t nodes_ready = nullptr;  // recorded behind the latest rgbdfe_upload_node_device copies; every batch waits for it
  hipEvent_t nodes_ready_ev = nullptr;  // (storage; nodes_ready points here once the first such upload happened)
  rgbdfe_match_result* h_results = nullptr;  // pinned staging of the synchronous host-output entry points
  // rgbdfe_submit_pair_list_host / rgbdfe_wait_host: one job per lane -- the results of the batch on lane li go device ->
  // pinned stage li (or straight into the caller's buffer when that is pinned) behind the batch, on the lane's stream, while
  // the other lane computes the next batch; the copy-out to pageable caller memory happens in rgbdfe_wait_host
  struct HostJob {
    bool pending = false;
    bool direct = false;          // the download went straight into the caller's (pinned / registered) buffer
    int payload = 0;              // RGBDFE_HOST_RECORDS / RGBDFE_HOST_INLIERS
    int64_t ticket = 0;
    int32_t n = 0;
    void* out = nullptr;
    size_t out_bytes = 0;
    hipEvent_t copied = nullptr;  // the download has ended (inlier payload: headers + the list block's length)
  };
  HostJob host_jobs[kLanes];
  uint8_t* h_stage[kLanes] = {};    // pinned: max_pairs records, or the largest inlier stream of max_pairs pairs
  uint8_t* d_inl_stream[kLanes] = {};  // inlier payload: the packed stream in HBM
  int32_t* d_inl_total[kLanes] = {};
  int32_t* h_inl_total[kLanes] = {};   // pinned
  // scratch for single-pair helpers / project_to_3d
  void* d_scratch = nullptr;
  size_t scratch_bytes = 0;
  OrbWorkspace orb;
  OrbWorkspace orb_super;  // rgbdfe_detect_describe_batch: up to 7 frames per launch chain (its own image sets)
  std::unique_ptr<TaskPool> detect_pool, stage_pool;  // its worker threads (created by the first batch call, kept)
  SiftExtractor sift2;          // rgbdfe_sift_detect_batch alternates between two extractors (two chunks in flight)
  hipStream_t sift_stream1 = nullptr, sift_stream2 = nullptr;
  SiftExtractor sift;  // rgbdfe_sift_detect (sift_extract.hip)
  int orb_max_keypoints = 0;  // 0 = detector not configured yet
  std::unordered_map<int32_t, NodeEntry> nodes;
  std::unordered_map<int32_t, CloudEntry> clouds;
  double emm_q_lo = 0.0, emm_q_hi = 0.0;  // cdf boundaries 0.001 / 0.999 as arguments of erf
  std::vector<uint32_t> free_slots;
  RansacConst rc{};
  // profiling
  bool profiling = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  double k_ms[RGBDFE_KERNEL_COUNT] = {};
  int64_t k_launches[RGBDFE_KERNEL_COUNT] = {};
  int64_t k_pairs[RGBDFE_KERNEL_COUNT] = {};
  // ORB batch: a -[hamming]- b -[ransac]- c ; SIFT batch: a -[dot]- b -[finish]- c -[ransac]- d
  struct Pending { hipEvent_t a, b, c, d; int32_t pairs; bool sift; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace {
// PairWork::pad bit 0 for a SIFT pair: dot products < 2^19 (Cauchy-Schwarz over the two nodes' squared norms) and at
// most 32 column tiles of 32 on either side -- see sift_row_top2_kernel
inline uint32_t sift_fast_keys(const rgbdfe_ctx* ctx, const NodeEntry& q, const NodeEntry& t) {
  return (ctx->sift_fast && (q.flags & t.flags & 1u) && q.n <= 1024u && t.n <= 1024u) ? 1u : 0u;
}
}  // namespace


namespace {

int fail(rgbdfe_ctx* ctx, int code, const std::string& msg) {
  if (ctx) {
    std::lock_guard<std::mutex> g(ctx->err_mu);
    ctx->last_error = msg;
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                          \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return fail(ctx, RGBDFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

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
struct PhasePlan { int ends[4]; int n_phases; };
// the error pool of the select+RANSAC launches of a lane: one region per wave of the largest grid
constexpr size_t kMaxEcRegions = (size_t)1 << 16;  // 1.2 GB
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
  if ((n <= 256 && !force_phases) || env_phases == 1) { plan->n_phases = 1; plan->ends[0] = I; }
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

int enqueue_pairs(rgbdfe_ctx* ctx, const int32_t* qids, const int32_t* tids, int32_t n,
                  rgbdfe_match_result* d_out, hipEvent_t wait_for, int64_t* ticket_out,
                  int* lane_out, int matcher = 0, float* d_out_dist = nullptr, double flann_ratio = 0.95) {
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

}  // namespace

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
  auto bail = [&](int code) { rgbdfe_destroy(ctx); return code; };
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
  for (hipStream_t st : {ctx->orb_upload_stream, ctx->orb_compute_stream, ctx->sift_stream1, ctx->sift_stream2})
    if (st) (void)hipStreamSynchronize(st);
  drain_pending(ctx);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& ge : ctx->graphs) { (void)hipGraphExecDestroy(ge.exec); (void)hipGraphDestroy(ge.graph); }
  if (ctx->capture_stream) (void)hipStreamDestroy(ctx->capture_stream);
  if (ctx->upload_stage) (void)hipHostFree(ctx->upload_stage);
  if (ctx->sift_stream1) (void)hipStreamDestroy(ctx->sift_stream1);
  if (ctx->sift_stream2) (void)hipStreamDestroy(ctx->sift_stream2);
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
static int upload_nodes_locked(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                               const float* const* xyz1, const int32_t* counts);
int rgbdfe_upload_nodes(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                        const float* const* xyz1, const int32_t* counts) {
  if (!ctx || n_nodes < 0 || (n_nodes > 0 && (!node_ids || !desc || !xyz1 || !counts)))
    return fail(ctx, RGBDFE_ERR_INVALID_ARG, "bad upload arguments");
  std::lock_guard<std::mutex> g(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->cfg.device_id));
  return upload_nodes_locked(ctx, n_nodes, node_ids, desc, xyz1, counts);
}

static int upload_nodes_locked(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
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
    HIP_TRY(ctx, hipMemcpy(ctx->d_kp2d + (size_t)it->second.slot * (size_t)ctx->cfg.max_keypoints * 2, kp_xy,
                           (size_t)n * 8, hipMemcpyHostToDevice));
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

// out_stride (in records): the multi-device group hands every device the interleaved positions of its shard
int rgbdfe_match_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids,
                           int32_t n_pairs, rgbdfe_match_result* out, int64_t out_stride = 1) {
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

// is `p` host memory the device can copy into asynchronously (hipHostMalloc / hipHostRegister)?
static bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
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
  job.direct = n_pairs > 0 && is_pinned_host(out);
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
  job.payload = payload;
  job.ticket = *ticket;
  job.n = n_pairs;
  job.out = out;
  job.out_bytes = out_bytes;
  return RGBDFE_OK;
}

int rgbdfe_wait_host(rgbdfe_ctx* ctx, int64_t ticket, int64_t* bytes_written) {
  if (!ctx) return RGBDFE_ERR_INVALID_ARG;
  rgbdfe_ctx::HostJob job;
  int li = -1;
  {
    std::lock_guard<std::mutex> g(ctx->mu);
    for (int k = 0; k < rgbdfe_ctx::kLanes; ++k)
      if (ctx->host_jobs[k].pending && ctx->host_jobs[k].ticket == ticket) li = k;
    if (li < 0) return fail(ctx, RGBDFE_ERR_INVALID_ARG, "rgbdfe_wait_host: no host job with this ticket");
    job = ctx->host_jobs[li];
  }
  // (the context is not locked while this thread waits and copies: another thread may submit the next batch meanwhile)
  hipError_t e = hipSetDevice(ctx->cfg.device_id);
  if (e == hipSuccess) e = hipEventSynchronize(job.copied);
  const size_t rec = sizeof(rgbdfe_match_result), hdr = sizeof(rgbdfe_inlier_header);
  size_t written = 0;
  int rc = RGBDFE_OK;
  if (e == hipSuccess && job.n > 0) {
    if (job.payload == RGBDFE_HOST_RECORDS) {
      written = rec * (size_t)job.n;
      if (!job.direct) memcpy(job.out, ctx->h_stage[li], written);
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
    ctx->host_jobs[li].pending = false;
  }
  if (bytes_written) *bytes_written = (int64_t)written;
  if (e != hipSuccess) return fail(ctx, RGBDFE_ERR_HIP, std::string("rgbdfe_wait_host: ") + hipGetErrorString(e));
  if (rc != RGBDFE_OK) return fail(ctx, rc, "rgbdfe_wait_host: the inlier stream does not fit the caller's buffer");
  return RGBDFE_OK;
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
                                int32_t n_pairs, rgbdfe_match_result* out, float* out_dist, int64_t out_stride = 1,
                                int matcher = 1, double flann_ratio = 0.95) {
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


// ---------------------------------------------------------------------------------------------
// per-frame feature path
// ---------------------------------------------------------------------------------------------
static void ensure_detector(rgbdfe_ctx* ctx) {
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

static void kp_to_abi(const std::vector<KpOut>& v, rgbdfe_keypoint* out) {
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
  std::vector<SiftKey> keys[SiftExtractor::kMaxBatch];
  const float* desc[SiftExtractor::kMaxBatch];
  std::string err;
  // Two extractors, two streams: the shape-static first half of chunk c + 1 (pyramids, extremum flags, candidate lists) is
  // enqueued before the host collects chunk c, so it runs on the device beside chunk c's orientation / descriptor launches and
  // behind the host's waits and list work.
  constexpr int B = SiftExtractor::kMaxBatch;
  const int32_t n_chunks = (n_frames + B - 1) / B;
  SiftExtractor* ex[2] = {&ctx->sift, &ctx->sift2};
  // (both chunk streams come from the high-priority class, created back to back: two queues of a pool nothing else in the
  // process is likely to use -- see create_side_stream; equal priority, so neither chunk starves the other)
  if (n_chunks > 1 && !ctx->sift_stream2) {
    HIP_TRY(ctx, create_side_stream(&ctx->sift_stream1, +1));
    HIP_TRY(ctx, create_side_stream(&ctx->sift_stream2, +1));
  }
  hipStream_t st[2] = {ctx->sift_stream1 ? ctx->sift_stream1 : ctx->stream, ctx->sift_stream2 ? ctx->sift_stream2 : ctx->stream};
  auto count_of = [&](int32_t c) { return std::min<int32_t>(B, n_frames - c * B); };
  if (n_chunks > 0) {
    const int rc = ex[0]->begin_batch(gray, count_of(0), rows, cols, st[0], err);
    if (rc != RGBDFE_OK) return fail(ctx, rc, err);
  }
  for (int32_t c = 0; c < n_chunks; ++c) {
    const int32_t f0 = c * B;
    const int nf = count_of(c);
    if (c + 1 < n_chunks) {
      const int rcb = ex[(c + 1) & 1]->begin_batch(gray + (size_t)(c + 1) * B, count_of(c + 1), rows, cols, st[(c + 1) & 1], err);
      if (rcb != RGBDFE_OK) { (void)hipStreamSynchronize(st[c & 1]); return fail(ctx, rcb, err); }
    }
    const int rc = ex[c & 1]->finish_batch(max_keypoints, keys, desc, st[c & 1], err);
    if (rc != RGBDFE_OK) { (void)hipStreamSynchronize(st[(c + 1) & 1]); return fail(ctx, rc, err); }
    for (int k = 0; k < nf; ++k) {
      const int32_t f = f0 + k;
      n_out[f] = (int32_t)keys[k].size();
      if ((int32_t)keys[k].size() > out_stride) { overflow = true; continue; }
      rgbdfe_keypoint* kp = keypoints + (size_t)f * out_stride;
      for (size_t i = 0; i < keys[k].size(); ++i) {
        kp[i].x = keys[k][i].x;
        kp[i].y = keys[k][i].y;
        kp[i].size = (float)(12.0 * keys[k][i].s);
        kp[i].angle = (float)(keys[k][i].o * 180.0 / 3.1415927);
        kp[i].response = 0.f;
        kp[i].octave = 0;
      }
      if (!keys[k].empty()) memcpy(desc128 + (size_t)f * out_stride * 128, desc[k], keys[k].size() * 128 * sizeof(float));
    }
  }
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
  const int rc = ctx->sift.debug_candidates(octave, dog_level, v);
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
  const int B = std::min(64 / pc, 7);
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
    // frame is detected.  Every fresh id is counted as needing a slot -- how many features a frame has is not known before
    // it is detected; a frame that ends up WITHOUT features registers nothing (a fresh id stays unknown, an existing node
    // keeps its features), so the check can refuse a batch that would have fitted by exactly that many slots.
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
  // helper thread: the caller's pageable images of super-frame s -> pinned staging buffer s % (D + 1), as soon as super-frame
  // s - (D + 1) (the buffer's previous user) has been detected (its upload from that buffer is complete then)
  std::mutex m;
  std::condition_variable cv;
  int staged = 0, detected = 0;
  bool stop = false;
  if (!ctx->stage_pool) ctx->stage_pool.reset(new TaskPool(4));
  if (!ctx->detect_pool || ctx->detect_pool->size() != B) ctx->detect_pool.reset(new TaskPool(B));  // one worker per frame of a super-frame
  TaskPool& stage_pool = *ctx->stage_pool;
  TaskPool& pool = *ctx->detect_pool;
  pool.failed_ = false;
  std::thread helper([&]() {
    for (int s = 0; s < S; ++s) {
      {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return stop || detected >= s - D; });
        if (stop) return;
      }
      // (2 W H bytes per frame through one core's memcpy would bound the whole pipeline: 75 us per 640 x 480 frame)
      for (int k = 0; k < count_of(s); ++k) {
        const int f = first_of(s) + k;
        stage_pool.submit([&orb, &gray, &mask, f, s, k, D] { orb.stage_image_at(gray[f], mask ? mask[f] : nullptr, s % (D + 1), k); });
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
    const int r = orb.enqueue_staged_super(count_of(s), up, err, s % D, s % (D + 1));
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
  }
  if (rc == RGBDFE_OK) rc = enqueue_describe(S - 1);
  if (rc == RGBDFE_OK) rc = finish(S - 1);
  pool.wait_all();
  (void)hipStreamSynchronize(st2);
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
                                 uint8_t* descriptors, float* xyz1, int32_t* n_out, const int32_t* node_ids = nullptr) {
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

int rgbdfe_abi_version(void) { return 5; }  // 5: rgbdfe_submit_pair_list_host / rgbdfe_wait_host (the refinement kernel's round-4 debug hooks are gone); 2: multi-device handles, rgbdfe_set_hamming_mode, RGBDFE_ERR_INTERNAL; 3: compact gather records, rgbdfe_sift_detect; 4: rgbdfe_graph_stats, rgbdfe_set_graph_capture, rgbdfe_pack_inliers, rgbdfe_match_pair_list_allgather_inliers

}  // namespace impl

// =====================================================================================================================
// Multi-device group (rgbdfe_create_multi) -- SURVEY.md 8(e), north_star "shard across the 8 GPUs of one node".
//
// The reference's caller is ONE process (GraphManager::nodeComparisons, graph_manager.cpp:541-548), so the drop-in
// form of "8 GPUs" lives behind the same handle: a group owns one ordinary context per device and one host thread per
// device.  Node features are replicated on every device (trivial in 288 GB), the pair list of a call is sharded
// pair k -> device k mod G, every device writes its results straight to the caller's positions k, k+G, ... and the
// call returns when all devices are done -- still the 1:1 replacement of blockingMapped's barrier.  For consumers on
// the GPUs, rgbdfe_match_pair_list_allgather leaves ALL results on EVERY device: one ncclAllGather (RCCL over xGMI,
// loaded with dlopen) of the fixed-size result PODs, or peer copies when RCCL cannot be used (the same device listed
// twice -- how the single-GPU test box exercises two shards -- or RGBDFE_GATHER=p2p).
// =====================================================================================================================
#include <dlfcn.h>

#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

namespace {

// the handful of RCCL entry points the gather needs, resolved at run time (no link-time dependency for 1-GPU users)
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    if (handle) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
  }
};
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar (rccl.h: ncclDataType_t)

struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, quit = false;
  int rc = RGBDFE_OK;
};

}  // namespace

struct Group {
  std::vector<rgbdfe_ctx*> children;
  std::vector<int> device_ids;
  std::vector<std::unique_ptr<Worker>> workers;
  Rccl rccl;
  std::vector<void*> comms;      // one communicator per device once the RCCL path has been set up
  bool rccl_tried = false, rccl_ok = false;
  std::vector<hipStream_t> gather_streams;  // one per device
  std::vector<hipEvent_t> gather_events;
  std::string transport = "none";
  // edges-only gather: per device a compacted copy of its shard, the survivors' global pair indices, scan scratch
  std::vector<rgbdfe_match_result*> edge_recs;
  std::vector<int32_t*> edge_idx, edge_dst, edge_cnt;
  std::vector<int32_t*> edge_cnt_host;  // pinned
  std::vector<char*> inl_stream;        // inlier gather: per device the shard's inlier stream (rgbdfe_inlier_header), worst case
  int32_t edge_cap = 0;                 // records per device the scratch holds
  double last_submit_us = 0.0;          // host time the calling thread spent enqueueing the latest sharded batch on all devices
  // One call at a time on a group handle (rgbdfe.h: calls on one context serialise): covers the workers' job slots and
  // transport / rccl_* / edge_* above.  Recursive: the gather entry points hold it around their group_run.
  std::recursive_mutex mu;
};

namespace {

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
// ORB shards that fit one batch are SUBMITTED BY THE CALLING THREAD, device after device (one hipGraphLaunch + one
// read-back enqueue each once the batch shape has been seen: host threads inside the HIP runtime at the same time
// serialise on its locks), then collected; everything else goes through the per-device worker threads.
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
    int first = RGBDFE_OK;
    const double t0 = orb_now_us();
    for (int i = 0; i < G; ++i) {
      rgbdfe_ctx* c = g.children[(size_t)i];
      const int32_t ni = (int32_t)qs[(size_t)i].size();
      if (ni == 0) continue;
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
      else if (first == RGBDFE_OK) {
        first = r;
        std::string msg; { std::lock_guard<std::mutex> e(c->err_mu); msg = c->last_error; }
        fail(gctx, first, "device " + std::to_string(g.device_ids[(size_t)i]) + ": " + msg);
      }
    }
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
                          int32_t* records_per_device, bool compact = false) {
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
  // 1. every device computes its shard into its own segment of its own buffer: enqueued by this thread, device after device
  const double t_sub0 = orb_now_us();
  int rc = group_each(gctx, [&](int i) -> int {
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
// of 1744.  Every device packs its shard (per headers + its lists) into scratch, the host learns the list lengths, and the
// exchange moves stride = per * 104 + 4 * max(length) bytes per device.  On return d_out[j] holds device i's stream at byte
// offset i * stride: pair k of the caller's list = header k / G of device k mod G.
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
  // 1. every device: its shard into scratch records, then the inlier stream
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
    HIP_TRY(c, hipStreamSynchronize(gs));
    return RGBDFE_OK;
  });
  if (rc != RGBDFE_OK) return rc;
  int32_t longest = 0;
  for (int i = 0; i < G; ++i) { totals[i] = *g.edge_cnt_host[(size_t)i]; longest = std::max(longest, totals[i]); }
  const size_t stride = hdr_bytes + (size_t)longest * 4;
  *stride_bytes = (int64_t)stride;
  // 2. the exchange
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
                                         g.device_ids[(size_t)i], hdr_bytes + (size_t)totals[i] * 4, g.gather_streams[(size_t)i]));
    }
  }
  for (int i = 0; i < G; ++i) {
    HIP_TRY(gctx, hipSetDevice(g.device_ids[(size_t)i]));
    HIP_TRY(gctx, hipStreamSynchronize(g.gather_streams[(size_t)i]));
  }
  return RGBDFE_OK;
}

// ---- the exception barrier: nothing thrown inside the library crosses the C ABI (node.cpp:1424 "never throws") ------
template <class F>
int guarded(rgbdfe_ctx* ctx, F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    try { return fail(ctx, RGBDFE_ERR_OUT_OF_MEMORY, "host allocation failed"); } catch (...) { return RGBDFE_ERR_OUT_OF_MEMORY; }
  } catch (const std::exception& e) {
    try { return fail(ctx, RGBDFE_ERR_INTERNAL, std::string("internal error: ") + e.what()); } catch (...) { return RGBDFE_ERR_INTERNAL; }
  } catch (...) {
    return RGBDFE_ERR_INTERNAL;
  }
}

int group_only_single(rgbdfe_ctx* ctx, const char* what) {
  return fail(ctx, RGBDFE_ERR_INVALID_ARG,
              std::string(what) + " takes device pointers: call it on one device's context (rgbdfe_device_context)");
}

}  // namespace

#define RGBDFE_IS_GROUP(ctx) ((ctx) && (ctx)->group)
// broadcast to every device of a group, or the plain call
#define RGBDFE_ALL(ctx, call_on_c)                                                          \
  guarded(ctx, [&]() -> int {                                                               \
    if (RGBDFE_IS_GROUP(ctx))                                                               \
      return group_run(ctx, [&](int i_) -> int { rgbdfe_ctx* c = ctx->group->children[(size_t)i_]; return call_on_c; }); \
    rgbdfe_ctx* c = ctx;                                                                    \
    return call_on_c;                                                                       \
  })
// frame-level work of a group runs on its first device
#define RGBDFE_FIRST(ctx, call_on_c)                                                        \
  guarded(ctx, [&]() -> int {                                                               \
    rgbdfe_ctx* c = RGBDFE_IS_GROUP(ctx) ? ctx->group->children[0] : ctx;                   \
    const int rc_ = call_on_c;                                                              \
    if (rc_ != RGBDFE_OK && RGBDFE_IS_GROUP(ctx)) {                                         \
      std::string m_; { std::lock_guard<std::mutex> e_(c->err_mu); m_ = c->last_error; }    \
      fail(ctx, rc_, m_);                                                                   \
    }                                                                                       \
    return rc_;                                                                             \
  })

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
