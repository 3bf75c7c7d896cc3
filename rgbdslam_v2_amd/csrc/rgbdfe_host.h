// rgbdfe_host.h -- what the host-side translation units of librgbdfe.so share: the context (node store, lanes, graph cache,
// staging), the batch machinery's entry points (api_batches.hip), the single-device implementation of every entry point
// (namespace impl: api_context / api_pairs / api_detect / api_frame.hip) and the multi-device group (api_group.hip).
// rgbdfe_api.hip holds the extern "C" layer only.
#pragma once
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

namespace rgbdfe_host {

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

}  // namespace rgbdfe_host
using namespace rgbdfe_host;

namespace rgbdfe_host {

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

}  // namespace rgbdfe_host
using namespace rgbdfe_host;

// A stream that must run BESIDE the context's main stream gets another priority class: the runtime maps the streams of one
// priority onto a small pool of hardware queues (4 by default), and two streams that land on the same queue execute one
// after the other -- which two do depends on every stream the process created before (measured: the SIFT batch's second chunk
// stream shared the main stream's queue in a process that had run the ORB batch before, 0.24 instead of 0.18 ms per frame).
// Priority classes have their own queues.  which: -1 = the lowest, +1 = the highest priority the device offers.
inline hipError_t create_side_stream(hipStream_t* s, int which) {
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
    bool waiting = false;         // a thread is inside rgbdfe_wait_host for this job (one waiter per job)
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
  SiftExtractor sift2, sift3;   // rgbdfe_sift_detect_batch rotates over three extractors (three chunks in flight)
  hipStream_t sift_stream1 = nullptr, sift_stream2 = nullptr, sift_stream3 = nullptr;
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

namespace rgbdfe_host {
// PairWork::pad bit 0 for a SIFT pair: dot products < 2^19 (Cauchy-Schwarz over the two nodes' squared norms) and at
// most 32 column tiles of 32 on either side -- see sift_row_top2_kernel
inline uint32_t sift_fast_keys(const rgbdfe_ctx* ctx, const NodeEntry& q, const NodeEntry& t) {
  return (ctx->sift_fast && (q.flags & t.flags & 1u) && q.n <= 1024u && t.n <= 1024u) ? 1u : 0u;
}
}  // namespace rgbdfe_host


using namespace rgbdfe_host;

// ---- batch machinery (api_batches.hip)
namespace rgbdfe_host {
struct PhasePlan { int ends[4]; int n_phases; };
constexpr size_t kMaxEcRegions = (size_t)1 << 16;  // 1.2 GB
int fail(rgbdfe_ctx* ctx, int code, const std::string& msg);
void fill_ransac_const(rgbdfe_ctx* ctx);
int validate_params(rgbdfe_ctx* ctx, const rgbdfe_params& p);
int ensure_scratch(rgbdfe_ctx* ctx, size_t bytes);
hipEvent_t get_event(rgbdfe_ctx* ctx);
void drain_pending(rgbdfe_ctx* ctx);
int ensure_ec_pool(rgbdfe_ctx* ctx, rgbdfe_ctx::Lane& lane, size_t regions, hipStream_t stream);
int want_latency_path(rgbdfe_ctx* ctx, rgbdfe_ctx::Lane& lane, int32_t n, hipStream_t stream, bool* use, int* chunk_out, PhasePlan* plan);
bool hamming_on_mfma(const rgbdfe_ctx* ctx);
HammingGeometry hamming_geometry(const rgbdfe_ctx* ctx, uint32_t n, uint32_t max_nq, uint32_t max_nt);
uint32_t launch_hamming(rgbdfe_ctx* ctx, const PairWork* d_work, uint32_t* d_keys, uint32_t n, HammingGeometry geom, hipStream_t stream);
uint32_t launch_hamming(rgbdfe_ctx* ctx, const PairWork* d_work, uint32_t* d_keys, uint32_t n, uint32_t max_nq, uint32_t max_nt, hipStream_t stream);
bool capture_stream_ready(rgbdfe_ctx* ctx);
int enqueue_pairs(rgbdfe_ctx* ctx, const int32_t* qids, const int32_t* tids, int32_t n, rgbdfe_match_result* d_out, hipEvent_t wait_for, int64_t* ticket_out, int* lane_out, int matcher = 0, float* d_out_dist = nullptr, double flann_ratio = 0.95);
int wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, hipStream_t stream);
}  // namespace rgbdfe_host
#define HIP_TRY(ctx, expr)                                                          \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess)                                                           \
      return fail(ctx, RGBDFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// ---- the single-device implementation of the entry points (api_context / api_pairs / api_detect / api_frame.hip)
namespace impl {
// helpers shared between the implementation files
int upload_nodes_locked(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc,
                        const float* const* xyz1, const int32_t* counts);
void ensure_detector(rgbdfe_ctx* ctx);
void kp_to_abi(const std::vector<KpOut>& v, rgbdfe_keypoint* out);
void rgbdfe_default_config(rgbdfe_config* cfg);
int rgbdfe_create(const rgbdfe_config* cfg, rgbdfe_ctx** out);
void rgbdfe_destroy(rgbdfe_ctx* ctx);
int rgbdfe_set_params(rgbdfe_ctx* ctx, const rgbdfe_params* p);
const char* rgbdfe_status_string(int status);
const char* rgbdfe_last_error(rgbdfe_ctx* ctx);
int rgbdfe_upload_node(rgbdfe_ctx* ctx, int32_t node_id, const uint8_t* desc, const float* xyz1, int32_t n);
int rgbdfe_upload_nodes(rgbdfe_ctx* ctx, int32_t n_nodes, const int32_t* node_ids, const uint8_t* const* desc, const float* const* xyz1, const int32_t* counts);
int rgbdfe_upload_node_device(rgbdfe_ctx* ctx, int32_t node_id, const void* d_desc, const void* d_xyz1, int32_t n, void* stream);
int rgbdfe_upload_node_keypoints(rgbdfe_ctx* ctx, int32_t node_id, const float* kp_xy, int32_t n);
int rgbdfe_release_node(rgbdfe_ctx* ctx, int32_t node_id);
int rgbdfe_node_count(rgbdfe_ctx* ctx, int32_t node_id);
int rgbdfe_match_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, rgbdfe_match_result* out, int64_t out_stride = 1);
int rgbdfe_match_node_pairs(rgbdfe_ctx* ctx, int32_t new_node_id, const int32_t* candidate_ids, int32_t n_pairs, rgbdfe_match_result* out);
int rgbdfe_match_pair_list_device(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, void* d_out, void* stream);
int rgbdfe_submit_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, void* d_out, int64_t* ticket);
int rgbdfe_wait_ticket(rgbdfe_ctx* ctx, int64_t ticket, void* stream);
int rgbdfe_submit_pair_list_host(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, void* out, size_t out_bytes, int payload, int64_t* ticket);
int rgbdfe_wait_host(rgbdfe_ctx* ctx, int64_t ticket, int64_t* bytes_written);
int rgbdfe_wait_host_into(rgbdfe_ctx* ctx, int64_t ticket, void* out, size_t out_bytes, int64_t* bytes_written);
int rgbdfe_synchronize(rgbdfe_ctx* ctx);
int rgbdfe_upload_sift_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc128, const float* xyz1, int32_t n);
int rgbdfe_upload_float_node(rgbdfe_ctx* ctx, int32_t node_id, const float* desc, int32_t dim, const float* xyz1, int32_t n);
int rgbdfe_match_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, rgbdfe_match_result* out, float* out_dist, int64_t out_stride = 1, int matcher = 1, double flann_ratio = 0.95);
int rgbdfe_submit_sift_pair_list(rgbdfe_ctx* ctx, const int32_t* query_ids, const int32_t* train_ids, int32_t n_pairs, void* d_out, void* d_out_dist, int64_t* ticket);
int rgbdfe_sift_match_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* match_q, int32_t* match_t, float* match_dist, int32_t* n_matches);
int rgbdfe_detector_configure(rgbdfe_ctx* ctx, int32_t max_keypoints, int32_t grid_resolution, int32_t adjuster_max_iterations);
int rgbdfe_detector_thresholds(rgbdfe_ctx* ctx, double* thresholds, int32_t* n_cells);
int rgbdfe_orb_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, int32_t rows, int32_t cols, int32_t fast_threshold, rgbdfe_keypoint* keypoints, int32_t capacity, int32_t* n_out);
int rgbdfe_orb_compute(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints, int32_t n, uint8_t* descriptors, int32_t* n_out);
int rgbdfe_sift_detect(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* /*mask*/, int32_t rows, int32_t cols, int32_t max_keypoints, rgbdfe_keypoint* keypoints, float* desc128, int32_t capacity, int32_t* n_out);
int rgbdfe_sift_describe(rgbdfe_ctx* ctx, const uint8_t* gray, int32_t rows, int32_t cols, rgbdfe_keypoint* keypoints, int32_t n, float* desc128);
int rgbdfe_sift_detect_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, int32_t rows, int32_t cols, int32_t max_keypoints, int32_t out_stride, rgbdfe_keypoint* keypoints, float* desc128, int32_t* n_out);
int rgbdfe_sift_debug_plane(rgbdfe_ctx* ctx, int32_t octave, int32_t level, float* out, int32_t capacity_floats, int32_t* w, int32_t* h);
int rgbdfe_sift_debug_candidates(rgbdfe_ctx* ctx, int32_t octave, int32_t dog_level, float* out, int32_t capacity_rows, int32_t* n);
int rgbdfe_sift_geometry(rgbdfe_ctx* ctx, int32_t* octave_min, int32_t* octave_num, int32_t* levels, int32_t* dog_levels);
int rgbdfe_detect_describe(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling, rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out);
int rgbdfe_detect_describe_batch(rgbdfe_ctx* ctx, int32_t n_frames, const uint8_t* const* gray, const uint8_t* const* mask, const float* const* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling, int32_t out_stride, rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out, const int32_t* node_ids = nullptr);
int rgbdfe_hamming_nn_nodes(rgbdfe_ctx* ctx, int32_t query_id, int32_t train_id, int32_t* out_hd, int32_t* out_idx);
int rgbdfe_place_recognition_batch(rgbdfe_ctx* ctx, const int32_t* query_ids, int32_t n_queries, const int32_t* candidate_offsets, const int32_t* candidate_ids, int32_t k_neighbours, int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores, int32_t* out_counts);
int rgbdfe_place_recognition(rgbdfe_ctx* ctx, int32_t query_id, const int32_t* candidate_ids, int32_t n_candidates, int32_t k_neighbours, int32_t max_hd, int32_t max_out, int32_t* out_ids, float* out_scores, int32_t* n_out);
int rgbdfe_hamming_nn_host(rgbdfe_ctx* ctx, const uint8_t* qdesc, int32_t nq, const uint8_t* tdesc, int32_t nt, int32_t* out_hd, int32_t* out_idx);
int rgbdfe_project_to_3d(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling, int32_t max_keypoints, int32_t* kept_idx, float* xyz1, int32_t* n_out);
int rgbdfe_set_feature_min_depth(rgbdfe_ctx* ctx, int32_t on);
int rgbdfe_project_to_3d_min_depth(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp, const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling, int32_t max_keypoints, int32_t* kept_idx, float* xyz1, int32_t* n_out);
int rgbdfe_project_to_3d_cloud(rgbdfe_ctx* ctx, const float* kp_xy, int32_t n_kp, const float* cloud, int32_t rows, int32_t cols, double maximum_depth, int32_t max_keypoints, int32_t* kept_idx, float* xyz1, int32_t* n_out);
int rgbdfe_detect_describe_cloud(rgbdfe_ctx* ctx, const uint8_t* gray, const uint8_t* mask, const float* cloud, int32_t rows, int32_t cols, double maximum_depth, rgbdfe_keypoint* keypoints, uint8_t* descriptors, float* xyz1, int32_t* n_out);
int rgbdfe_sift_node_features(rgbdfe_ctx* ctx, const float* kp_xy, const float* kp_size, int32_t n_kp, const float* desc_in, const float* depth, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double depth_scaling, int32_t max_keypoints, int32_t use_root_sift, int32_t* kept_idx, float* xyz1, float* siftgpu_descriptors, float* feature_descriptors, int32_t* n_out);
int rgbdfe_depth_to_mono8(rgbdfe_ctx* ctx, const void* depth, int32_t depth_is_u16, int32_t rows, int32_t cols, uint8_t* mono8, float* depth_m);
int rgbdfe_upload_node_cloud(rgbdfe_ctx* ctx, int32_t node_id, const float* depth, int32_t rows, int32_t cols, const uint8_t* rgb, int32_t rgb_channels, int32_t encoding_bgr, double fx, double fy, double cx, double cy, double depth_scaling, double min_depth, int32_t cloud_skip, float* cloud_out);
int rgbdfe_release_node_cloud(rgbdfe_ctx* ctx, int32_t node_id);
int rgbdfe_observation_likelihood(rgbdfe_ctx* ctx, int32_t n, const int32_t* new_ids, const int32_t* old_ids, const float* transforms, int32_t emm_skip_step, rgbdfe_emm_counts* out);
int rgbdfe_observation_criterion_met(uint32_t inliers, uint32_t outliers, uint32_t all, double observability_threshold, double* quality);
int rgbdfe_set_latency_mode(rgbdfe_ctx* ctx, int32_t max_pairs, int32_t chunk_iterations);
int rgbdfe_set_hamming_mode(rgbdfe_ctx* ctx, int32_t mode);
int rgbdfe_set_profiling(rgbdfe_ctx* ctx, int enable);
int rgbdfe_get_kernel_time(rgbdfe_ctx* ctx, int which, double* total_ms, int64_t* launches, int64_t* pairs);
int rgbdfe_reset_kernel_time(rgbdfe_ctx* ctx);
int rgbdfe_sizeof_match_result(void);
int rgbdfe_sizeof_compact_result(void);
int rgbdfe_pack_compact(rgbdfe_ctx* ctx, const void* d_records, int32_t n, void* d_compact, void* stream);
int rgbdfe_set_graph_capture(rgbdfe_ctx* ctx, int enable);
int rgbdfe_pack_inliers(rgbdfe_ctx* ctx, const void* d_records, int32_t n, int32_t n_headers, void* d_stream, int32_t* d_total, void* stream);
int rgbdfe_sizeof_inlier_header(void);
int rgbdfe_graph_stats(rgbdfe_ctx* ctx, int64_t* out, int32_t n_out);
int rgbdfe_abi_version(void);
}  // namespace impl

// ---- several devices behind one handle (api_group.hip)
#include <dlfcn.h>
namespace rgbdfe_host {

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

}  // namespace rgbdfe_host

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
  // inlier gather: entries of a device's list block the exchange is sized for BEFORE the devices have counted their lists
  // (the longest list of the earlier calls + a quarter; 0: nothing seen yet), and how many exchanges the latest call issued
  size_t inl_cap_entries = 0;
  int inl_exchanges = 0;
  double last_submit_us = 0.0;          // host time the calling thread spent enqueueing the latest sharded batch on all devices
  // One call at a time on a group handle (rgbdfe.h: calls on one context serialise): covers the workers' job slots and
  // transport / rccl_* / edge_* above.  Recursive: the gather entry points hold it around their group_run.
  std::recursive_mutex mu;
};
namespace rgbdfe_host {
void worker_main(Worker* w);
int group_run(rgbdfe_ctx* gctx, const std::function<int(int)>& fn);
void group_destroy(rgbdfe_ctx* gctx);
int group_create(const rgbdfe_config* cfg, const int32_t* device_ids, int32_t n, rgbdfe_ctx** out);
int group_match(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, rgbdfe_match_result* out, bool sift, float* out_dist);
int group_each(rgbdfe_ctx* gctx, const std::function<int(int)>& fn);
int group_submit(rgbdfe_ctx* gctx, const std::function<int(int)>& fn);
bool group_setup_rccl(rgbdfe_ctx* gctx);
int group_ensure_edge_scratch(rgbdfe_ctx* gctx, int32_t per);
int group_match_allgather(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out, int32_t* records_per_device, bool compact = false);
int group_match_allgather_edges(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out, int32_t* const* d_index, int32_t* counts, int32_t* stride_out);
int group_match_allgather_inliers(rgbdfe_ctx* gctx, const int32_t* q, const int32_t* t, int32_t n, void* const* d_out, int32_t* records_per_device, int32_t* totals, int64_t* stride_bytes);
int group_only_single(rgbdfe_ctx* ctx, const char* what);
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
}  // namespace rgbdfe_host
using namespace rgbdfe_host;
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
