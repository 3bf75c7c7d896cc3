// rgbdfe_internal.h -- shared between the HIP kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rgbdfe.h"

namespace rgbdfe {

// One (newer node, older node) pair as the kernels see it.  Node features live in
// two slabs (descriptor slab: slot x max_kp x 32 B, xyz1 slab: slot x max_kp x 16 B);
// a pair is a couple of slot indices plus the row counts.
struct PairWork {
  uint32_t q_slot, t_slot;  // slab slots of the newer (query) / older (train) node
  uint32_t nq, nt;          // rows
  uint32_t uid;             // RNG stream id of the pair (D1), from (qid, tid)
  int32_t qid, tid;         // node ids (edge.id2 / edge.id1)
  uint32_t pad;
};
static_assert(sizeof(PairWork) == 32, "PairWork layout");

// "no neighbour": hd = 257, idx = 0xFFFF (features.cpp:172-173 -> (257,-1))
constexpr uint32_t kNoMatchKey = (257u << 16) | 0xFFFFu;

struct RansacConst {
  int32_t max_matches, min_matches, ransac_iterations;
  float max_dist_m;
  double sq_max_dist;   // (double)(max_dist_m*max_dist_m)  node.cpp:1152
  double depth_cov;     // D3
  double raster_cov_x;  // misc.cpp:708
  double raster_cov_y;  // misc.cpp:709
  uint32_t seed;
  int32_t g2o_iterations;  // "g2o_transformation_refinement" (parameter_server.cpp:103), 0 = off
};

// launchers (defined in the .hip files)
// returns the number of key planes written per pair (keys[pair][plane][row]); the consumer
// takes the min over planes.  key_planes_capacity = planes the keys buffer can hold in total.
// HammingGeometry = everything of a Hamming launch that depends on the batch's node sizes: query blocks per pair and
// train-row splits (key planes) per pair.  Batches with equal (n_pairs, geometry) are the same launch -- what the
// hipGraph cache of api_batches.hip keys on.
struct HammingGeometry { uint32_t qblocks, tsplit; };
HammingGeometry hamming_nn_geometry(uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_planes_capacity);
uint32_t launch_hamming_nn(const uint32_t* desc_pool, const PairWork* work, uint32_t* keys,
                           uint32_t max_kp, uint32_t n_pairs, HammingGeometry geom, hipStream_t stream);
// fp4 MFMA form of the same search (hamming_mfma.hip): descriptors expanded to one fp4 operand nibble per bit, 128 B per
// row in MFMA fragment order, tiles of 32 rows; same keys, bit for bit.  mode 1: row term added by the MFMA (C operand),
// mode 2: by v_add_f32.  Valid for max_kp <= 32768.
uint32_t hamming_mfma_tiles_per_slot(uint32_t max_kp);
size_t hamming_mfma_slab_bytes(uint32_t max_nodes, uint32_t max_kp);
void launch_hamming_expand(const uint32_t* node_rows, uint32_t* slab, uint32_t slot, uint32_t max_kp, uint32_t n,
                           hipStream_t stream);
HammingGeometry hamming_mfma_geometry(uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_planes_capacity);
uint32_t launch_hamming_mfma(const uint32_t* slab, const PairWork* work, uint32_t* keys, uint32_t max_kp,
                             uint32_t n_pairs, HammingGeometry geom, int mode, hipStream_t stream);
// place_recognition.hip: every query descriptor votes k - rank for the k candidate nodes holding its nearest matches
// (keys[candidate][plane][row] from the Hamming stage; k <= 8; candidates < 65536)
void launch_place_votes(const uint32_t* keys, uint32_t planes, uint32_t max_kp, const PairWork* work, const uint32_t* seg,
                        uint32_t n_queries, uint32_t max_nq, uint32_t k_neighbours, uint32_t max_hd, uint32_t* votes,
                        hipStream_t stream);
void launch_select_ransac(const float4* xyz_pool, const PairWork* work, const uint32_t* keys,
                          uint32_t key_planes, rgbdfe_match_result* results, uint32_t max_kp,
                          uint32_t n_pairs, const RansacConst& rc, struct PairPrep* prep, double* ec_pool,
                          hipStream_t stream);
// outcome of one RANSAC iteration's refinement loop (node.cpp:1140-1169): refined transform, inlier set, error
struct IterRec {
  float rR[9], rt[3];
  uint64_t rmask[5];
  double rerr;
  int32_t rn;
  int32_t pad;
};
// what the in-order walk needs of an iteration (refined_matches.size(), refined_error): a dense array next to the
// records, so that the walk reads -- and a batch of junk iterations writes, lane = iteration -- whole cache lines instead
// of one 104-byte record per lane (a junk iteration leaves only {1e6, 0} here and no IterRec at all)
struct IterSum {
  double rerr;
  int32_t rn;
  int32_t pad;
};
// what pair_prep_kernel leaves for every select+RANSAC wave of a pair: the selected matches' 3-D points as 7-word
// records (from.xyz, to.xyz, 1/(from.z*to.z)) and the facts about them the waves need
struct alignas(16) PairPrep {
  float M[RGBDFE_MAX_MATCHES * 7];
  uint64_t w_nonzero[RGBDFE_MASK_WORDS];  // matches with a non-zero weight
  int32_t n_all;                          // selected matches
  float pmax;                             // largest finite |coordinate| of their points
  uint32_t fast_alpha;                    // every weight inside the window of the unscaled float division
  uint32_t pad;
};
// per-pair progress of the record / replay schedule: the reference's in-order bookkeeping (node.cpp:1171-1190),
// resumed phase by phase by replay_walk_kernel
struct WalkState {
  int32_t state;             // >= 0: upper bound of the iterations that may still be needed; < 0: the loop has ended
  int32_t it, real_iterations, valid_iterations;
  int32_t best_idx;          // iteration whose record is the best hypothesis so far, -1 = none
  int32_t best_n;
  float rmse;
  int32_t speculate;         // class of the pair, set by the walk of the first phase: 0 = `it` has jumped ahead (a hypothesis
                             // with more than half of the matches as inliers: the loop may end early); 1 = no jump yet: the
                             // pair will most likely run all its iterations, the next recording launch records ALL of them;
                             // 2 = ... and at most 1/4 of the iterations produced a refined hypothesis ("junk-heavy")
};
// parameters of one record / replay phase (select_ransac.hip)
struct RecordPlan {
  IterRec* recs = nullptr;   // [pair][iteration]
  IterSum* sums = nullptr;   // [pair][iteration], behind the records in the same allocation
  WalkState* walk = nullptr; // [pair]
  uint32_t n_chunks = 1;     // recording waves per pair in this phase
  int chunk_iters = 0;       // iterations per recording wave
  int phase_begin = 0, phase_end = 0;
  int spec_end = 0;          // the launch's waves cover [phase_begin, spec_end); pairs of class 1 / 2 record beyond phase_end
  uint32_t n_chunks_b = 0;   // sub-grid B (class-2 pairs): recording waves per pair, in shares of chunk_iters_b iterations
  int chunk_iters_b = 0;
  int n_phases_total = 0;    // phases of the whole plan
  const PairPrep* prep = nullptr;  // [pair], every mode
  double* ec_pool = nullptr;  // every mode: select_ransac_ec_region_bytes() per launched wave (the inlier errors of
                              // a refinement round's scorings, read back lane = slot by the sequential error sums)
  // result waves of the split path: the walk over [WalkState::real_iterations, WalkState::speculate) is theirs
  int final_walk = 0;
  const uint64_t* vmask = nullptr;  // [pair][vmask_words]: SplitPlan::vmask
  int vmask_words = 0;
};
size_t select_ransac_ec_region_bytes();
// ransac_split.hip: the recording stage as a hypothesis kernel (lane = iteration) + ONE refinement launch over the viable
// iterations of the whole batch.
struct SplitPlan {
  IterRec* recs = nullptr;     // [pair][iteration]: the hypothesis kernel leaves a viable iteration's transform in rR / rt,
                               // the refinement kernel the iteration's outcome
  IterSum* sums = nullptr;     // [pair][iteration]
  uint64_t* vmask = nullptr;   // [pair][vmask_words]: bit k of a pair = iteration k passed the pre-screen
  WalkState* walk = nullptr;   // [pair] (+ the batch's counters in walk[n_pairs])
  const PairPrep* prep = nullptr;
  int vmask_words = 0;
  // phased = 0 (small batches, full speculation): a workgroup unit is (pair, share of share_iters iterations); everything is
  //          recorded, the result waves walk.
  // phased = 1: a unit is a pair.  Its range is recorded in WINDOWS inside the kernel -- [0, phase_ends[0]) first, or
  //          [0, I) at once for a pair the pre-screen shows to be junk-heavy -- and between two windows the workgroup's
  //          server runs the reference's in-order bookkeeping over what has been recorded: the loop has ended, or the next
  //          window (the next phase; everything that is left for a junk-heavy pair without a jump of `it`; never beyond
  //          the iterations the pair can still need).  The pair's match records stay in LDS across its windows.
  int phased = 0;
  int n_phases = 1;
  int phase_ends[4] = {0, 0, 0, 0};
  int n_shares = 1;            // phased = 0: units per pair
  int share_iters = 0;
  uint8_t* preclass = nullptr; // [pair]: 2 = at most 9 of the first 14 iterations passed the pre-screen ("junk-heavy" whatever
                               // their refinement gives: such a pair will most likely run all its iterations) -- written by the
                               // hypothesis kernel when preclass_iters > 0
  int preclass_iters = 0;      // the first phase's length (14), 0 = no pre-classification
  uint32_t* unit_counter = nullptr;  // zero at launch: the refinement kernel's workgroups take their units off it
  // phased plans: the pairs that run RANSAC, bucketed by the number of their viable iterations (the hypothesis kernel
  // appends; kOrderBuckets counters, zeroed by pair_prep_kernel, + [bucket][n_pairs] pair indices).  The refinement kernel
  // takes the buckets from the top: a launch is about two generations of resident units per workgroup, and the pairs that
  // keep a workgroup longest should not be the ones it picks up last.
  uint32_t* order_cnt = nullptr;
  uint32_t* order = nullptr;
};
constexpr int kOrderBuckets = 64;
// bucket of a pair with n viable iterations: one per count below 32, then steps of 8
__host__ __device__ inline int order_bucket(int n) { return n < 32 ? n : (32 + (n - 32) / 8 < kOrderBuckets ? 32 + (n - 32) / 8 : kOrderBuckets - 1); }
void launch_ransac_hyp(const PairWork* work, uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan, hipStream_t stream);
void launch_ransac_refine(uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan, hipStream_t stream);
int ransac_split_words_per_pair(int ransac_iterations);
int ransac_split_max_share();
int ransac_split_wgs();   // persistent workgroups of a refinement launch on this device
int ransac_split_init();  // once per process before the first batch (not inside a stream capture); returns the CU count
// bytes of the per-pair iteration masks behind the records + summaries of a record buffer of `rec_capacity` records
// (+ 8 bytes per pair: preclass; + the order buckets: kOrderBuckets counters and kOrderBuckets x max_pairs indices)
inline size_t ransac_split_mask_bytes(size_t rec_capacity, size_t max_pairs) { return (rec_capacity / 64 + (5 + 32) * max_pairs + 64 + 64) * 8; }
// where the order buckets of a batch of n_pairs live behind its records (the same for pair_prep_kernel, which zeroes the
// counters, and the plan of the kernels that use them)
inline uint32_t* ransac_split_order_cnt(IterRec* recs, size_t n_pairs, int ransac_iterations) {
  const size_t I = ransac_iterations > 0 ? (size_t)ransac_iterations : 0, n_recs = n_pairs * I;
  uint64_t* vmask = reinterpret_cast<uint64_t*>(reinterpret_cast<IterSum*>(recs + n_recs) + n_recs);
  uint8_t* preclass = reinterpret_cast<uint8_t*>(vmask + n_pairs * (size_t)ransac_split_words_per_pair(ransac_iterations));
  return reinterpret_cast<uint32_t*>(preclass + ((n_pairs + 8 + 7) & ~(size_t)7));
}
// edges.hip: stable compaction of the accepted edges (id1 >= 0) of a shard
void launch_compact_edges(const rgbdfe_match_result* in, uint32_t n, rgbdfe_match_result* out, int32_t* out_index,
                          int32_t index_scale, int32_t index_offset, int32_t* d_dst, int32_t* d_count, hipStream_t stream);
// edges.hip: records -> rgbdfe_compact_result (header + inlier mask, 144 of the 1744 bytes)
void launch_compact_pack(const rgbdfe_match_result* in, uint32_t n, rgbdfe_compact_result* out, hipStream_t stream);
// the inlier stream of a shard (include/rgbdfe.h: rgbdfe_inlier_header): n_headers headers, then the list block; *d_total = its entries
void launch_pack_inliers(const rgbdfe_match_result* in, uint32_t n, uint32_t n_headers, void* stream_out, int32_t* d_total,
                         hipStream_t stream);
// node.cpp:1222-1268 after the RANSAC results exist (rc.g2o_iterations > 0): two-view Gauss-Newton refinement over the
// inliers + re-scoring + the adopt rules.  kp_pool: KeyPoint.pt slab [slot][row] (float2).
void launch_g2o_refine(const PairWork* work, rgbdfe_match_result* results, uint32_t n_pairs, const RansacConst& rc,
                       const struct PairPrep* prep, const float* kp_pool, uint32_t max_kp, double* ec_pool, hipStream_t stream);
void launch_select_ransac_latency(const float4* xyz_pool, const PairWork* work, const uint32_t* keys,
                                  uint32_t key_planes, rgbdfe_match_result* results, uint32_t max_kp,
                                  uint32_t n_pairs, const RansacConst& rc, PairPrep* prep, IterRec* recs, WalkState* walk,
                                  double* ec_pool, int chunk_iters, const int* phase_ends, int n_phases, hipStream_t stream);
void launch_select_ransac_sift_latency(const float4* xyz_pool, const PairWork* work, uint16_t* sm_q,
                                       uint16_t* sm_t, float* sm_d, const int32_t* sm_n, float* all_dist,
                                       rgbdfe_match_result* results, uint32_t max_kp, uint32_t n_pairs,
                                       const RansacConst& rc, PairPrep* prep, IterRec* recs, WalkState* walk,
                                       double* ec_pool, int chunk_iters, const int* phase_ends, int n_phases, hipStream_t stream);
// (the SIFT launchers first sort each pair's match list in place: sift_sort_kernel)
void launch_select_ransac_sift(const float4* xyz_pool, const PairWork* work, uint16_t* sm_q,
                               uint16_t* sm_t, float* sm_d, const int32_t* sm_n,
                               float* all_dist, rgbdfe_match_result* results, uint32_t max_kp,
                               uint32_t n_pairs, const RansacConst& rc, struct PairPrep* prep, double* ec_pool,
                               hipStream_t stream);
// SIFT matcher (sift_match.hip): u8-quantised descriptors as bf16, exact integer dot products
// on the bf16 MFMA, SiftMatchGPU row/column/mutual-best semantics.
void launch_sift_dot(const uint16_t* bf16_pool, const PairWork* work, uint32_t max_kp,
                     uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_kinds,
                     uint32_t* row_part, uint32_t* col_part, uint2* col_blocks, hipStream_t stream);
void launch_sift_finish(const float* f32_pool, const PairWork* work, uint32_t max_kp,
                        uint32_t n_pairs, const uint32_t* row_part, uint32_t* col_part, const uint2* col_blocks,
                        uint16_t* sm_q, uint16_t* sm_t, float* sm_d, int32_t* sm_n,
                        hipStream_t stream);
// col_blocks (one-pass float-key path): per pair and 256-row block of the query node, the two largest dot products of every
// train column (float key bits); sift_col_block_bytes_per_pair() bytes per pair.  nullptr = the two-pass form.
size_t sift_col_block_bytes_per_pair();
// FLANN branch with exact neighbours (l2_knn.hip): knn[pair][row] = (d1 bits, d2 bits, nearest train row); then the
// ratio test + train-unique rule -> (queryIdx, trainIdx, ratio) lists in query order
void launch_l2_knn2(const float* f32_pool, const PairWork* work, uint32_t max_kp, uint32_t n_pairs, uint32_t max_nq,
                    uint32_t* knn, hipStream_t stream);
void launch_l2_ratio(const PairWork* work, uint32_t max_kp, uint32_t n_pairs, const uint32_t* knn, uint32_t* claim,
                     double max_ratio, uint16_t* sm_q, uint16_t* sm_t, float* sm_d, int32_t* sm_n, hipStream_t stream);
void launch_sift_quantise(const float* f32, uint16_t* bf16, size_t n_elems, hipStream_t stream);
void launch_project_to_3d(const float* kp_xy, int n_kp, const float* depth, int rows, int cols,
                          float fxinv, float fyinv, float cx, float cy, double depth_scaling,
                          int max_keypoints, int32_t* kept_idx, float4* xyz1, int32_t* n_out,
                          hipStream_t stream, bool truncate = false, const float* z_gathered = nullptr);
constexpr int kProjectFramesMax = 32;   // frames of a super-frame (api_detect.hip)
struct ProjectFrames { int n_frames; int off[kProjectFramesMax]; int n[kProjectFramesMax]; };
void launch_project_to_3d_frames(const ProjectFrames& fr, const float* kpxy, int rows, int cols, float fxinv, float fyinv,
                                 float cx, float cy, double depth_scaling, int max_keypoints, int32_t* kept_idx, float4* xyz1,
                                 int32_t* n_out, hipStream_t stream);
// kp_xys: n_kp x (x, y, KeyPoint::size); z_out[i] = getMinDepthInNeighborhood (misc.cpp:774-793) or NaN
void launch_min_depth(const float* kp_xys, int n_kp, const float* depth, int rows, int cols, float* z_out,
                      hipStream_t stream);
void launch_project_cloud(const float* kp_xy, int n_kp, const float4* pts, bool gathered, int rows, int cols,
                          double maximum_depth, int max_keypoints, int32_t* kept_idx, float4* xyz1, int32_t* n_out,
                          hipStream_t stream);
// one direction of one edge for the environment measurement model (emm.hip)
struct EmmJob {
  const float4* new_samples;  // the sampled points (every skip-th row / column) of the frame that is projected
  const float* old_z;       // ... into this frame's raster: its depth plane (z of every cloud point)
  float T[12];              // rows 0..2 of the 4x4 new -> old transform, row-major
  float fx, fy, cx, cy;     // old camera intrinsics, already divided by cloud_creation_skip_step
};
void launch_depth_to_mono8_f32(const float* depth, size_t n, uint8_t* mono8, hipStream_t stream);
void launch_depth_u16(const uint16_t* depth_mm, size_t n, uint8_t* mono8, float* depth_m, hipStream_t stream);
void launch_create_cloud(const float* depth, int rows, int cols, const uint8_t* rgb, int channels,
                         int encoding_bgr, float fxinv, float fyinv, float cx, float cy, double depth_scaling,
                         float min_depth, int s, int ch, int cw, float4* cloud, float* zplane, hipStream_t stream);
void launch_decimate_cloud(const float4* cloud, int ch, int cw, int skip_step, float4* out, hipStream_t stream);
void launch_emm(const EmmJob* jobs, int n_jobs, int ch, int cw, int skip_step, double d_lo, double d_hi,
                uint32_t* counts, hipStream_t stream);
void launch_sift_pack(const float* desc_in, const int32_t* kept_idx, const int32_t* n_ptr, int max_rows,
                      bool root_sift, float* raw, float* feat, hipStream_t stream);

}  // namespace rgbdfe
