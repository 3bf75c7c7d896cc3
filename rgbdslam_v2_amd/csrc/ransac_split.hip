// ransac_split.hip -- the recording stage of the record / replay RANSAC schedule as two kernels (gfx950).
//
// Replaces, per (newer node, older node) pair, the body of the RANSAC loop of Node::getRelativeTransformationTo
// (node.cpp:1130-1169): sample_matches_prefer_by_distance (node.cpp:1024-1047), getTransformFromMatches
// (transformation_estimation_euclidean.cpp:7-61), computeInliersAndError + errorFunction2 (node.cpp:968-1020,
// misc.cpp:697-770) and the refinement loop (node.cpp:1140-1169).  The in-order bookkeeping (node.cpp:1171-1190, walk_records
// in ransac_device.h) runs between a pair's WINDOWS inside the refinement kernel -- a phased plan records a pair's iteration
// range window by window in ONE launch and stops where the reference's loop stops -- and is finished by the result waves,
// select_ransac_kernel<kReplay> (select_ransac.hip).  Five launches per batch: Hamming, pair_prep, the two kernels here, result.
//
// An iteration's outcome is a pure function of its index (counter-based sampling, D1), so the work of a batch is cut by
// WHAT IS DONE, not by pair:
//   ransac_hyp_kernel     LANE = ITERATION, one 256-thread workgroup per pair, all iterations of the pair at once: the
//                         4-point sample, the weighted fit, the 3x3 Jacobi SVD and the pre-screen (an upper bound of the
//                         matches that can pass errorFunction2's shortcut test; fewer than the inlier threshold => the
//                         iteration certainly ends with refined_matches empty, node.cpp:1148-1154).  Leaves, per pair, a
//                         bit mask of the viable iterations and their transforms (in the rR / rt fields of the iteration's
//                         record), the pair's initial walk state, and its place in the ORDER BUCKETS (pairs by their number of
//                         viable iterations: the refinement launch takes the fullest first).  No slot machinery, no scoring state.
//   ransac_refine_kernel  the refinement loops of the viable iterations only.  Persistent workgroups of 8 waves =
//                         7 WORKERS + 1 SERVER; up to four units (a pair, or a share of its range) are resident in LDS (the pair's
//                         PairPrep: match records + facts, 9 KB each) and their viable iterations occupy SLOTS (7 per
//                         worker and group).  The slots form two groups that take turns: in a half-round the workers run
//                         one pass of the refinement loop (node.cpp:1140) for the slots of group g
//                           scoring          LANE = MATCH; the inliers' errors stay in LDS and the reference's strictly
//                                            sequential error sum (node.cpp:1006) runs right behind the scoring.  Cheap
//                                            passes (junk hypotheses end after the float prefilter): every worker scores
//                                            the slots the server dealt it.  Expensive passes (>= 24 slots with a refined set: ~140
//                                            Cholesky solves per scoring): all 8 waves take the group's scorings one by
//                                            one off a ticket counter (ds_add_rtn) and meet at a barrier behind them --
//                                            fixed shares left most waves waiting for the one with the good hypotheses,
//                           bookkeeping      LANE = SLOT (node.cpp:1154-1166),
//                           refit            the PCL weighted-mean recurrences of the wave's slots side by side
//                                            (9 state elements per slot, 63 lanes),
//                         while the server does for the OTHER group everything that concerns more than one wave:
//                           3x3 Jacobi SVD   LANE = SLOT OF THE GROUP: one SVD for every refit the workers left in the
//                                            slots' mailboxes in the half-round before (at 7 lanes per wave the SVDs were
//                                            40 % of the recording stage's instructions),
//                           recycling        finished iterations' slots and drained units' buffers,
//                           hand-out         the next viable iterations of the resident units (with their 4-point
//                                            hypotheses) to the free slots,
//                           the deal         which worker scores, books and refits which slots of the group's next pass
//                                            (a slot belongs to nobody): the slots that already hold a refined set -- a
//                                            scoring ~4x that of a junk hypothesis -- go round the workers first, the
//                                            cheap ones level the rest; closed form, lane = slot,
//                           unit loading     the launch's units come off a global counter a block at a time (lane = unit:
//                                            units whose pair has ended cost nothing); PairPrep, the words of the viable
//                                            mask and the next block's facts go global -> LDS directly (global_load_lds),
//                                            issued in one half-round and awaited in the next.
//                         s_barrier -- one per half-round, two when the scorings go out by ticket -- is the ONLY way a
//                         wave of this kernel ever waits for another: the queue state is touched by the server alone, a
//                         slot by one wave at a time, the ticket is a single returning add.  No spin wait, no lock, no
//                         polling (round 4's streaming version handed work out through LDS spin locks and stalled about
//                         twice in 10^4 small launches beside context churn): a launch cannot wait for
//                         anything but its own waves' arrival at a hardware barrier.
//                         LDS: 80 KB per workgroup (four resident units) => 2 workgroups = 16 waves per CU at <= 128 VGPRs
//                         (4 per SIMD).
// Same bytes as the one-wave kernel (select_ransac_kernel<kWhole>): every float / double operation is the one
// oracle/rgbd_oracle.c performs, in the same order (-ffp-contract=off), so every discrete RANSAC decision is the same.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "ransac_device.h"

namespace rgbdfe {

namespace {

// wave-local ordering of LDS traffic: the LDS unit executes one wave's operations in order, the compiler must not move
// memory operations across this point, and results of earlier reads are in registers
__device__ __forceinline__ void lsync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The workgroup barrier of the refinement kernel: this wave's LDS operations have been performed, then s_barrier.  Vector
// memory is NOT drained (the server's unit loads and everybody's record stores stay in flight across it) -- which is why
// this is not __syncthreads(), in front of which the compiler waits for every counter.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// a lane index (or anything derived from it) the compiler may not carry across this point: addresses formed from it are
// recomputed where they are used instead of being hoisted to the top of the kernel and spilled
__device__ __forceinline__ int fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

constexpr int kHypThreads = 256;

// Diagnostics build only (-DRGBDFE_SPLIT_STATS): what the refinement kernel's servers saw, summed over workgroups and launches
// [0] half-rounds, [1] workgroups, [2] half-rounds scored by ticket, [3] scorings, [4] SVD requests, [5] units loaded,
// [6] hand-outs, [7] longest run of half-rounds of a workgroup
// [20] iterations ended, [21] ... after their first scoring, [22] ... with refined_matches empty
// [23] recurrence passes (a worker's refits of a half-round), [24] their steps (longest list), [25] their refits,
// [26] scorings that ran pass 2, [27] pass-2 rounds of 64 candidates, [28] error-sum additions, [29] candidates of pass 1
// per half-round over the workgroup's workers (ticks): [30] longest scoring phase, [31] mean scoring phase, [32] longest
// bookkeeping + refit phase, [33] its mean, [34] longest busy time (both), [35] mean busy time
// [8..15] the server's time (100 MHz ticks): SVD, recycle, load completion, hand-out, active list, load issue, scoring by
// ticket, waiting at the barriers; [16..19] a worker's (wave 0): scoring, bookkeeping + refits, waiting at the barriers, -
#ifdef RGBDFE_SPLIT_STATS
// (a wave counts in registers and adds to the global array once, when it leaves the kernel: per-event atomics on 24 words
// shared by 4096 waves serialised the launch they were meant to describe -- 4.8 instead of 1.0 ms per batch)
__device__ unsigned long long g_split_stats[40];
#define ST_DECL unsigned long long st_c[40] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_mx = 0;
#define ST_ADD(I, V) { st_c[I] += (unsigned long long)(V); }
#define ST_MAX(I, V) { st_mx = (unsigned long long)(V) > st_mx ? (unsigned long long)(V) : st_mx; }
#define ST_T0 unsigned long long st_t = __builtin_amdgcn_s_memrealtime();
#define ST_LAP(I) { const unsigned long long st_n = __builtin_amdgcn_s_memrealtime(); st_c[I] += st_n - st_t; st_t = st_n; }
#define ST_OUT { if ((threadIdx.x & 63) == 0) { for (int st_i = 0; st_i < 40; ++st_i) if (st_c[st_i] != 0ull) atomicAdd(&g_split_stats[st_i], st_c[st_i]); \
                 if (st_mx != 0ull) atomicMax(&g_split_stats[7], st_mx); } }
#else
#define ST_DECL
#define ST_ADD(I, V)
#define ST_MAX(I, V)
#define ST_T0
#define ST_LAP(I)
#define ST_OUT
#endif

#ifndef RGBDFE_SPLIT_WAVES
#define RGBDFE_SPLIT_WAVES 8
#endif
constexpr int kStreamWaves = RGBDFE_SPLIT_WAVES;       // waves of a refinement workgroup: the workers + the server
#ifndef RGBDFE_SPLIT_WGS_PER_CU
#define RGBDFE_SPLIT_WGS_PER_CU (16 / RGBDFE_SPLIT_WAVES)
#endif
constexpr int kWgsPerCu = RGBDFE_SPLIT_WGS_PER_CU;     // (default: 16 waves per CU = 4 per SIMD at <= 128 VGPRs)
static_assert(kStreamWaves >= 3 && (kStreamWaves * kWgsPerCu) % 4 == 0, "whole waves per SIMD");
#define RGBDFE_SPLIT_EU_WAVES (RGBDFE_SPLIT_WAVES * RGBDFE_SPLIT_WGS_PER_CU / 4)
constexpr int kWorkers = kStreamWaves - 1;
#ifndef RGBDFE_SPLIT_SERVER_SCORES
#define RGBDFE_SPLIT_SERVER_SCORES (RGBDFE_SPLIT_WAVES >= 8)
#endif
constexpr bool kServerScores = RGBDFE_SPLIT_SERVER_SCORES;   // the server takes tickets too (and owns a scoring scratch)
constexpr int kStreamThreads = kStreamWaves * kWave;
#ifndef RGBDFE_SPLIT_WAVE_SLOTS
#define RGBDFE_SPLIT_WAVE_SLOTS 7
#endif
#ifndef RGBDFE_SPLIT_BUFS
#define RGBDFE_SPLIT_BUFS (RGBDFE_SPLIT_WAVES >= 8 ? 4 : 2)
#endif
constexpr int kWaveSlots = RGBDFE_SPLIT_WAVE_SLOTS;    // iterations a worker refines side by side, per group (7 x 9 = 63 lanes of the refit)
static_assert(kWaveSlots <= kSlots, "the refit phase (fit_compact / fit_recurrence) holds kSlots lists per wave");
constexpr int kGroupSlots = kWorkers * kWaveSlots;     // 49 slots of a group: one lane each in the server's SVD
constexpr int kStreamSlots = 2 * kGroupSlots;
static_assert(kGroupSlots <= kWave, "the server looks at a group's slots with one lane each");
static_assert(kStreamSlots <= 2 * kWave, "the server's end-of-work test looks at two slots per lane");
#ifndef RGBDFE_SPLIT_TICKETS_FROM
#define RGBDFE_SPLIT_TICKETS_FROM (24 * (RGBDFE_SPLIT_WAVES - 1) * RGBDFE_SPLIT_WAVE_SLOTS / 49)   // (half of a group's slots)
#endif
constexpr int kCostDear = 4;                           // a scoring that runs pass 2, in scorings of a junk hypothesis (the deal's weight)
constexpr int kTicketsFrom = RGBDFE_SPLIT_TICKETS_FROM;  // expensive scorings in a group's pass from which they go out by ticket
#ifndef RGBDFE_SPLIT_PACK_FROM
#define RGBDFE_SPLIT_PACK_FROM 1000
#endif
constexpr int kPackFrom = RGBDFE_SPLIT_PACK_FROM;     // refined matches of a pass's dear slots from which its refits are packed (-1: never)
constexpr int kBufs = RGBDFE_SPLIT_BUFS;              // units (pair, iteration range) resident in a workgroup's LDS
#ifndef RGBDFE_SPLIT_MAX_SHARE
#define RGBDFE_SPLIT_MAX_SHARE 256
#endif
constexpr int kMaxShare = RGBDFE_SPLIT_MAX_SHARE;      // iterations of a unit at most (the host cuts longer ranges)
constexpr int kMaskLanes = kMaxShare / kWave + 1;     // words of a pair's viable mask that can overlap a unit's range
constexpr int kMVec = RGBDFE_MAX_MATCHES * kRec / 4;  // float4s of a pair's match records
constexpr int kPrepVec = (int)(sizeof(PairPrep) / 16);  // a pair's match records + facts as 16-byte pieces
static_assert(sizeof(PairPrep) % 16 == 0, "PairPrep goes global -> LDS in 16-byte pieces");

// one RANSAC iteration in flight (see select_ransac.hip: Slot); the facts of its unit are in the unit's buffer
constexpr int kSlotDone = 0, kSlotActive = 1, kSlotRecorded = 2;
struct SlotS {
  union {
    struct { float R[9], t[3]; } x;  // transform to score next (first the 4-point hypothesis, then the refits')
    float svd_in[15];                // ... or, between a refit's recurrences and its SVD: C[9], mean1[3], mean2[3]
  } u;
  // (refined_transformation and refined_matches, node.cpp:1137 / :1163, are not kept here: whenever a pass's scoring is
  // accepted they go straight into the iteration's outcome record in memory -- the last acceptance is the outcome -- and the
  // refit that follows an acceptance reads the set it has just accepted, cmask.  88 bytes per slot: a fourth resident unit.)
  uint64_t cmask[kRounds];   // inlier set of the scoring of the current pass
  double rerr;               // refined_error
  double csum;               // sequential sum of the current scoring's inlier errors (node.cpp:1006)
  int rn;                    // refined_matches.size()
  int cn;                    // inliers of the current scoring
  int active;                // kSlotActive: inside the refinement loop (:1140-1169); kSlotDone: it has left the loop;
                             // kSlotRecorded: ... and its worker has written the outcome record
  int round;                 // refinement passes done
  int iter;                  // RANSAC iteration held by the slot, -1 = free (written by the server only)
  int buf;                   // LDS buffer of the slot's unit
  uint32_t pair;             // the unit's pair
  // facts of the unit, copied at the hand-out: a scoring finds them with the slot's transform in one LDS round trip instead
  // of chasing buf -> PairPrep / UnitCtx
  int n_all;                 // PairPrep::n_all
  uint32_t thr;              // UnitCtx::thr
  float pmax;                // PairPrep::pmax
};
// scoring phase of a wave: candidates of pass 1, inlier bits, the inliers' errors in match order
struct ScoreB {
  uint16_t cand[RGBDFE_MAX_MATCHES];
  uint32_t mbits[2 * kRounds];
  uint32_t pad[2];
  double err[RGBDFE_MAX_MATCHES + 8];  // + one block of zeros behind the list (the sum runs in blocks of 8)
};
static_assert(offsetof(ScoreB, err) % 16 == 0, "16-byte reads of the error list");
struct alignas(16) WaveLds {
  union {  // the phases of a half-round never overlap
    ScoreB sc;
    FitBuf fit;
  } u;
};
// a unit resident in LDS: the viable iterations of its range, handed out in order.  Server only.
constexpr int kUnitFree = 0, kUnitLoading = 1, kUnitReady = 2, kUnitDrained = 3;
struct alignas(16) UnitCtx {
  uint64_t mw[kMaskLanes + 1];  // the words of the pair's viable mask that overlap the window (global -> LDS with the records)
  int state;                 // kUnitFree / kUnitLoading (the records are on their way) / kUnitReady / kUnitDrained (every
                             // iteration of the window has ended: the next window, or the buffer is free again)
  int next;                  // iterations handed out so far
  int done;                  // iterations whose refinement has ended; == n_items => the window has drained
  int n_items;
  uint32_t pair;
  int kb, ke;                // the window: the unit's iteration range / the part of the pair's range that is recorded now
  uint32_t thr;              // inlier threshold of the pair (:1094-1098)
  int base;                  // iteration index of bit 0 of mw[0]
  int cls;                   // phased plans: 2 = junk-heavy and no jump of `it` (everything that is left is one window), 0 / 1 =
                             // phase by phase, -1 = not decided yet (the first walk does)
  // the pair's in-order bookkeeping up to the windows walked so far (node.cpp:1171-1190), phased plans
  int w_it, w_real, w_valid, w_best_idx, w_best_n, w_state;
  float w_rmse;
  int pad[3];
  uint16_t klist[kMaxShare];  // the viable iterations of the unit's range, ascending, relative to `base` (< kMaxShare + 64)
};
struct alignas(16) StreamLds {
  PairPrep prep[kBufs];            // the resident units' pairs: match records + facts, as pair_prep_kernel left them
  WaveLds w[kServerScores ? kStreamWaves : kWorkers];
  SlotS slot[kStreamSlots];        // group g: slots g * kGroupSlots + 0 .. kGroupSlots - 1
  UnitCtx ctx[kBufs];
  // facts of the units of the block the server has taken off the counter last (lane = unit), global -> LDS like the records
  // (a load into registers that stays in flight across the server's loop makes the compiler wait before every reuse of
  // any register such a load might be pending on -- between the pieces of a unit's record load, for one)
  int claim_nall[kWave];           // PairPrep::n_all
  int claim_cls[kWave];            // the hypothesis kernel's pre-classification byte (phased plans)
  // the scorings of a half-round: the group's active slots as a list, taken one by one by whichever wave is free
  uint8_t act_list[2][kWave];
  int n_act[2];
  int task[2];                     // next entry of act_list[g] to score (ds_add_rtn: a ticket, nobody waits for anybody)
  int tickets[2];                  // the group's next pass hands its scorings out by ticket (many expensive ones) / by list
  // the work of a pass: which slots of the group each worker scores (unless the pass goes by ticket), keeps the books of and
  // refits -- at most kWaveSlots each, dealt by the server so that the workers' loads are level (a slot belongs to nobody)
  uint8_t wlist[2][kWorkers][8];   // [..][7] = entries
  // set by the server in half-round h: every unit of the launch has been refined.  One word per half-round parity: the
  // workers read quit[h & 1] behind the barrier that ends half-round h, and the server -- which may be well into h + 1 by
  // then -- writes only quit[(h + 1) & 1] there; quit[h & 1] is written again in h + 2, behind a barrier every worker has
  // passed after its read
  int quit[2];
  int sum_rn[2];                   // refined matches held by the group's active slots (hand_out -> deal_work: the packed deal)
  uint16_t ord_cnt[kWave], ord_start[kWave];   // phased plans: pairs per order bucket (lane b: bucket 63 - b) and their exclusive prefix
                                               // sums (a batch has at most 65 535 pairs) -- kept here, not in registers of the server:
                                               // it has none to spare
#ifdef RGBDFE_SPLIT_STATS
  unsigned int st_w[2][8][2];     // [half-round parity][worker][scoring, bookkeeping + refit] ticks of the half-round
#endif
};
static_assert(sizeof(StreamLds) <= 160 * 1024 / kWgsPerCu, "kWgsPerCu workgroups per CU");

// ---------------------------------------------------------------------------------
// computeInliersAndError (node.cpp:968-1020) with errorFunction2 (misc.cpp:697-770) for one wave-uniform transform:
// inlier set, count, and -- when the count can be accepted at all -- the sequential sum of the inliers' errors.
//   pass 1 (lane = match): float prefilter of the shortcut test (misc.cpp:726-735) with a proven error band, the round in
//           double when a lane is too close to call; ballot compaction of the candidates;
//   pass 2 (lane = candidate): double-precision covariance + 3x3 Cholesky solve; the inliers' errors go to LDS in match
//           order;
//   sum    (all lanes alike, broadcast reads): mean_error += mahal_dist in match order.
// Fewer candidates than `need` => the caller rejects the scoring whatever the numbers are: (0, -) is returned.
// Same arithmetic as score_passes + sum_rows of select_ransac.hip.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void score_b(const float* R, const float* tr, const float* __restrict__ M, int n_all,
                                        uint32_t need, const RansacConst& rc, ScoreB& sb, float pmax, uint64_t* mask,
                                        int& n_inl, double& sum, int& n_cand_out) {
  const int lane = threadIdx.x & (kWave - 1);
  double Rd[9], td[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rd[i] = (double)R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) td[i] = (double)tr[i];
  const double rcx = rc.raster_cov_x, rcy = rc.raster_cov_y, dc = rc.depth_cov;
  const double smax = rcx > dc ? rcx : dc;
  const double shortcut = 2.0 * (smax + smax);
  float lo_f, hi_f;
  {
    const float u4 = 4.0f * 5.9604645e-8f;
    float es = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      es += u4 * (((fabsf(R[3 * i]) + fabsf(R[3 * i + 1])) + fabsf(R[3 * i + 2]) + 1.0f) * pmax + fabsf(tr[i]));
    const float S = (float)shortcut;
    const float E = 2.0f * ((2.0f * sqrtf(S) * 1.001f) * es + es * es + u4 * S) + 1e-30f;
    lo_f = S * 0.999999f - E;
    hi_f = S * 1.000001f + E;
  }
  // (three steps over all five rounds -- every record first, then the float arithmetic, then the compaction -- instead of
  // five rounds one after the other: a round that waits for its own LDS reads and ends in branches cost a wave five exposed
  // LDS round trips per scoring, and these waves are bound by their own latencies, not by issue slots)
  int n_cand = 0;
  float rec[kRounds][6];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int m = r * kWave + lane;
#pragma unroll
    for (int c = 0; c < 6; ++c) rec[r][c] = M[m * kRec + c];
  }
  bool pre[kRounds], cand[kRounds];
  bool unsure = false;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int m = r * kWave + lane;
    const float pxf = rec[r][0], pyf = rec[r][1], pzf = rec[r][2];
    const float qxf = rec[r][3], qyf = rec[r][4], qzf = rec[r][5];
    // node.cpp:994 (z == 0 skip) ; misc.cpp:712-717 (NaN -> DBL_MAX)
    pre[r] = (m < n_all) && !(pzf == 0.0f || qzf == 0.0f) && !(__builtin_isnan(pzf) || __builtin_isnan(qzf));
    const float f0 = __builtin_fmaf(R[0], pxf, __builtin_fmaf(R[1], pyf, __builtin_fmaf(R[2], pzf, tr[0]))) - qxf;
    const float f1 = __builtin_fmaf(R[3], pxf, __builtin_fmaf(R[4], pyf, __builtin_fmaf(R[5], pzf, tr[1]))) - qyf;
    const float f2 = __builtin_fmaf(R[6], pxf, __builtin_fmaf(R[7], pyf, __builtin_fmaf(R[8], pzf, tr[2]))) - qzf;
    const float dsq_f = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, f2 * f2));
    const bool sure_in = dsq_f < lo_f, sure_out = dsq_f > hi_f;
    cand[r] = pre[r] && sure_in;
    unsure = unsure || (pre[r] && !(sure_in || sure_out));
  }
  if (__ballot(unsure) != 0ull) {  // a lane too close to call in some round: those rounds in double
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const float pxf = rec[r][0], pyf = rec[r][1], pzf = rec[r][2];
      const float qxf = rec[r][3], qyf = rec[r][4], qzf = rec[r][5];
      const double a0 = (double)pxf, a1 = (double)pyf, a2 = (double)pzf;
      const double d0 = (((Rd[0] * a0 + Rd[1] * a1) + Rd[2] * a2) + td[0]) - (double)qxf;
      const double d1 = (((Rd[3] * a0 + Rd[4] * a1) + Rd[5] * a2) + td[1]) - (double)qyf;
      const double d2 = (((Rd[6] * a0 + Rd[7] * a1) + Rd[8] * a2) + td[2]) - (double)qzf;
      const double dsq = (d0 * d0 + d1 * d1) + d2 * d2;
      cand[r] = pre[r] && !(dsq > shortcut) && !__builtin_isnan(d2);  // misc.cpp:731, 755
    }
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint64_t cm = __ballot(cand[r]);
    if (cand[r]) sb.cand[n_cand + (int)lane_rank(cm)] = (uint16_t)(r * kWave + lane);
    n_cand += __popcll(cm);
  }
  if (lane < 2 * kRounds) sb.mbits[lane] = 0u;
  lsync();
  n_inl = 0;
  sum = 0.0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) mask[r] = 0ull;
  n_cand_out = n_cand;
  if ((uint32_t)n_cand < need) return;  // hopeless: nobody looks at the exact numbers
  // ---- pass 2
  for (int k0 = 0; k0 < n_cand; k0 += kWave) {
    const int k = k0 + lane;
    const bool act = k < n_cand;
    const int m = act ? (int)sb.cand[k] : 0;
    const double a0 = (double)M[m * kRec + 0], a1 = (double)M[m * kRec + 1], a2 = (double)M[m * kRec + 2];
    const double b0 = (double)M[m * kRec + 3], b1 = (double)M[m * kRec + 4], b2 = (double)M[m * kRec + 5];
    double d[3];
    // mu_1_in_frame_2 = (T * x1).head<3>() with x1.w == 1 (misc.cpp:724)
    d[0] = (((Rd[0] * a0 + Rd[1] * a1) + Rd[2] * a2) + td[0]) - b0;
    d[1] = (((Rd[3] * a0 + Rd[4] * a1) + Rd[5] * a2) + td[1]) - b1;
    d[2] = (((Rd[6] * a0 + Rd[7] * a1) + Rd[8] * a2) + td[2]) - b2;
    double e = DBL_MAX;
    {
      const double c1[3] = {rcx * a2, rcy * a2, dc};
      const double c2[3] = {rcx * b2, rcy * b2, dc};
      // S = R^T * cov1 * R + cov2 (misc.cpp:751,760), lower triangle only
      double A[9];  // A[i*3+k] = R(k,i) * c1_k
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) A[i * 3 + kk] = Rd[kk * 3 + i] * c1[kk];
      const double S00 = ((A[0] * Rd[0] + A[1] * Rd[3]) + A[2] * Rd[6]) + c2[0];
      const double S10 = (A[3] * Rd[0] + A[4] * Rd[3]) + A[5] * Rd[6];
      const double S11 = ((A[3] * Rd[1] + A[4] * Rd[4]) + A[5] * Rd[7]) + c2[1];
      const double S20 = (A[6] * Rd[0] + A[7] * Rd[3]) + A[8] * Rd[6];
      const double S21 = (A[6] * Rd[1] + A[7] * Rd[4]) + A[8] * Rd[7];
      const double S22 = ((A[6] * Rd[2] + A[7] * Rd[5]) + A[8] * Rd[8]) + c2[2];
      // LLT + solve (misc.cpp:763), D5: non-positive pivot -> DBL_MAX
      bool ok, unsafe;
      double ee = mahal_llt_fast(S00, S10, S11, S20, S21, S22, d, ok, unsafe);
      if (__ballot(act && unsafe) != 0ull)  // an operand outside the fast path's exponent window (or 0, NaN)
        ee = mahal_llt_ieee(S00, S10, S11, S20, S21, S22, d, ok);
      if (ok && (ee >= 0.0)) e = ee;  // misc.cpp:765-768
    }
    const bool inl = act && !(e > rc.sq_max_dist) && (e >= 0.0);  // node.cpp:998,1001
    const uint64_t im = __ballot(inl);
    if (inl) {
      sb.err[n_inl + (int)lane_rank(im)] = e;  // candidates ascend in match index: so do the inliers
      atomicOr(&sb.mbits[m >> 5], 1u << (m & 31));
    }
    n_inl += __popcll(im);
  }
  if (lane < 8) sb.err[n_inl + lane] = 0.0;  // x + 0.0 == x for these non-negative sums
  lsync();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(sb.mbits[2 * r]);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(sb.mbits[2 * r + 1]);
    mask[r] = ((uint64_t)hi << 32) | lo;
  }
  // mean_error += mahal_dist in match order (node.cpp:1006): one dependent chain, every lane alike; only when the
  // scoring can be accepted (else mean_error = 1e9, node.cpp:1012-1014, or the caller rejects it by its count)
  if (!((uint32_t)n_inl < need || n_inl < 3)) {
    double s = 0.0;
    for (int k = 0; k < n_inl; k += 8) {
      const double2 v0 = *reinterpret_cast<const double2*>(&sb.err[k]);
      const double2 v1 = *reinterpret_cast<const double2*>(&sb.err[k + 2]);
      const double2 v2 = *reinterpret_cast<const double2*>(&sb.err[k + 4]);
      const double2 v3 = *reinterpret_cast<const double2*>(&sb.err[k + 6]);
      s += v0.x; s += v0.y; s += v1.x; s += v1.y;
      s += v2.x; s += v2.y; s += v3.x; s += v3.y;
    }
    sum = s;
  }
  lsync();  // the next scoring writes the candidate list again
}

}  // namespace

// ---------------------------------------------------------------------------------
// LANE = ITERATION: sample + 4-point fit + pre-screen of every iteration of a pair.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kHypThreads) void ransac_hyp_kernel(const PairWork* __restrict__ work, uint32_t n_pairs,
                                                                const RansacConst rc, const SplitPlan plan) {
  __shared__ __attribute__((aligned(16))) float M[RGBDFE_MAX_MATCHES * kRec];
  __shared__ __attribute__((aligned(16))) float S4[(RGBDFE_MAX_MATCHES / 4) * kS4Block];   // the pre-screen's copy (ransac_device.h)
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  // the batch's counters (walk[n_pairs]: class-1 pairs, the refinement launches' unit counters, "still running" flag)
  // start at zero: done here, ahead of every kernel that uses them (a memset node captured into the batch's hipGraph
  // was not reliably in effect when the graph was replayed)
  if (pair == 0 && threadIdx.x < sizeof(WalkState) / 4) reinterpret_cast<uint32_t*>(plan.walk + n_pairs)[threadIdx.x] = 0u;
  const PairPrep* __restrict__ pp = plan.prep + pair;
  const int n_all = pp->n_all;
  // the pair's in-order bookkeeping starts here (:1112, :1130); WalkState::speculate = end of the pair's recorded range:
  // everything when the batch is recorded with full speculation, else whatever the refinement kernel leaves behind
  if (threadIdx.x == 0) {
    WalkState ws;
    ws.state = rc.ransac_iterations;
    ws.it = 0; ws.real_iterations = 0; ws.valid_iterations = 0;
    ws.best_idx = -1; ws.best_n = 0;
    ws.rmse = 1e6f;
    ws.speculate = plan.phased ? 0 : rc.ransac_iterations;
    plan.walk[pair] = ws;
  }
  // no RANSAC for this pair (node.cpp:1087, :1130)
  if (!(n_all > rc.min_matches && n_all >= 4)) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pp->M);
    float4* __restrict__ dst = reinterpret_cast<float4*>(M);
    for (int v = tid; v < kMVec; v += kHypThreads) dst[v] = src[v];
  }
  __shared__ int n_viable_sh;   // viable iterations of the pair, summed over the waves at the end
  if (tid == 0) n_viable_sh = 0;
  __syncthreads();
  transpose_records(M, S4, tid, kHypThreads);
  __syncthreads();
  const float pmax = pp->pmax;
  uint32_t thr = (uint32_t)rc.min_matches;                                         // :1094
  if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
  const uint32_t seed_uid = mix32(mix32(rc.seed ^ 0x9E3779B9u) + work[pair].uid * 0x85EBCA6Bu);
  const int I = rc.ransac_iterations;
  IterRec* __restrict__ rec_pair = plan.recs + (size_t)pair * (size_t)I;
  uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
  int n_viable_wave = 0;
  for (int k0 = 0; k0 < I; k0 += kHypThreads) {
    const int k = k0 + tid;
    const bool in_range = k < I;
    const uint32_t iter = (uint32_t)k;
    uint32_t ids[4] = {0, 0, 0, 0};
    int cnt = 0;
    {
      // sample_matches_prefer_by_distance (node.cpp:1024-1047): ascending std::set of 4 ids
      int safety_net = 0;
      uint32_t kk = 0;
      const uint32_t n = (uint32_t)n_all;
      while (cnt < 4) {
        uint32_t id1 = rand31(seed_uid, iter, kk) % n;
        uint32_t id2 = rand31(seed_uid, iter, kk + 1) % n;
        kk += 2;
        if (id1 > id2) id1 = id2;
        const bool dup = (cnt > 0 && ids[0] == id1) || (cnt > 1 && ids[1] == id1) || (cnt > 2 && ids[2] == id1);
        if (!dup) {
          uint32_t v = id1;  // sorted insert
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 < cnt) {
              if (ids[s4] > v) { uint32_t tmp = ids[s4]; ids[s4] = v; v = tmp; }
            } else if (s4 == cnt) {
              ids[s4] = v;
            }
          }
          ++cnt;
        }
        if (++safety_net > 10000) break;
      }
    }
    Tfc acc;
    acc.reset();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      if (s4 < cnt) acc.add(M, (int)ids[s4]);
    float hypR[9], hypt[3];
    tfc_get_transformation(acc, hypR, hypt);
    // a NaN transform leaves the refinement loop at once (:1144); so does, after its first scoring, a hypothesis that
    // cannot reach `thr` candidates (:1154) -- with the same outcome: refined_matches stays empty (:1133-1134)
    const uint32_t may_pass = prescreen_may_pass_s4(hypR, hypt, S4, n_all, pmax, rc);
    const bool viable = in_range && !has_nan12(hypR, hypt) && may_pass >= thr;
    const uint64_t vm = __ballot(viable);
    if (lane == 0) vm_pair[k >> 6] = vm;  // (k0 and the wave's first lane are multiples of 64)
    n_viable_wave += __popcll(vm);
    if (k0 == 0 && tid < kWave && plan.preclass_iters > 0) {  // the first wave: the pair's class by the pre-screen alone
      // junk-heavy for certain (at most 9 of the first 14 iterations can give a refined hypothesis) AND no sign of a
      // hypothesis with more than half of the matches as inliers among them (the loop's early exits, :1186-1188): the
      // largest candidate bound stays below 30 % of the matches.  Only a scheduling hint: the walk decides the outcome.
      const int n14 = min(plan.preclass_iters, min(I, kWave));
      uint32_t best = (lane < n14 && viable) ? may_pass : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, d));
      const int viable14 = __popcll(vm & (n14 >= 64 ? ~0ull : ((1ull << n14) - 1ull)));
      if (lane == 0)
        plan.preclass[pair] = (viable14 * kClass2Den <= n14 * kClass2Num && best * 10u <= (uint32_t)n_all * 3u) ? 2 : 0;
    }
    if (viable) {
      IterRec& r = rec_pair[k];
#pragma unroll
      for (int i = 0; i < 9; ++i) r.rR[i] = hypR[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) r.rt[i] = hypt[i];
#ifdef RGBDFE_PROFILE_PHASES
      r.rn = -77;  // diagnostics: "a hypothesis, not yet refined"
#endif
    }
    // (an iteration that is not viable leaves nothing behind: the walk reads its cleared bit of the mask as {1e6, 0})
  }
  // phased plans: the pair joins the order bucket of its number of viable iterations (SplitPlan::order)
  if (plan.phased) {
    if (lane == 0) atomicAdd(&n_viable_sh, n_viable_wave);
    __syncthreads();
    if (tid == 0) {
      const int b = order_bucket(n_viable_sh);
      const uint32_t at = atomicAdd(plan.order_cnt + b, 1u);
      if (at < n_pairs) plan.order[(size_t)b * (size_t)n_pairs + at] = pair;   // (a bucket holds every pair at most once: the
                                                                               // counters were zeroed by pair_prep_kernel)
    }
  }
}

// ---------------------------------------------------------------------------------
// The refinement loops (node.cpp:1140-1169) of the viable iterations of [phase_begin, phase_end / spec_end): persistent
// workgroups of 7 workers + 1 server, two slot groups taking turns, one s_barrier per half-round (see the file header).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(RGBDFE_SPLIT_EU_WAVES, RGBDFE_SPLIT_EU_WAVES))) void ransac_refine_kernel(
    uint32_t n_pairs, const RansacConst rc, const SplitPlan plan, uint32_t n_units) {
  extern __shared__ __attribute__((aligned(16))) char stream_smem[];
  StreamLds& lds = *reinterpret_cast<StreamLds*>(stream_smem);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int I = rc.ransac_iterations;
  ST_DECL

  if (wave < 2) {
    const int s = wave * kWave + lane;
    if (s < kStreamSlots) { lds.slot[s].active = kSlotDone; lds.slot[s].iter = -1; lds.slot[s].buf = 0; }
    if (wave == 0 && lane < kBufs) { lds.ctx[lane].state = kUnitFree; lds.ctx[lane].next = 0; lds.ctx[lane].done = 0; lds.ctx[lane].n_items = 0; }
    if (wave == 0 && lane < 2) lds.quit[lane] = 0;
  }
  lds_barrier();

  // an iteration has left its refinement loop (its worker's lane, or the server's): what the in-order walk needs of it.  The
  // outcome record itself was written when the pass that set refined_matches was accepted (accept_record); an iteration whose
  // refined_matches stayed empty has none -- the walk never adopts it (:1171).
  auto write_summary = [&](const SlotS& sl) {
    plan.sums[(size_t)sl.pair * (size_t)I + (size_t)sl.iter] = IterSum{sl.rerr, sl.rn, 0};
  };
  // :1160-1165: the pass's transform, inlier set and error become the iteration's outcome so far
  auto accept_record = [&](const SlotS& sl, int n_inl, double err) {
    IterRec& r = plan.recs[(size_t)sl.pair * (size_t)I + (size_t)sl.iter];
#pragma unroll
    for (int i = 0; i < 9; ++i) r.rR[i] = sl.u.x.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) r.rt[i] = sl.u.x.t[i];
#pragma unroll
    for (int q = 0; q < kRounds; ++q) r.rmask[q] = sl.cmask[q];
    r.rerr = err;
    r.rn = n_inl;
    r.pad = 0;
  };

  // ---- the scorings (:1148) of group g's pass: every wave takes the next active slot off the group's list until there is
  // none left (a ticket per scoring: scorings differ a lot in their cost -- a junk hypothesis ends after the float
  // prefilter, a good one runs ~140 Cholesky solves -- so fixed shares would leave most waves waiting at the barrier)
  // ---- one scoring (:1148) of slot sl by this wave (lane = match): the inlier set, its size and error sum into the slot
  auto score_slot = [&](SlotS& sl) {
    WaveLds& wl = lds.w[wave];
    float curR[9], curt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) curR[i] = sl.u.x.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) curt[i] = sl.u.x.t[i];
    const int b = __builtin_amdgcn_readfirstlane(sl.buf);
    const PairPrep& pp = lds.prep[b];
    const int n_all = __builtin_amdgcn_readfirstlane(sl.n_all);
    const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.thr);
    const float pmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sl.pmax)));
    // a scoring with fewer inliers than max(threshold, refined_matches.size()) is rejected whatever its error
    // is (:1154, :1160): the scorer may stop counting as soon as that is certain
    const uint32_t need = max(thr, (uint32_t)__builtin_amdgcn_readfirstlane(sl.rn));
    uint64_t inl_mask[kRounds];
    int n_inl;
    double sum;
    int n_cand;
    score_b(curR, curt, pp.M, n_all, need, rc, wl.u.sc, pmax, inl_mask, n_inl, sum, n_cand);
#ifdef RGBDFE_SPLIT_STATS
    ST_ADD(29, n_cand)
    if ((uint32_t)n_cand >= need) {
      ST_ADD(26, 1) ST_ADD(27, (n_cand + 63) / 64)
      if (!((uint32_t)n_inl < need || n_inl < 3)) ST_ADD(28, n_inl)
    }
#endif
    // (the lane tests inside loops are opaque to the compiler, each on its own: seeing the same `lane == 0` twice in a
    // loop body it threads the first branch into the second and splits the loop by lane -- in the ticket loop below lane 0
    // left with its ticket and the other 63 lanes stayed behind in a copy whose readfirstlane never saw a new ticket: an
    // endless loop, found on the GPU)
    if (fresh(threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
      for (int r = 0; r < kRounds; ++r) sl.cmask[r] = inl_mask[r];
      sl.cn = n_inl;
      sl.csum = sum;
    }
  };
  // ---- the scorings of group g's pass when they are expensive (good hypotheses: ~140 Cholesky solves each, against a
  // junk hypothesis that ends after the float prefilter): every wave takes the next active slot off the group's list
  // until there is none left -- a ticket per scoring, nobody waits for anybody; fixed shares would leave most waves
  // waiting at the barrier for the one with the most good hypotheses
  auto score_tickets = [&](int g) {
    const int n_act = __builtin_amdgcn_readfirstlane(lds.n_act[g]);
    for (;;) {
      int t = 0;
      if (fresh(threadIdx.x & (kWave - 1)) == 0) t = atomicAdd(&lds.task[g], 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= n_act) break;
      score_slot(lds.slot[g * kGroupSlots + (int)lds.act_list[g][t]]);
    }
  };

  if (wave < kWorkers) {
    // =========================================================================================== a worker
    WaveLds& wl = lds.w[wave];
    lds_barrier();   // (the server's prologue: the first unit's iterations are in group 0's slots)
    ST_T0
    for (int h = 0;; ++h) {
      const int g = h & 1;
      const int lane = fresh(threadIdx.x & (kWave - 1));
#ifdef RGBDFE_SPLIT_STATS
      const unsigned long long st_s0 = st_c[16], st_f0 = st_c[17];
#endif
      // this pass's slots of this wave (the server's deal): lane j < n_mine_slots holds the j-th
      const int n_mine_slots = __builtin_amdgcn_readfirstlane((int)lds.wlist[g][wave][7]);
      const int my_slot = g * kGroupSlots + (lane < n_mine_slots ? (int)lds.wlist[g][wave][min(lane, kWaveSlots - 1)] : 0);
      auto slot_at = [&](int j) -> SlotS& { return lds.slot[__builtin_amdgcn_readlane(my_slot, j)]; };
      // ================================ one pass of the refinement loop (:1140) for every active slot of group g
      if (__builtin_amdgcn_readfirstlane(lds.tickets[g]) != 0) {
        score_tickets(g);
        ST_LAP(16)
        lds_barrier();   // every scoring of the pass is done
        ST_LAP(18)
      } else {           // cheap scorings: every worker scores its own slots and goes on without meeting the others
        for (int j = 0; j < n_mine_slots; ++j) score_slot(slot_at(j));
        lsync();
        ST_LAP(16)
      }
      // ---- the loop's bookkeeping (:1154-1166), lane = slot
      bool still = false;
      if (fresh(lane) < n_mine_slots) {
        SlotS& sl = lds.slot[my_slot];
        if (sl.active == kSlotActive) {
          const int n_inl = sl.cn, rn = sl.rn;
          const uint32_t thr = sl.thr;
          const uint32_t need = max(thr, (uint32_t)rn);
          // mean_error = 1e9 below 3 inliers (:1012-1014); a count below `need` is rejected by the count
          const double err_mine = !((uint32_t)n_inl < need || n_inl < 3) ? sqrt(sl.csum / (double)n_inl) : 1e9;  // :1016-1017
          float max_dist_f = rc.max_dist_m;
          asm volatile("" : "+v"(max_dist_f));  // (kept out of the loop-invariant registers: they are scarce)
          if (!((uint32_t)n_inl < thr || err_mine > (double)max_dist_f)) {  // :1154
            if (n_inl >= rn && err_mine <= sl.rerr) {               // :1160
              still = (n_inl != rn);                                // :1166
              accept_record(sl, n_inl, err_mine);
              sl.rn = n_inl;
              sl.rerr = err_mine;
            }
          }
          if (sl.round == 18) still = false;  // the 19th pass was the last one (:1140)
          sl.round++;
          if (!still) {  // the iteration has left its loop: the outcome record, from the lane that holds the slot
            write_summary(sl);
            sl.active = kSlotRecorded;
          }
        }
      }
      const uint64_t refit = __ballot(still);
#ifdef RGBDFE_SPLIT_STATS
      {  // [20] iterations that left their loop in this pass, [21] ... after their first scoring, [22] ... without a refined set
        bool ended = false, first = false, empty = false;
        if (fresh(lane) < n_mine_slots) {
          const SlotS& sl = lds.slot[my_slot];
          ended = sl.active == kSlotRecorded && !still;
          first = ended && sl.round == 1;
          empty = ended && sl.rn == 0;
        }
        const int st_e = __popcll(__ballot(ended)), st_f = __popcll(__ballot(first)), st_z = __popcll(__ballot(empty));
        ST_ADD(20, st_e) ST_ADD(21, st_f) ST_ADD(22, st_z)
      }
#endif
      if (refit != 0ull) {
        lsync();
        // ---- refits (:1142): the weighted-mean recurrences of the wave's active slots side by side; their SVDs are
        // the server's, next half-round
        // (the unscaled division of the recurrence needs every slot's pair inside its window)
        bool slow_div = false;
        if (still) slow_div = lds.prep[lds.slot[my_slot].buf].fast_alpha == 0u;
        const bool all_fast = __ballot(slow_div) == 0ull;
        int n_mine = 0, k256_mine = 0, n_max = 0, n_min = RGBDFE_MAX_MATCHES;
        {
          uint64_t todo = refit;
          while (todo != 0ull) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const SlotS& sl = slot_at(j);
            const PairPrep& pp = lds.prep[__builtin_amdgcn_readfirstlane(sl.buf)];
            // refined_matches without zero weights: tfc.add skips weight == 0; NaN depths never reach an inlier set
            // (misc.cpp:712-717)
            uint64_t m5[kRounds], nz[kRounds];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) { m5[r] = uniform_u64(sl.cmask[r]); nz[r] = uniform_u64(pp.w_nonzero[r]); }
            int k256_j;
            const int n_j = fit_compact(j, m5, nz, wl.u.fit, k256_j, fresh(lane));
            if (lane / 9 == j) { n_mine = n_j; k256_mine = k256_j; }
            n_max = max(n_max, n_j);
            n_min = min(n_min, n_j);
          }
        }
        lsync();
        {
          const int lane_here = fresh(lane);
          const int s = min(lane_here / 9, kWaveSlots - 1), x = lane_here % 9;
          // (per lane: the records of the lane's slot's unit; lanes of positions without a refit read some valid slot)
          const int slot_s = __shfl(my_slot, min(s, max(n_mine_slots - 1, 0)));
          const float* __restrict__ M = lds.prep[lds.slot[slot_s].buf].M;
          float C, m1, m2;
          ST_ADD(23, 1) ST_ADD(24, n_max) ST_ADD(25, __popcll(refit))
          if (all_fast) fit_recurrence<true>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2, lane_here);
          else fit_recurrence<false>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2, lane_here);
          // lane 9s+x holds C[x], lane 9s+j mean1[j], lane 9s+3i mean2[i] of slot s: into the slot's mailbox (the
          // transform it was scored with is history: the same bytes)
          if (lane_here < 9 * kWaveSlots && ((refit >> s) & 1ull)) {
            float* __restrict__ in = lds.slot[slot_s].u.svd_in;
            in[x] = C;
            if (x < 3) in[9 + x] = m1;
            if (x % 3 == 0) in[12 + x / 3] = m2;
          }
        }
      }
      ST_LAP(17)
#ifdef RGBDFE_SPLIT_STATS
      if ((threadIdx.x & 63) == 0) { lds.st_w[g][wave][0] = (unsigned int)(st_c[16] - st_s0); lds.st_w[g][wave][1] = (unsigned int)(st_c[17] - st_f0); }
#endif
      // (the outcome records written in this pass are in memory before the server learns that their iterations have ended:
      // its walk between two windows of a pair reads them back)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      ST_LAP(18)
      if (__builtin_amdgcn_readfirstlane(lds.quit[g]) != 0) break;
    }
#ifdef RGBDFE_SPLIT_STATS
    if (wave != 0) { st_c[16] = st_c[17] = st_c[18] = 0ull; }
    ST_OUT
#endif
    return;
  }

  // ============================================================================================= the server
  // (s_setprio 3 for this wave -- its SVD is ~1500 dependent instructions on a SIMD it shares with three other waves -- was
  // measured: 1.068 -> 1.048 ms serial stage, 1.117 -> 1.133 ms per pipelined step, i.e. nothing.)
  // Units come off the launch's counter a block at a time, lane = unit.  `nx_*`: the block taken last, its facts still on
  // their way (the loads are not awaited until the block is needed); `live` / `u_*`: the block in use.
  uint64_t live = 0ull;        // units of the block in use that have work and are not loaded yet
  bool units_left = true;      // the counter has not run past the last unit yet
  uint32_t u_pair = 0;
  int u_kb = 0, u_ke = 0, u_cls = 0;
  int nx_n = 0;                // units of the prefetched block (0 = none)
  uint32_t nx_pair = 0;
  int nx_share = 0;
  const int blk = 1;           // (every unit of the launch has work: the workgroups take them one by one -- pairs differ a lot)
  const bool phased = plan.phased != 0;
  const bool use_preclass = phased && plan.preclass_iters > 0;
  // phased plans: the launch's units are the pairs of the order buckets, fullest bucket first (lane b: bucket 63 - b)
  uint32_t n_units_here = n_units;
  if (phased) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    const int cnt = (int)min(plan.order_cnt[kOrderBuckets - 1 - lane], n_pairs);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    lds.ord_cnt[lane] = (uint16_t)cnt;
    lds.ord_start[lane] = (uint16_t)(incl - cnt);
    n_units_here = (uint32_t)__builtin_amdgcn_readlane(incl, kWave - 1);
    lsync();
  }

  // ---- the next block of units off the counter; the facts that decide which of them have work are requested (global ->
  // LDS), not awaited
  auto prefetch_block = [&]() {
    nx_n = 0;
    if (!units_left) return;
    const int lane = fresh(threadIdx.x & (kWave - 1));
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(plan.unit_counter, (uint32_t)blk);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= n_units_here) { units_left = false; return; }
    nx_n = (int)min((uint32_t)blk, n_units_here - base);
    const uint32_t unit = base + (uint32_t)min(lane, nx_n - 1);
    if (phased) {   // unit `base` of the launch = entry (base - start) of the bucket whose range holds it
      const int ord_start = (int)lds.ord_start[lane], ord_cnt = (int)lds.ord_cnt[lane];
      const uint64_t holds = __ballot((int)base >= ord_start && (int)base < ord_start + ord_cnt);
      const int bl = (int)__builtin_ctzll(holds);
      const uint32_t idx = base - (uint32_t)__builtin_amdgcn_readlane(ord_start, bl);
      const uint32_t pr = plan.order[(size_t)(kOrderBuckets - 1 - bl) * (size_t)n_pairs + idx];
      nx_pair = min((uint32_t)__builtin_amdgcn_readfirstlane((int)pr), n_pairs - 1u);   // (an index of this batch, whatever the memory held)
      nx_share = 0;
    } else {
      nx_pair = unit / (uint32_t)plan.n_shares;
      nx_share = (int)(unit - nx_pair * (uint32_t)plan.n_shares);
    }
    if (lane < nx_n) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)&plan.prep[nx_pair].n_all,
                                       (__attribute__((address_space(3))) void*)lds.claim_nall, 4, 0, 0);
      if (use_preclass)  // (one byte per pair, read as the low byte of a dword: the array has slack behind it)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(plan.preclass + nx_pair),
                                         (__attribute__((address_space(3))) void*)lds.claim_cls, 4, 0, 0);
    }
  };
  // ---- the prefetched block becomes the block in use: which of its units have anything to do in this launch
  auto adopt_block = [&]() {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (requested at least one pass of the server's loop ago)
    const int n_all = lds.claim_nall[lane];
    // phased plans: a pair the hypothesis kernel has found junk-heavy (class 2) is recorded in one window, the others start
    // with the first phase; full speculation: the unit's share of the range
    const int cls = use_preclass ? (((lds.claim_cls[lane] & 0xFF) == 2) ? 2 : -1) : -1;
    const int k_begin = phased ? 0 : nx_share * plan.share_iters;
    const int k_end = phased ? min(cls == 2 ? I : plan.phase_ends[0], kMaxShare) : min(k_begin + plan.share_iters, I);
    // no RANSAC for this pair (node.cpp:1087, :1130), or an empty share
    const bool have = lane < nx_n && k_begin < k_end && n_all > rc.min_matches && n_all >= 4;
    live = __ballot(have);
    u_pair = nx_pair; u_kb = k_begin; u_ke = k_end; u_cls = cls;
    lsync();   // (the staging arrays have been read before the next block's facts land in them)
    prefetch_block();
  };

  // ---- units with work -> the free buffers: the pair's match records and facts (PairPrep, 8.4 KB) and the words of its
  // viable mask go global -> LDS directly (global_load_lds: no registers, nothing awaited here)
  auto issue_loads = [&]() {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    uint64_t free_bufs = __ballot(lane < kBufs && lds.ctx[min(lane, kBufs - 1)].state == kUnitFree);
    while (free_bufs != 0ull) {
      while (live == 0ull && nx_n != 0) adopt_block();
      if (live == 0ull) return;
      const int src = (int)__builtin_ctzll(live);
      live &= live - 1ull;
      const int b = (int)__builtin_ctzll(free_bufs);
      free_bufs &= free_bufs - 1ull;
      const uint32_t pair = (uint32_t)__builtin_amdgcn_readlane((int)u_pair, src);
      const int kb = __builtin_amdgcn_readlane(u_kb, src), ke = __builtin_amdgcn_readlane(u_ke, src);
      const int cls = __builtin_amdgcn_readlane(u_cls, src);
      UnitCtx& cx = lds.ctx[b];
      const char* __restrict__ srcp = reinterpret_cast<const char*>(plan.prep + pair);
      char* const dst = reinterpret_cast<char*>(&lds.prep[b]);
      for (int i = 0; i < (kPrepVec + kWave - 1) / kWave; ++i) {
        const int v = i * kWave + lane;
        if (v < kPrepVec)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcp + (size_t)v * 16),
                                           (__attribute__((address_space(3))) void*)(dst + i * (kWave * 16)), 16, 0, 0);
      }
      const int blk0 = kb >> 6;
      const int n_dw = 2 * min(kMaskLanes, plan.vmask_words - blk0);
      const char* __restrict__ srcm = reinterpret_cast<const char*>(plan.vmask + (size_t)pair * (size_t)plan.vmask_words + blk0);
      if (lane < n_dw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcm + lane * 4),
                                         (__attribute__((address_space(3))) void*)cx.mw, 4, 0, 0);
      if (lane == 0) {
        cx.pair = pair;
        cx.kb = kb;
        cx.ke = ke;
        cx.cls = cls;
        cx.state = kUnitLoading;
      }
      ST_ADD(5, 1)
    }
  };

  // ---- the list of the viable iterations of unit b's window [kb, ke) from the mask words in cx.mw (lane = iteration of a
  // word); the window is ready to be handed out, or has drained already when nothing of it passed the pre-screen
  auto build_list = [&](int b) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    UnitCtx& cx = lds.ctx[b];
    const int kb = __builtin_amdgcn_readfirstlane(cx.kb), ke = __builtin_amdgcn_readfirstlane(cx.ke);
    const int blk0 = kb >> 6;
    const int n_words = ((ke - 1) >> 6) - blk0 + 1;  // <= kMaskLanes
    int total = 0;
    for (int c = 0; c < n_words; ++c) {
      uint64_t wc = uniform_u64(cx.mw[c]);
      const int lo = (blk0 + c) << 6;
      if (kb > lo) wc &= ~0ull << (kb - lo);
      if (ke - lo < 64) wc &= (1ull << (ke - lo)) - 1ull;
      if ((wc >> lane) & 1ull) cx.klist[total + (int)lane_rank(wc)] = (uint16_t)((c << 6) + lane);
      total += __popcll(wc);
    }
    if (lane == 0) {
      cx.base = blk0 << 6;
      cx.n_items = total;
      cx.next = 0;
      cx.done = 0;
      cx.state = total > 0 ? kUnitReady : kUnitDrained;
    }
  };

  // ---- the loads issued a half-round ago have arrived: the first window of each unit
  auto complete_loads = [&]() {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    uint64_t loading = __ballot(lane < kBufs && lds.ctx[min(lane, kBufs - 1)].state == kUnitLoading);
    if (loading == 0ull) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the match records, facts and mask words are in LDS
    while (loading != 0ull) {
      const int b = (int)__builtin_ctzll(loading);
      loading &= loading - 1ull;
      UnitCtx& cx = lds.ctx[b];
      const int n_all = lds.prep[b].n_all;
      uint32_t thr = (uint32_t)rc.min_matches;                                       // :1094
      if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
      if (lane == 0) {
        cx.thr = thr;
        cx.w_it = 0; cx.w_real = 0; cx.w_valid = 0; cx.w_best_idx = -1; cx.w_best_n = 0;
        cx.w_rmse = 1e6f;   // :1112
        cx.w_state = I;
      }
      build_list(b);
    }
    lsync();
  };

  // ---- units whose window has drained.  Full speculation: the buffer is free.  Phased plans: the reference's in-order
  // bookkeeping (node.cpp:1171-1190) over what the pair has recorded so far -- unless nothing can follow the window anyway
  // -- and then either the pair's loop has ended / everything it can still need is recorded (the walk state goes to memory
  // for the result wave, WalkState::speculate = end of the recorded range), or the next window.
  auto advance_units = [&]() {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    uint64_t drained = __ballot(lane < kBufs && lds.ctx[min(lane, kBufs - 1)].state == kUnitDrained);
    if (drained == 0ull) return;
    while (drained != 0ull) {
      const int b = (int)__builtin_ctzll(drained);
      drained &= drained - 1ull;
      UnitCtx& cx = lds.ctx[b];
      if (!phased) {
        if (lane == 0) cx.state = kUnitFree;
        continue;
      }
      const uint32_t pair = (uint32_t)__builtin_amdgcn_readfirstlane((int)cx.pair);
      const int n_all = __builtin_amdgcn_readfirstlane(lds.prep[b].n_all);
      const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)cx.thr);
      int cls = __builtin_amdgcn_readfirstlane(cx.cls);
      WalkRegs wr{__builtin_amdgcn_readfirstlane(cx.w_it), __builtin_amdgcn_readfirstlane(cx.w_real),
                  __builtin_amdgcn_readfirstlane(cx.w_valid), __builtin_amdgcn_readfirstlane(cx.w_best_idx),
                  __builtin_amdgcn_readfirstlane(cx.w_best_n),
                  __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx.w_rmse))), false};
      int state = __builtin_amdgcn_readfirstlane(cx.w_state);
      int ke = __builtin_amdgcn_readfirstlane(cx.ke);
      bool finished = false;
      for (;;) {   // (a window without a viable iteration drains at once: the loop goes on to the next one)
        if (ke >= min(I, state)) { finished = true; break; }   // nothing can follow this window: the result wave walks
        // every record of the window is in memory: the workers wait for their stores before the barrier behind which the
        // server sees `done`; the server's own (iterations ended by a NaN refit) are awaited here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const IterSum* __restrict__ sum_pair = plan.sums + (size_t)pair * (size_t)I;
        const uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
        walk_records<true>(wr, ke, I, n_all, thr, sum_pair,
                           [&](int k) { return ((vm_pair[k >> 6] >> (k & 63)) & 1ull) != 0ull; }, lane);
        state = (!wr.done && wr.it < I) ? wr.real_iterations + (I - wr.it) : -1;
        if (state < 0) { finished = true; break; }
        if (cls < 0) {   // the first walk: 2 = nothing has jumped `it` ahead and at most 9 of 14 iterations were valid
          const bool no_jump = wr.it == wr.real_iterations;
          const bool junk_heavy = wr.valid_iterations * kClass2Den <= wr.real_iterations * kClass2Num;
          cls = no_jump ? (junk_heavy ? 2 : 1) : 0;
        }
        int target = I;
        if (cls != 2)
          for (int ph = plan.n_phases - 1; ph >= 0; --ph)
            if (plan.phase_ends[ph] > ke) target = plan.phase_ends[ph];
        const int end = min(min(target, state), I);
        if (ke >= end) { finished = true; break; }
        const int kb = ke;
        ke = min(end, kb + kMaxShare);
        // the window's words of the viable mask (written by the hypothesis kernel, a launch ago)
        const int blk0 = kb >> 6;
        const int n_w = min(kMaskLanes, plan.vmask_words - blk0);
        {   // (lane tests inside this loop: each opaque to the compiler on its own, see score_slot)
          const int ln = fresh(threadIdx.x & (kWave - 1));
          if (ln < n_w) cx.mw[ln] = vm_pair[blk0 + ln];
        }
        if (fresh(threadIdx.x & (kWave - 1)) == 0) { cx.kb = kb; cx.ke = ke; }
        lsync();
        build_list(b);
        lsync();
        if (__builtin_amdgcn_readfirstlane(cx.state) == kUnitReady) break;
      }
      if (lane == 0) {
        if (finished) {
          WalkState ws;
          ws.state = state; ws.it = wr.it; ws.real_iterations = wr.real_iterations; ws.valid_iterations = wr.valid_iterations;
          ws.best_idx = wr.best_idx; ws.best_n = wr.best_n; ws.rmse = wr.rmse;
          ws.speculate = ke;
          plan.walk[pair] = ws;
          cx.state = kUnitFree;
        } else {
          cx.cls = cls;
          cx.w_it = wr.it; cx.w_real = wr.real_iterations; cx.w_valid = wr.valid_iterations;
          cx.w_best_idx = wr.best_idx; cx.w_best_n = wr.best_n; cx.w_rmse = wr.rmse; cx.w_state = state;
        }
      }
    }
    lsync();
  };

  // ---- the 3x3 SVDs of the refits the workers left in group gs' mailbox (lane = slot): the transforms to score next
  auto serve_svd = [&](int gs) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    const int s = gs * kGroupSlots + min(lane, kGroupSlots - 1);
    SlotS& sl = lds.slot[s];
    const bool p = lane < kGroupSlots && sl.iter >= 0 && sl.active == kSlotActive;
    if (__ballot(p) == 0ull) return;
#ifdef RGBDFE_SPLIT_STATS
    { const int st_n = __popcll(__ballot(p)); ST_ADD(4, st_n) }   // (the ballot outside the macro's lane-0 branch)
#endif
    Tfc mine;
    mine.reset();  // lanes without a request: the zero matrix (no rotation, one sweep)
    if (p) {
      const float* __restrict__ in = sl.u.svd_in;
#pragma unroll
      for (int i = 0; i < 9; ++i) mine.C[i] = in[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) { mine.m1[i] = in[9 + i]; mine.m2[i] = in[12 + i]; }
    }
    float fR[9], ft[3];
    tfc_get_transformation(mine, fR, ft);
    if (p) {
#pragma unroll
      for (int i = 0; i < 9; ++i) sl.u.x.R[i] = fR[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) sl.u.x.t[i] = ft[i];
      if (has_nan12(fR, ft)) sl.active = kSlotDone;  // :1144: the iteration ends with what it has refined so far
    }
    lsync();
  };

  // ---- iterations of group gs that have left their refinement loop: the slot is free again, and the unit's window has
  // drained once all its iterations have ended
  auto recycle = [&](int gs) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    SlotS& sl = lds.slot[gs * kGroupSlots + min(lane, kGroupSlots - 1)];
    const bool fin = lane < kGroupSlots && sl.iter >= 0 && sl.active != kSlotActive;
    if (__ballot(fin) == 0ull) return;
    const int b = fin ? sl.buf : -1;
    if (fin) {
      if (sl.active == kSlotDone) write_summary(sl);  // (ended by a NaN refit: nobody has written its summary yet)
      sl.iter = -1;
      sl.active = kSlotDone;
    }
#pragma unroll
    for (int q = 0; q < kBufs; ++q) {
      const int c = __popcll(__ballot(b == q));
      if (c > 0 && lane == 0) {
        UnitCtx& cx = lds.ctx[q];
        cx.done += c;
        if (cx.done == cx.n_items) cx.state = kUnitDrained;
      }
    }
    lsync();
  };

  // ---- free slots of group gs take the next viable iterations of the resident units (any unit: a slot names its unit's
  // buffer), the least occupied workers first; the two groups are kept level
  auto hand_out = [&](int gs) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    int av = 0, nx = 0;
    if (lane < kBufs) {
      const UnitCtx& cx = lds.ctx[lane];
      nx = cx.next;
      av = cx.state == kUnitReady ? cx.n_items - nx : 0;
    }
    int avs[kBufs], nxs[kBufs], total = 0;
#pragma unroll
    for (int q = 0; q < kBufs; ++q) {
      avs[q] = __builtin_amdgcn_readlane(av, q);
      nxs[q] = __builtin_amdgcn_readlane(nx, q);
      total += avs[q];
    }
    if (total == 0) return;
    const bool in_g = lane < kGroupSlots;
    const bool held = in_g && lds.slot[gs * kGroupSlots + min(lane, kGroupSlots - 1)].iter >= 0;
    const bool held_o = in_g && lds.slot[(1 - gs) * kGroupSlots + min(lane, kGroupSlots - 1)].iter >= 0;
    const uint64_t held_m = __ballot(held), free_m = __ballot(in_g && !held);
    const int n_held = __popcll(held_m), n_free = kGroupSlots - n_held, n_held_o = __popcll(__ballot(held_o));
    // the share of this group: what levels the two groups (the other one takes its share in the next half-round),
    // everything when the other group is full
    int n_take = n_held_o == kGroupSlots ? total : (total + n_held_o - n_held + 1) / 2;
    n_take = max(0, min(n_take, min(n_free, total)));
    if (n_take == 0) return;
    const bool is_free = in_g && !held;
    const int ord = (int)lane_rank(free_m);   // (a slot belongs to no worker: the work lists are dealt pass by pass)
    const bool take = is_free && ord < n_take;
    ST_ADD(6, n_take)
    int b = 0, at = 0;
    {
      int before = 0;   // items of the buffers in front of buffer q
#pragma unroll
      for (int q = 0; q < kBufs; ++q) {
        if (take && ord >= before && ord < before + avs[q]) { b = q; at = nxs[q] + (ord - before); }
        if (lane == q) lds.ctx[q].next = nxs[q] + max(0, min(avs[q], n_take - before));   // items taken from buffer q
        before += avs[q];
      }
    }
    if (take) {
      const UnitCtx& cx = lds.ctx[b];
      SlotS& sl = lds.slot[gs * kGroupSlots + lane];
      // the 4-point hypothesis ransac_hyp_kernel left in the iteration's record: the first transform to score
      const int k = cx.base + (int)cx.klist[at];
      const float2* __restrict__ hyp = reinterpret_cast<const float2*>(plan.recs[(size_t)cx.pair * (size_t)I + (size_t)k].rR);  // rR[9], rt[3]
      float2 v[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) v[e] = hyp[e];
#pragma unroll
      for (int e = 0; e < 6; ++e) reinterpret_cast<float2*>(sl.u.x.R)[e] = v[e];  // -> R[9], t[3]
      // refined_error = 1e6 (:1133), written as its two words: as a double the constant was hoisted out of the server's loop into
      // a register PAIR, which the allocator cannot rematerialise -- the one value of the kernel that went to scratch
      // (... and the halves kept apart, or the compiler fuses them back into the pair)
      uint32_t rerr_hi = 0x412E8480u;
      asm volatile("" : "+v"(rerr_hi));
      reinterpret_cast<uint32_t*>(&sl.rerr)[0] = 0u;
      reinterpret_cast<uint32_t*>(&sl.rerr)[1] = rerr_hi;
      sl.rn = 0;      // :1134
      sl.active = kSlotActive;
      sl.round = 0;
      sl.iter = k;
      sl.buf = b;
      sl.pair = cx.pair;
      sl.n_all = lds.prep[b].n_all;
      sl.thr = cx.thr;
      sl.pmax = lds.prep[b].pmax;
    }
    lsync();
  };

  // ---- the refined matches the active slots of group gs hold (deal_work: whether the pass's refits are packed)
  auto sum_refined = [&](int gs) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    const SlotS& sl = lds.slot[gs * kGroupSlots + min(lane, kGroupSlots - 1)];
    int v = (lane < kGroupSlots && sl.iter >= 0 && sl.active == kSlotActive) ? sl.rn : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) lds.sum_rn[gs] = v;
    lsync();
  };

  // ---- the work of group gs' next pass: the list of its active slots (scorings by ticket are taken off it one by one) and
  // the DEAL -- every worker's slots of the pass.  A slot that already holds a refined set will pass the float prefilter
  // again (a scoring ~4x that of a junk hypothesis) and most likely be refitted: the dear slots go round the workers first,
  // the cheap ones then fill up the workers that got one dear slot less before they go round as well.  Closed form, lane =
  // slot (no step-by-step deal: that was measured and put the server on the critical path); a deal that would overflow a
  // worker's list (never with <= 28 active slots) is replaced by the plain round.
  auto deal_work = [&](int gs) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    const SlotS& sl = lds.slot[gs * kGroupSlots + min(lane, kGroupSlots - 1)];
    const bool a = lane < kGroupSlots && sl.iter >= 0 && sl.active == kSlotActive;
    const bool dear = a && sl.rn > 0;
    const uint64_t m = __ballot(a), dm = __ballot(dear), cm = m & ~dm;
    if (a) lds.act_list[gs][lane_rank(m)] = (uint8_t)lane;
    const int D = __popcll(dm), Cn = __popcll(cm);
    // what the pass's refits will cost: the sizes of the refined sets the dear slots hold (a recurrence walks its set) --
    // summed by the hand-out, which comes just before and has the registers for it (in here the reduction cost a spill)
    const bool packed = kPackFrom >= 0 && D >= 4 && __builtin_amdgcn_readfirstlane(lds.sum_rn[gs]) >= kPackFrom;
    if (lane == 0) {
      lds.n_act[gs] = D + Cn;
      lds.task[gs] = 0;
      lds.tickets[gs] = (packed || D >= kTicketsFrom) ? 1 : 0;
    }
    const int q = D / kWorkers, r = D - q * kWorkers;   // workers 0 .. r-1 hold q + 1 dear slots, the others q
    const int L = r > 0 ? kWorkers - r : 0;             // workers that are one dear slot short
    const int fill = kCostDear * L;                     // cheap slots that level them
    int w = 0, pos = 0;
    if (dear) {
      const int d = (int)lane_rank(dm);
      w = d % kWorkers;
      pos = d / kWorkers;
    } else if (a) {
      const int c = (int)lane_rank(cm);
      if (c < fill) {
        w = r + c % L;
        pos = q + c / L;
      } else {
        const int c2 = c - fill, n1 = min(fill, Cn);   // (n1: cheap slots dealt in the first stage)
        w = c2 % kWorkers;
        // entries of worker w before this one: its dear slots, its share of the first stage, the rounds before
        const int first = (w >= r && L > 0) ? (n1 / L + ((w - r) < n1 % L ? 1 : 0)) : 0;
        pos = (w < r ? q + 1 : q) + first + c2 / kWorkers;
      }
    }
    // A pass with much to refit (`packed`: the dear slots' refined sets add up to kPackFrom matches) has its bookkeeping and
    // refits PACKED -- the dear slots, the ones that will most likely be refitted, seven to a worker, the cheap ones behind
    // them -- and its scorings go out by ticket.  A worker's recurrence pass costs the same for one refit or seven (9 lanes
    // each; three refits per pass on average when they are dealt round), so the fewer workers run one the fewer instructions
    // the pass costs; the price is the second barrier of a ticket pass, which short recurrences (0.01 z^2: 41 steps) do not
    // repay and long ones (0.002 z^2: 164 steps) do.
    if (packed) {
      const int idx = dear ? (int)lane_rank(dm) : D + (int)lane_rank(cm);
      w = idx / kWaveSlots;
      pos = idx - w * kWaveSlots;
    }
    if (__ballot(a && pos >= kWaveSlots) != 0ull) {     // the plain round: position p of the list -> worker p mod 7
      const int p = (int)lane_rank(m);
      w = p % kWorkers;
      pos = p / kWorkers;
    }
    if (a) lds.wlist[gs][w][pos] = (uint8_t)lane;
    // entries per worker
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kWorkers; ++k) {
      const int ck = __popcll(__ballot(a && w == k));
      if (lane == k) cnt = ck;
    }
    if (lane < kWorkers) lds.wlist[gs][lane][7] = (uint8_t)cnt;
  };

  // the first units before the first half-round (the workers start with group 0)
  if (lane < 2) { lds.n_act[lane] = 0; lds.task[lane] = 0; lds.tickets[lane] = 0; }
  prefetch_block();
  adopt_block();
  issue_loads();
  complete_loads();
  advance_units();
  hand_out(0);
  sum_refined(0);
  deal_work(0);
  issue_loads();
  lds_barrier();
  ST_T0
  for (int h = 0;; ++h) {
    const int g = h & 1, gs = 1 - g;   // gs: the group nobody scores in this half-round
    const int by_ticket = __builtin_amdgcn_readfirstlane(lds.tickets[g]);
    ST_ADD(0, 1) ST_ADD(2, by_ticket) ST_ADD(3, lds.n_act[g])
    ST_LAP(15)
    serve_svd(gs);
    ST_LAP(8)
#ifndef RGBDFE_SPLIT_NO_EARLY_MID
    // A pass whose scorings go out by ticket has a barrier in its middle (every scoring done) -- and that barrier waits for this
    // wave too.  With all of the server's duties in front of it (13 us at 0.01 z^2) the workers, done with their tickets after
    // 8 us, stood there for the difference in every such half-round (a fifth of them at 0.01 z^2, more than half at 0.002 z^2).
    // So: the SVDs (which the other group's next scorings need, and which take as long as the tickets) and whatever tickets are
    // left in front of the barrier, everything else behind it, beside the workers' bookkeeping and refits.
    if (by_ticket) {
      if (kServerScores) score_tickets(g);
      ST_LAP(14)
      lds_barrier();
    }
#endif
    recycle(gs);
    ST_LAP(9)
    complete_loads();
    advance_units();
    ST_LAP(10)
    hand_out(gs);
    ST_LAP(11)
    sum_refined(gs);
    deal_work(gs);
    ST_LAP(12)
    issue_loads();
    ST_LAP(13)
    const int lane = fresh(threadIdx.x & (kWave - 1));
    bool busy = lane < kBufs && lds.ctx[min(lane, kBufs - 1)].state != kUnitFree;   // items to hand out, in flight, or a load
    busy = busy || (lane < kStreamSlots && lds.slot[min(lane, kStreamSlots - 1)].iter >= 0) ||
           (lane + kWave < kStreamSlots && lds.slot[min(lane + kWave, kStreamSlots - 1)].iter >= 0);
    const bool more = __ballot(busy) != 0ull || live != 0ull || nx_n != 0 || units_left;
    if (!more && lane == 0) lds.quit[g] = 1;
#ifdef RGBDFE_SPLIT_NO_EARLY_MID
    if (by_ticket) {
      if (kServerScores) score_tickets(g);   // its own duties done, the server scores like everybody else
      ST_LAP(14)
      lds_barrier();
    }
#endif
    lds_barrier();        // (the workers' bookkeeping and refits of group g)
    ST_LAP(15)
#ifdef RGBDFE_SPLIT_STATS
    {
      unsigned int ms = 0, mf = 0, mb = 0, ss = 0, sf = 0;
      for (int wk = 0; wk < kWorkers; ++wk) {
        const unsigned int a = lds.st_w[g][wk][0], b = lds.st_w[g][wk][1];
        ms = a > ms ? a : ms; mf = b > mf ? b : mf; mb = a + b > mb ? a + b : mb; ss += a; sf += b;
      }
      ST_ADD(30, ms) ST_ADD(31, ss / kWorkers) ST_ADD(32, mf) ST_ADD(33, sf / kWorkers) ST_ADD(34, mb) ST_ADD(35, (ss + sf) / kWorkers)
    }
#endif
    if (!more) { ST_ADD(1, 1) ST_MAX(7, h + 1) ST_OUT break; }
  }
}

void launch_ransac_hyp(const PairWork* work, uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan,
                       hipStream_t stream) {
  if (n_pairs == 0) return;   // (no iterations: the pairs' walk states are still initialised)
  hipLaunchKernelGGL(ransac_hyp_kernel, dim3(n_pairs), dim3(kHypThreads), 0, stream, work, n_pairs, rc, plan);
}

// Once per DEVICE, outside any stream capture (rgbdfe_create runs it with the context's device current): the refinement
// kernel's dynamic LDS exceeds the 64 KB a kernel gets by default, and that attribute belongs to the device's code object.
// Returns the device's CU count.
int ransac_split_init() {
  static int n_cus[64] = {};   // per device ordinal (a multi-device handle creates one context per device)
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> g(mu);
  if (n_cus[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
#ifdef RGBDFE_SPLIT_ONE_WG_PER_CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
#else
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_refine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(StreamLds));
#endif
    if (getenv("RGBDFE_SPLIT_VERBOSE")) {  // diagnostics: what the runtime makes of the kernel's resources
      int nb = -1;
      const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(ransac_refine_kernel), kStreamThreads, sizeof(StreamLds));
      fprintf(stderr, "ransac_refine_kernel: %zu B LDS per workgroup, %d workgroups per CU (%s), %d CUs (device %d)\n", sizeof(StreamLds), nb, hipGetErrorString(e), v, dev);
    }
    n_cus[dev] = v;
  }
  return n_cus[dev];
}

void launch_ransac_refine(uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan_in, hipStream_t stream) {
  SplitPlan plan = plan_in;
  // a unit's list of viable iterations holds kMaxShare entries: whatever the caller's plan says, no share is longer (the kernel
  // would write past the list -- found as a hang when a policy change dropped the caller's own clamp)
  if (!plan.phased && plan.share_iters > kMaxShare) {
    const int I = rc.ransac_iterations > 0 ? rc.ransac_iterations : 0;
    plan.n_shares = I > 0 ? (I + kMaxShare - 1) / kMaxShare : 1;
    plan.share_iters = I > 0 ? (I + plan.n_shares - 1) / plan.n_shares : kMaxShare;
  }
  const uint32_t units = n_pairs * (uint32_t)plan.n_shares;
  if (units == 0) return;
  // persistent workgroups: two per CU (LDS), each streaming units through its three LDS buffers; the units are handed out
  // through plan.unit_counter (zero at launch)
  const uint32_t max_wgs = (uint32_t)ransac_split_wgs();
#ifdef RGBDFE_SPLIT_ONE_WG_PER_CU   // diagnostics variant: more LDS than two workgroups of a CU can get
  const size_t lds_bytes = 100 * 1024;
#else
  const size_t lds_bytes = sizeof(StreamLds);
#endif
  hipLaunchKernelGGL(ransac_refine_kernel, dim3(units < max_wgs ? units : max_wgs), dim3(kStreamThreads), lds_bytes, stream,
                     n_pairs, rc, plan, units);
}

int ransac_split_words_per_pair(int ransac_iterations) {
  const int I = ransac_iterations > 0 ? ransac_iterations : 0;
  return 4 * ((I + kHypThreads - 1) / kHypThreads);  // a workgroup of the hypothesis kernel writes 4 words per pass
}
int ransac_split_max_share() { return kMaxShare; }
int ransac_split_wgs() { return kWgsPerCu * ransac_split_init(); }

}  // namespace rgbdfe

#ifdef RGBDFE_SPLIT_STATS
extern "C" int rgbdfe_debug_split_stats(unsigned long long* out40, int reset) {
  if (out40 && hipMemcpyFromSymbol(out40, HIP_SYMBOL(rgbdfe::g_split_stats), 320) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[40] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rgbdfe::g_split_stats), z, 320) != hipSuccess) return -1;
  }
  return 0;
}
#endif

