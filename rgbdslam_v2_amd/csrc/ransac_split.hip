// ransac_split.hip -- the recording stage of the record / replay RANSAC schedule as two kernels (gfx950).
//
// Replaces, per (newer node, older node) pair, the body of the RANSAC loop of Node::getRelativeTransformationTo
// (node.cpp:1130-1169): sample_matches_prefer_by_distance (node.cpp:1024-1047), getTransformFromMatches
// (transformation_estimation_euclidean.cpp:7-61), computeInliersAndError + errorFunction2 (node.cpp:968-1020,
// misc.cpp:697-770) and the refinement loop (node.cpp:1140-1169).  The in-order bookkeeping (node.cpp:1171-1190) stays with
// replay_walk_kernel, the result with select_ransac_kernel<kReplay> (select_ransac.hip).
//
// An iteration's outcome is a pure function of its index (counter-based sampling, D1), so the work of a batch is cut by
// WHAT IS DONE, not by pair:
//   ransac_hyp_kernel     LANE = ITERATION, one 256-thread workgroup per pair, all iterations of the pair at once: the
//                         4-point sample, the weighted fit, the 3x3 Jacobi SVD and the pre-screen (an upper bound of the
//                         matches that can pass errorFunction2's shortcut test; fewer than the inlier threshold => the
//                         iteration certainly ends with refined_matches empty, node.cpp:1148-1154).  Leaves, per pair, a
//                         bit mask of the viable iterations, their transforms (in the rR / rt fields of the iteration's
//                         record) and the empty summaries of the others.  No slot machinery, no scoring state.
//   ransac_refine_kernel  the refinement loops of the viable iterations only.  A workgroup = 8 waves = 2 pairs x 4
//                         waves: the 4 waves of a pair share ONE LDS copy of the pair's match records (PairPrep, 8.9 KB)
//                         and take the pair's viable iterations in turn (the k-th viable iteration goes to wave k mod 4).
//                         Every wave refines 7 iterations side by side (slots, refilled as iterations finish):
//                           scoring          LANE = MATCH; the inliers' errors stay in LDS and the reference's strictly
//                                            sequential error sum (node.cpp:1006) runs right behind the scoring from
//                                            broadcast LDS reads -- no error pool in global memory, no read-back;
//                           bookkeeping      LANE = SLOT (node.cpp:1154-1166);
//                           refit            the PCL weighted-mean recurrences of the wave's active slots share one
//                                            63-lane loop (9 state elements per slot);
//                           3x3 Jacobi SVD   LANE = SLOT OF THE WORKGROUP: a wave posts its slots' covariance / means in an
//                                            LDS mailbox and whichever wave finds the server lock free runs ONE SVD for
//                                            every request pending in the workgroup (56 slots), its own included.  At 7
//                                            lanes per wave the SVDs were 40 % of the recording stage's instructions;
//                                            requests that arrive while a server is busy ride with the next one, so the
//                                            batches grow with the load.
//                         LDS: 60 KB per workgroup => 16 waves per CU at <= 128 VGPRs (4 per SIMD).  Nothing but the
//                         hypothesis read and the record write touches global memory inside the loop, and the waves of a
//                         workgroup never meet at a barrier after the prologue (wave-local ordering: LDS executes a wave's
//                         operations in order).
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

// flags other waves of the workgroup poll: relaxed atomics on LDS (a plain ds_read / ds_write the compiler neither hoists
// out of a polling loop nor drops); ordering against the data they guard comes from lsync()
// a lane index (or anything derived from it) the compiler may not carry across this point: addresses formed from it are
// recomputed where they are used instead of being hoisted to the top of the kernel and spilled
__device__ __forceinline__ int fresh(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

constexpr int kHypThreads = 256;

// Optional phase timers of the refinement kernel (librgbdfe_prof.so, -DRGBDFE_PROFILE_PHASES): wall cycles per phase
// summed over the waves of all launches since the last reset.  Never enabled in the product build.
#ifdef RGBDFE_PROFILE_PHASES
// one row per wave (26 atomics per wave on shared counters clog the memory pipeline of the very waves being measured)
constexpr unsigned kSplitLogWaves = 1u << 18;
__device__ unsigned long long g_split_log[kSplitLogWaves][26];
__device__ unsigned int g_split_count;
__device__ unsigned int g_dbg_reopened;  // iterations opened whose record was not a fresh hypothesis
#define SP_DECL const uint64_t sp_rt0 = __builtin_amdgcn_s_memrealtime(); uint64_t sp_t0 = __builtin_readcyclecounter(); uint64_t sp[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SP_MARK(i) { const uint64_t sp_t1 = __builtin_readcyclecounter(); sp[i] += sp_t1 - sp_t0; sp_t0 = sp_t1; }
#define SP_COUNT(i, v) { sp[i] += (uint64_t)(v); }
#else
#define SP_DECL
#define SP_MARK(i)
#define SP_COUNT(i, v)
#endif

// Every spin wait of the refinement kernel is BOUNDED (round 4, after intermittent hangs: about one launch in 10^4 never
// ended -- a wave that had just taken or released the queue lock stopped making progress and its workgroup waited for it
// forever; see DESIGN.md 4.2b for the evidence and for what was ruled out).  A wave whose wait exceeds kSpinBound turns
// (milliseconds; normal waits are microseconds) raises plan.gave_up and ends; the waves that depend on it follow, the
// launch terminates, and the guarded select_ransac_kernel<kRecord> behind it (select_ransac.hip) records the phase
// instead -- same bytes.  The diagnostics build (-DRGBDFE_SPLIT_WATCHDOG, librgbdfe_wd.so) also records where the first
// wave stood and what the workgroup's shared state looked like (rgbdfe_debug_watchdog).
constexpr unsigned kSpinBound = 1u << 15;
__device__ unsigned int g_split_gave_up;   // launches' waves that gave up since the process started (rgbdfe_debug_split_gave_up)
__device__ int g_split_sabotage;           // tests: the next n refinement launches give up at once (rgbdfe_debug_split_sabotage)
#define WD_DECL unsigned wd_n = 0;
#define WD_RESET wd_n = 0;
#ifdef RGBDFE_SPLIT_WATCHDOG
__device__ unsigned int g_wd[64];
#define WD_SNAPSHOT(SITE)                                                                                 \
    if ((threadIdx.x & 63) == 0 && atomicCAS(&g_wd[0], 0u, (unsigned)(SITE)) == 0u) {                     \
      g_wd[1] = blockIdx.x; g_wd[2] = threadIdx.x >> 6; g_wd[3] = gridDim.x; g_wd[4] = n_units;           \
      g_wd[5] = (unsigned)lds.qlock; g_wd[6] = (unsigned)lds.lock; g_wd[7] = (unsigned)lds.no_more_units; \
      for (int b_ = 0; b_ < kBufs; ++b_) {                                                                \
        g_wd[8 + 4 * b_] = (unsigned)lds.ctx[b_].state; g_wd[9 + 4 * b_] = (unsigned)lds.ctx[b_].next;    \
        g_wd[10 + 4 * b_] = (unsigned)lds.ctx[b_].done; g_wd[11 + 4 * b_] = (unsigned)lds.ctx[b_].n_items; \
      }                                                                                                   \
      unsigned rq = 0, rq2 = 0;                                                                           \
      for (int q_ = 0; q_ < 32; ++q_) { rq |= lds.req[q_] ? 1u << q_ : 0u; rq2 |= lds.req[32 + q_] ? 1u << q_ : 0u; } \
      g_wd[20] = rq; g_wd[21] = rq2; g_wd[22] = *plan.unit_counter; g_wd[23] = (unsigned)plan.phase_index;   \
      for (int w_ = 0; w_ < kStreamWaves; ++w_) {                                                          \
        unsigned act = 0, held = 0;                                                                       \
        for (int s_ = 0; s_ < kSlots; ++s_) { act |= lds.w[w_].slot[s_].active ? 1u << s_ : 0u; held |= lds.w[w_].slot[s_].iter >= 0 ? 1u << s_ : 0u; } \
        g_wd[24 + w_] = act | (held << 8) | ((unsigned)lds.dbg[w_] << 16);                                \
      }                                                                                                   \
      g_wd[32] = n_pairs; g_wd[33] = (unsigned)plan.n_shares; g_wd[34] = (unsigned)plan.share_iters;      \
    }
#define WD_MARK(CODE) if ((threadIdx.x & 63) == 0) lds.dbg[threadIdx.x >> 6] = (CODE);
#define WD_OWNER (int)(threadIdx.x >> 6) + 1
#else
#define WD_SNAPSHOT(SITE)
#define WD_MARK(CODE)
#define WD_OWNER 1
#endif
#define WD_TICK(SITE)                                                                                     \
  if (++wd_n > kSpinBound) {                                                                              \
    WD_SNAPSHOT(SITE)                                                                                     \
    if ((threadIdx.x & 63) == 0) {                                                                        \
      atomicExch(plan.gave_up, 1);                                                                        \
      atomicAdd(&g_split_gave_up, 1u);                                                                    \
    }                                                                                                     \
    __builtin_amdgcn_endpgm();                                                                            \
  }

constexpr int kStreamWaves = 8;                       // waves of a refinement workgroup
constexpr int kStreamThreads = kStreamWaves * kWave;
constexpr int kStreamSlots = kStreamWaves * kSlots;   // slots of a workgroup: one lane each in the combined SVD
static_assert(kStreamSlots <= kWave, "the combined SVD is lane = slot of the workgroup");
constexpr int kBufs = 3;                              // units (pair, iteration range) resident in a workgroup's LDS
constexpr int kMaxShare = 512;                        // iterations of a unit at most (the host cuts longer ranges)
constexpr int kMVec = RGBDFE_MAX_MATCHES * kRec / 4;  // float4s of a pair's match records

// one RANSAC iteration in flight (see select_ransac.hip: Slot), with the facts of its unit the scoring needs
struct SlotS {
  float R[9], t[3];          // transform to score next (first the 4-point hypothesis, then the refits')
  int n_all;                 // the unit's pair: selected matches,
  uint32_t thr;              // inlier threshold (:1094-1098),
  float pmax;                // largest |coordinate| of its points,
  int fast;                  // every weight inside the window of the unscaled float division
  float rR[9], rt[3];        // refined_transformation (node.cpp:1137,1163)
  uint64_t rmask[kRounds];   // refined_matches
  uint64_t cmask[kRounds];   // inlier set of the scoring of the current round
  uint64_t fmask[kRounds];   // refined_matches without zero weights = the input of the next refit while the slot is active
  double rerr;               // refined_error
  double csum;               // sequential sum of the current scoring's inlier errors (node.cpp:1006)
  int rn;                    // refined_matches.size()
  int cn;                    // inliers of the current scoring
  int active;                // still inside the refinement loop (:1140-1169)
  int round;                 // refinement passes done
  int iter;                  // RANSAC iteration held by the slot, -1 = free
  int buf;                   // LDS buffer of the slot's unit
  uint32_t pair;             // the unit's pair
  int pad;
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
  union {  // the phases of a round never overlap
    ScoreB sc;
    FitBuf fit;
  } u;
  SlotS slot[kSlots];
};
// a unit resident in LDS: its pair's facts and the viable iterations of its range, handed out in order
constexpr int kUnitFree = 0, kUnitLoading = 1, kUnitReady = 2;
struct UnitCtx {
  int state;                 // kUnitFree / kUnitLoading (one wave is filling the buffer) / kUnitReady
  int next;                  // iterations handed out so far (may run past n_items)
  int done;                  // iterations whose refinement has ended; == n_items => the buffer is free again
  int n_items;
  uint32_t pair;
  int n_all;
  uint32_t thr;
  float pmax;
  int fast;
  int base;                  // iteration index of bit 0 of the first mask word of the unit's range
  int pad[2];
  uint64_t w_nonzero[kRounds];
  uint16_t klist[kMaxShare];  // the viable iterations of the unit's range, ascending, relative to `base` (< kMaxShare + 64)
};
struct alignas(16) StreamLds {
  float M[kBufs][RGBDFE_MAX_MATCHES * kRec];  // the resident units' match records (see PairPrep)
  WaveLds w[kStreamWaves];
  float svd_in[kStreamSlots][16];  // mailbox of the combined SVD: C[9], mean1[3], mean2[3] of a slot's refit
  UnitCtx ctx[kBufs];
  int req[kWave];                  // 1 = the slot's SVD is pending (relaxed workgroup atomics)
  int lock;                        // 1 = a wave is serving the pending requests
  int qlock;                       // 1 = a wave is taking iterations / choosing a buffer to fill
  int no_more_units;               // the launch's unit counter has run past the last unit
  int pad;
#ifdef RGBDFE_SPLIT_WATCHDOG
  int dbg[kStreamWaves];           // where each wave is (progress codes)
#endif
};
static_assert(sizeof(StreamLds) <= 80 * 1024, "two workgroups per CU");

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
                                        int& n_inl, double& sum) {
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
  int n_cand = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int m = r * kWave + lane;
    const float pxf = M[m * kRec + 0], pyf = M[m * kRec + 1], pzf = M[m * kRec + 2];
    const float qxf = M[m * kRec + 3], qyf = M[m * kRec + 4], qzf = M[m * kRec + 5];
    // node.cpp:994 (z == 0 skip) ; misc.cpp:712-717 (NaN -> DBL_MAX)
    const bool pre = (m < n_all) && !(pzf == 0.0f || qzf == 0.0f) && !(__builtin_isnan(pzf) || __builtin_isnan(qzf));
    const float f0 = __builtin_fmaf(R[0], pxf, __builtin_fmaf(R[1], pyf, __builtin_fmaf(R[2], pzf, tr[0]))) - qxf;
    const float f1 = __builtin_fmaf(R[3], pxf, __builtin_fmaf(R[4], pyf, __builtin_fmaf(R[5], pzf, tr[1]))) - qyf;
    const float f2 = __builtin_fmaf(R[6], pxf, __builtin_fmaf(R[7], pyf, __builtin_fmaf(R[8], pzf, tr[2]))) - qzf;
    const float dsq_f = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, f2 * f2));
    const bool sure_in = dsq_f < lo_f, sure_out = dsq_f > hi_f;
    bool cand = pre && sure_in;
    if (__ballot(pre && !(sure_in || sure_out)) != 0ull) {  // a lane too close to call: the round in double
      const double a0 = (double)pxf, a1 = (double)pyf, a2 = (double)pzf;
      const double d0 = (((Rd[0] * a0 + Rd[1] * a1) + Rd[2] * a2) + td[0]) - (double)qxf;
      const double d1 = (((Rd[3] * a0 + Rd[4] * a1) + Rd[5] * a2) + td[1]) - (double)qyf;
      const double d2 = (((Rd[6] * a0 + Rd[7] * a1) + Rd[8] * a2) + td[2]) - (double)qzf;
      const double dsq = (d0 * d0 + d1 * d1) + d2 * d2;
      cand = pre && !(dsq > shortcut) && !__builtin_isnan(d2);  // misc.cpp:731, 755
    }
    const uint64_t cm = __ballot(cand);
    if (cand) sb.cand[n_cand + (int)lane_rank(cm)] = (uint16_t)m;
    n_cand += __popcll(cm);
  }
  if (lane < 2 * kRounds) sb.mbits[lane] = 0u;
  lsync();
  n_inl = 0;
  sum = 0.0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) mask[r] = 0ull;
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
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  // the batch's counters (walk[n_pairs]: class-1 pairs, the refinement launches' unit counters, "still running" flag)
  // start at zero: done here, ahead of every kernel that uses them (a memset node captured into the batch's hipGraph
  // was not reliably in effect when the graph was replayed)
  if (pair == 0 && threadIdx.x < sizeof(WalkState) / 4) reinterpret_cast<uint32_t*>(plan.walk + n_pairs)[threadIdx.x] = 0u;
  const PairPrep* __restrict__ pp = plan.prep + pair;
  const int n_all = pp->n_all;
  // no RANSAC for this pair (node.cpp:1087, :1130)
  if (!(n_all > rc.min_matches && n_all >= 4)) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pp->M);
    float4* __restrict__ dst = reinterpret_cast<float4*>(M);
    for (int v = tid; v < kMVec; v += kHypThreads) dst[v] = src[v];
  }
  __syncthreads();
  const float pmax = pp->pmax;
  uint32_t thr = (uint32_t)rc.min_matches;                                         // :1094
  if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
  const uint32_t seed_uid = mix32(mix32(rc.seed ^ 0x9E3779B9u) + work[pair].uid * 0x85EBCA6Bu);
  const int I = rc.ransac_iterations;
  IterRec* __restrict__ rec_pair = plan.recs + (size_t)pair * (size_t)I;
  IterSum* __restrict__ sum_pair = plan.sums + (size_t)pair * (size_t)I;
  uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
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
    const uint32_t may_pass = prescreen_may_pass(hypR, hypt, M, n_all, pmax, rc);
    const bool viable = in_range && !has_nan12(hypR, hypt) && may_pass >= thr;
    const uint64_t vm = __ballot(viable);
    if (lane == 0) vm_pair[k >> 6] = vm;  // (k0 and the wave's first lane are multiples of 64)
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
    } else if (in_range) {
      sum_pair[k] = IterSum{1e6, 0, 0};
    }
  }
}

// ---------------------------------------------------------------------------------
// The refinement loops (node.cpp:1140-1169) of the viable iterations of [phase_begin, phase_end / spec_end), streamed:
// a workgroup takes units (pair, iteration range) off the launch's counter, keeps up to three of them resident in LDS and
// its 8 waves take whatever viable iteration is next, of any resident unit.
// ---------------------------------------------------------------------------------
#ifdef RGBDFE_SPLIT_NO_WAVES_ATTR
#define RGBDFE_REFINE_ATTR
#else
#define RGBDFE_REFINE_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
__global__ __launch_bounds__(kStreamThreads) RGBDFE_REFINE_ATTR void ransac_refine_kernel(
    uint32_t n_pairs, const RansacConst rc, const SplitPlan plan, uint32_t n_units) {
  // a later phase of a batch whose pairs have all ended (the walk of the phase before found nobody still running)
  if (plan.phase_index > 0 && plan.walk[n_pairs].best_n != plan.phase_index) return;
  extern __shared__ __attribute__((aligned(16))) char stream_smem[];
  StreamLds& lds = *reinterpret_cast<StreamLds*>(stream_smem);
  SP_DECL
  WD_DECL
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  WaveLds& wl = lds.w[wave];
  const int I = rc.ransac_iterations;

  if (lane < kSlots) { wl.slot[lane].active = 0; wl.slot[lane].iter = -1; }
  if (wave == 0) {
    lds.req[lane] = 0;
    if (lane < kBufs) { lds.ctx[lane].state = kUnitFree; lds.ctx[lane].next = 0; lds.ctx[lane].done = 0; lds.ctx[lane].n_items = 0; }
    if (lane == 0) {
      lds.lock = 0;
      lds.qlock = 0;
      lds.no_more_units = 0;
    }
#ifdef RGBDFE_SPLIT_WATCHDOG
    if (lane < kStreamWaves) lds.dbg[lane] = 1;
#endif
  }
  __syncthreads();  // the only workgroup barrier: from here on the waves run on their own
  SP_MARK(0)
  // Test hook (rgbdfe_debug_split_sabotage): the first wave of the launch gives up before doing anything, as a wave whose
  // wait reached its bound would -- the launch's records are void and the guarded fallback launch has to produce them.
  if (blockIdx.x == 0 && wave == 0 && g_split_sabotage > 0) {
    if (lane == 0) {
      atomicSub(&g_split_sabotage, 1);
      atomicExch(plan.gave_up, 1);
      atomicAdd(&g_split_gave_up, 1u);
    }
    __builtin_amdgcn_endpgm();
  }

  // ---- a unit -> LDS buffer b (one wave): the pair's facts, its match records (global -> LDS directly:
  // global_load_lds_dwordx4, 64 x 16 bytes per instruction), the viable iterations of the unit's range as a list.
  // Leaves the buffer ready, or free again when the unit has nothing to do in this launch.
  auto load_unit = [&](int b, uint32_t unit) {
    const int lane = fresh(threadIdx.x & (kWave - 1));
    const uint32_t pair = unit / (uint32_t)plan.n_shares;
    const int share = (int)(unit - pair * (uint32_t)plan.n_shares);
    const PairPrep* __restrict__ pp = plan.prep + pair;
    const uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
    UnitCtx& cx = lds.ctx[b];
    auto load_records = [&]() {
#ifdef RGBDFE_SPLIT_NO_LDSDMA   // diagnostics variant: the records through registers
      const float4* __restrict__ src4 = reinterpret_cast<const float4*>(pp->M);
      float4* __restrict__ dst4 = reinterpret_cast<float4*>(lds.M[b]);
      for (int i = 0; i < (kMVec + kWave - 1) / kWave; ++i) {
        const int v = i * kWave + lane;
        if (v < kMVec) dst4[v] = src4[v];
      }
#else
      const char* __restrict__ src = reinterpret_cast<const char*>(pp->M);
      for (int i = 0; i < (kMVec + kWave - 1) / kWave; ++i) {
        const int v = i * kWave + lane;
        if (v < kMVec)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)v * 16),
                                           (__attribute__((address_space(3))) void*)(lds.M[b] + i * (kWave * 4)), 16, 0, 0);
      }
#endif
    };
    // the first launch of a batch: every pair is still running, the records need not wait for the pair's state
    if (plan.phase_begin == 0) load_records();
    // walk[pair].state >= 0: upper bound of the iterations the pair can still need; < 0: its loop has ended
    const WalkState ws = plan.walk[pair];
    const int batch_class1 = plan.walk[n_pairs].state;
    const int pre = plan.first_spec ? (int)plan.preclass[pair] : 0;
    const int n_all = __builtin_amdgcn_readfirstlane(pp->n_all);
    const float pmax = pp->pmax;
    const uint32_t fast = pp->fast_alpha;
    const uint64_t wnz = lane < kRounds ? pp->w_nonzero[lane] : 0ull;
    const int k_first = plan.phase_begin + share * plan.share_iters;
    const int blk0 = k_first >> 6;
    const uint64_t wv_ld = (blk0 + lane < plan.vmask_words && lane <= kMaxShare / kWave) ? vm_pair[blk0 + lane] : 0ull;
    const int pair_state = plan.phase_begin == 0 ? I : __builtin_amdgcn_readfirstlane(ws.state);
    // class 2 (no jump of `it` so far, junk-heavy; class 1 when the batch has few such pairs: effective_class): everything
    // that is left is recorded in this launch
    int cls = plan.phase_begin != 0 ? __builtin_amdgcn_readfirstlane(ws.speculate) : __builtin_amdgcn_readfirstlane(pre);
    if (cls == 1) cls = ((uint32_t)__builtin_amdgcn_readfirstlane(batch_class1) * 64u <= n_pairs) ? 2 : 0;
    const int end = min(cls == 2 ? plan.spec_end : plan.phase_end, pair_state);
    const int k_begin = k_first;
    const int k_end = min(k_begin + plan.share_iters, end);
    // no RANSAC for this pair (node.cpp:1087, :1130), or nothing of this range is needed (any more)
    const bool have = pair_state >= 0 && k_begin < k_end && n_all > rc.min_matches && n_all >= 4;
    int total = 0;
    if (have) {
      if (plan.phase_begin != 0) load_records();
      // lane w holds word blk0 + w of the pair's mask, cut to the range; lane = iteration builds the list
      uint64_t wv = wv_ld;
      {
        const int lo = (blk0 + lane) << 6;
        if (k_begin > lo) wv &= (k_begin - lo >= 64) ? 0ull : (~0ull << (k_begin - lo));
        if (k_end - lo < 64) wv &= (k_end - lo <= 0) ? 0ull : ((1ull << (k_end - lo)) - 1ull);
      }
      const int n_words = ((k_end - 1) >> 6) - blk0 + 1;  // <= kMaxShare / 64 + 1
      for (int c = 0; c < n_words; ++c) {
        const uint64_t wc = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wv >> 32), c) << 32) |
                            (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wv, c);
        if ((wc >> lane) & 1ull) cx.klist[total + (int)lane_rank(wc)] = (uint16_t)((c << 6) + lane);
        total += __popcll(wc);
      }
      uint32_t thr = (uint32_t)rc.min_matches;                                         // :1094
      if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
      if (lane < kRounds) cx.w_nonzero[lane] = wnz;
      if (lane == 0) {
        cx.pair = pair;
        cx.n_all = n_all;
        cx.thr = thr;
        cx.pmax = pmax;
        cx.fast = (int)fast;
        cx.base = blk0 << 6;
        cx.n_items = total;
        cx.next = 0;
        cx.done = 0;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the match records are in LDS
    lsync();
    if (lane == 0) flag_store(&cx.state, total > 0 ? kUnitReady : kUnitFree);
    SP_COUNT(21, 1)
  };

  // ---- iterations that have left their refinement loop: the outcome record; the slot is free again, and so is the
  // unit's buffer once all its iterations have ended
  auto close_finished = [&]() {
    WD_MARK(50)
    const int lane = fresh(threadIdx.x & (kWave - 1));
    if (lane < kSlots) {
      SlotS& sl = wl.slot[lane];
      if (sl.iter >= 0 && !sl.active) {
        const size_t at = (size_t)sl.pair * (size_t)I + (size_t)sl.iter;
        IterRec& r = plan.recs[at];
#pragma unroll
        for (int i = 0; i < 9; ++i) r.rR[i] = sl.rR[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) r.rt[i] = sl.rt[i];
#pragma unroll
        for (int q = 0; q < kRounds; ++q) r.rmask[q] = sl.rmask[q];
        r.rerr = sl.rerr;
        r.rn = sl.rn;
        r.pad = 0;
        plan.sums[at] = IterSum{sl.rerr, sl.rn, 0};
        sl.iter = -1;
        UnitCtx& cx = lds.ctx[sl.buf];
        const int n_items = cx.n_items;  // (read before the count: once the last iteration is counted the buffer may be refilled)
        if (atomicAdd(&cx.done, 1) + 1 == n_items) flag_store(&cx.state, kUnitFree);
      }
    }
    lsync();
    SP_MARK(1)
  };

  // ---- free slots take the next viable iterations of the resident units (any unit: a slot carries its unit's facts);
  // a wave that finds nothing to take brings the workgroup's next unit in.  Returns false when the wave holds no
  // iteration and none will come.
  auto refill = [&]() -> bool {
    WD_RESET
    const int lane = fresh(threadIdx.x & (kWave - 1));
    uint64_t free_slots = __ballot(lane < kSlots && wl.slot[min(lane, kSlots - 1)].iter < 0);
    const int n_free = __popcll(free_slots);
    const bool had = n_free < kSlots;
    int got = 0;
    int my_g = -1, my_at = 0, my_b = 0;  // lanes 6j .. 6j+5: the j-th slot opened by this call
    while (got < n_free) {
      // The hand-out of iterations and the choice of a buffer to fill happen under the workgroup's queue lock: a unit
      // with iterations left cannot be recycled, so what a wave sees inside the lock is what it takes.  (Lock-free
      // claims on `next` could land on a buffer that had been drained, freed and refilled in between.)
      int locked = 0;
      do {
#ifdef RGBDFE_SPLIT_TTAS   // test-and-test-and-set: a plain read first, the returning atomic only when the lock looks free
        if (lane == 0) locked = flag_load(&lds.qlock) == 0 ? (atomicCAS(&lds.qlock, 0, WD_OWNER) == 0 ? 1 : 0) : 0;
#else
        if (lane == 0) locked = atomicCAS(&lds.qlock, 0, WD_OWNER) == 0 ? 1 : 0;
#endif
        locked = __builtin_amdgcn_readfirstlane(locked);
        if (!locked) __builtin_amdgcn_s_sleep(1);
        WD_TICK(1)
      } while (!locked);
      WD_MARK(10)
      asm volatile("" ::: "memory");
      int st = kUnitLoading, nx = 0, ni = 0;
      if (lane < kBufs) {
        st = flag_load(&lds.ctx[lane].state);
        nx = flag_load(&lds.ctx[lane].next);
        ni = flag_load(&lds.ctx[lane].n_items);
      }
      const uint64_t avail = __ballot(lane < kBufs && st == kUnitReady && nx < ni);
      const bool units_left = flag_load(&lds.no_more_units) == 0;
      const uint64_t free_bufs = __ballot(lane < kBufs && st == kUnitFree);
      const uint64_t loading = __ballot(lane < kBufs && st == kUnitLoading);
      int b = -1, idx = 0, cnt = 0;
      bool fill = false;
      if (avail != 0ull) {
        b = (int)__builtin_ctzll(avail);
        idx = __builtin_amdgcn_readlane(nx, b);
        cnt = min(n_free - got, __builtin_amdgcn_readlane(ni, b) - idx);
        if (lane == 0) flag_store(&lds.ctx[b].next, idx + cnt);
      } else if (units_left && free_bufs != 0ull) {
        b = (int)__builtin_ctzll(free_bufs);
        fill = true;
        if (lane == 0) flag_store(&lds.ctx[b].state, kUnitLoading);
      }
      lsync();
      // (Diagnostics variant -DRGBDFE_SPLIT_UNLOCK_ALL_LANES: the release as a store of the same word by every lane.  In the
      // watchdog build it ran 4447 iterations of the stress loop without a stall where the single-lane forms stalled within
      // 30 .. 886; in the product build it made stalls ~10x rarer, not impossible -- the effect is one of timing, the cause
      // of the stall is still open -- and the 64-lane same-address store is not free, so it is not the default.)
#if defined(RGBDFE_SPLIT_UNLOCK_ALL_LANES)
      flag_store(&lds.qlock, 0);
#else
      if (lane == 0) atomicExch(&lds.qlock, 0);
#endif
      WD_MARK(11)
      if (cnt > 0) {
        for (int j = 0; j < cnt; ++j) {
          const int g = (int)__builtin_ctzll(free_slots);
          free_slots &= free_slots - 1ull;
          if (lane / 6 == got + j) { my_g = g; my_at = idx + j; my_b = b; }
        }
        got += cnt;
        continue;
      }
      if (fill) {
        // the launch's units are handed out one by one, to whichever workgroup has a buffer free (pairs differ a lot in
        // their work, neighbours alike: equal shares of the pair list would leave half the chip waiting for the rest)
        uint32_t unit = 0;
        if (lane == 0) unit = atomicAdd(plan.unit_counter, 1u);
        unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
        if (unit < n_units) {
          WD_MARK(20)
          load_unit(b, unit);
          WD_MARK(21)
        } else if (lane == 0) {
          flag_store(&lds.no_more_units, 1);
          flag_store(&lds.ctx[b].state, kUnitFree);
        }
        continue;
      }
      if (!units_left && loading == 0ull) break;  // nothing more will come
      if (had || got > 0) break;                  // this wave has work: it looks again after the round
#ifdef RGBDFE_SPLIT_IDLE_EXIT   // diagnostics variant: a wave without work does not poll, it leaves (while somebody else loads or works)
      if (loading != 0ull || free_bufs == 0ull) break;
#endif
      __builtin_amdgcn_s_sleep(4);                // idle: a loader is at work, or every buffer is still in use
      WD_TICK(2)
    }
    SP_MARK(2)
    SP_COUNT(12, got)
    WD_MARK(12)
    if (my_g >= 0) {
      const int e2 = lane % 6;
      const UnitCtx& cx = lds.ctx[my_b];
      SlotS& sl = wl.slot[my_g];
      const int k = cx.base + (int)cx.klist[my_at];
      const uint32_t pair = cx.pair;
      const float2 v = reinterpret_cast<const float2*>(plan.recs[(size_t)pair * (size_t)I + (size_t)k].rR)[e2];  // rR[9], rt[3]
      reinterpret_cast<float2*>(sl.R)[e2] = v;                                                                  // -> R[9], t[3]
#ifdef RGBDFE_PROFILE_PHASES
      if (e2 == 0 && plan.recs[(size_t)pair * (size_t)I + (size_t)k].rn != -77) atomicAdd(&g_dbg_reopened, 1u);
#endif
      if (e2 == 0) {
        const float IR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int i = 0; i < 9; ++i) sl.rR[i] = IR[i];  // :1137 refined = Identity
#pragma unroll
        for (int i = 0; i < 3; ++i) sl.rt[i] = 0.f;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) { sl.rmask[r] = 0ull; sl.fmask[r] = 0ull; }
        sl.rerr = 1e6;  // :1133
        sl.rn = 0;      // :1134
        sl.active = 1;
        sl.round = 0;
        sl.iter = k;
        sl.buf = my_b;
        sl.pair = pair;
        sl.n_all = cx.n_all;
        sl.thr = cx.thr;
        sl.pmax = cx.pmax;
        sl.fast = cx.fast;
      }
    }
    lsync();
    SP_MARK(3)
    return had || got > 0;
  };

  unsigned wd_rounds = 0;
  for (bool occupied = refill(); occupied; close_finished(), occupied = refill()) {
    SP_COUNT(13, 1)
    if (++wd_rounds > (1u << 16)) { wd_n = kSpinBound; WD_TICK(4) }   // (a wave holds at most a few thousand iterations' rounds)
    // ================================ one pass of the refinement loop (:1140) for every active slot
    // ---- scorings (:1148), one after the other (lane = match), each followed by its error sum
    WD_MARK(30)
    uint64_t act = __ballot(lane < kSlots && wl.slot[min(lane, kSlots - 1)].active != 0);
    while (act != 0ull) {
      const int g = (int)__builtin_ctzll(act);
      act &= act - 1ull;
      SlotS& sl = wl.slot[g];
      float curR[9], curt[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) curR[i] = sl.R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) curt[i] = sl.t[i];
      const int n_all = __builtin_amdgcn_readfirstlane(sl.n_all);
      const uint32_t thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.thr);
      const float pmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sl.pmax)));
      const float* __restrict__ M = lds.M[__builtin_amdgcn_readfirstlane(sl.buf)];
      // a scoring with fewer inliers than max(threshold, refined_matches.size()) is rejected whatever its error
      // is (:1154, :1160): the scorer may stop counting as soon as that is certain
      const uint32_t need = max(thr, (uint32_t)__builtin_amdgcn_readfirstlane(sl.rn));
      uint64_t inl_mask[kRounds];
      int n_inl;
      double sum;
      score_b(curR, curt, M, n_all, need, rc, wl.u.sc, pmax, inl_mask, n_inl, sum);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < kRounds; ++r) sl.cmask[r] = inl_mask[r];
        sl.cn = n_inl;
        sl.csum = sum;
      }
      SP_COUNT(14, 1)
    }
    lsync();
    SP_MARK(4)
    // ---- the loop's bookkeeping (:1154-1166), lane = slot
    bool still = false;
    if (fresh(lane) < kSlots) {
      SlotS& sl = wl.slot[fresh(lane)];
      if (sl.active) {
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
            const UnitCtx& cx = lds.ctx[sl.buf];
#pragma unroll
            for (int i = 0; i < 9; ++i) sl.rR[i] = sl.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sl.rt[i] = sl.t[i];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
              const uint64_t m = sl.cmask[r];
              sl.rmask[r] = m;
              // tfc.add skips weight == 0; NaN depths never reach an inlier set (misc.cpp:712-717)
              sl.fmask[r] = m & cx.w_nonzero[r];
            }
            sl.rn = n_inl;
            sl.rerr = err_mine;
          }
        }
        if (sl.round == 18) still = false;  // the 19th pass was the last one (:1140)
        sl.round++;
        sl.active = still ? 1 : 0;
      }
    }
    uint64_t refit = __ballot(still);
    lsync();
    SP_MARK(5)
    if (refit == 0ull) continue;
    // ---- refits (:1142): the weighted-mean recurrences of the wave's active slots side by side ...
    // (the unscaled division of the recurrence needs every slot's pair inside its window)
    const bool all_fast = __ballot(still && wl.slot[min(lane, kSlots - 1)].fast == 0) == 0ull;
    int n_mine = 0, k256_mine = 0, n_max = 0, n_min = RGBDFE_MAX_MATCHES;
    {
      const uint64_t ones[kRounds] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
      uint64_t todo = refit;
      while (todo != 0ull) {
        const int g = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        SlotS& sl = wl.slot[g];
        uint64_t m5[kRounds];
#pragma unroll
        for (int r = 0; r < kRounds; ++r) m5[r] = uniform_u64(sl.fmask[r]);
        int k256_g;
        const int n_g = fit_compact(g, m5, ones, wl.u.fit, k256_g, fresh(lane));
        if (lane / 9 == g) { n_mine = n_g; k256_mine = k256_g; }
        n_max = max(n_max, n_g);
        n_min = min(n_min, n_g);
      }
    }
    lsync();
    SP_MARK(6)
    {
      const int lane_here = fresh(lane);
      const int s = min(lane_here / 9, kSlots - 1), x = lane_here % 9;
      const float* __restrict__ M = lds.M[wl.slot[s].buf];  // (per lane: the records of the lane's slot's unit)
      float C, m1, m2;
      if (all_fast) fit_recurrence<true>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2, lane_here);
      else fit_recurrence<false>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2, lane_here);
      // lane 9s+x holds C[x], lane 9s+j mean1[j], lane 9s+3i mean2[i] of slot s: into the slot's mailbox
      if (lane_here < 9 * kSlots && ((refit >> s) & 1ull)) {
        float* __restrict__ in = lds.svd_in[wave * kSlots + s];
        in[x] = C;
        if (x < 3) in[9 + x] = m1;
        if (x % 3 == 0) in[12 + x / 3] = m2;
      }
    }
    lsync();
    SP_MARK(7)
    // ---- ... then their 3x3 SVDs, combined over the workgroup: post the requests, then serve whatever is pending
    // (every wave's, this one's included) if no other wave is serving, else wait for the server
    const int my_req = wave * kSlots + min(lane, kSlots - 1);
    if (lane < kSlots && ((refit >> lane) & 1ull)) flag_store(&lds.req[my_req], 1);
    lsync();
    WD_MARK(40)
    for (;;) {
      const bool pending = lane < kSlots && flag_load(&lds.req[my_req]) != 0;
      if (__ballot(pending) == 0ull) break;
      int got = 0;
      if (lane == 0) got = atomicCAS(&lds.lock, 0, 1) == 0 ? 1 : 0;
      got = __builtin_amdgcn_readfirstlane(got);
      if (got) {
        SP_MARK(8)
        asm volatile("" ::: "memory");
        const bool p = lane < kStreamSlots && flag_load(&lds.req[lane]) == 1;
        if (__ballot(p) != 0ull) {
          Tfc mine;
          mine.reset();  // lanes without a request: the zero matrix (no rotation, one sweep)
          if (p) {
            const float* __restrict__ in = lds.svd_in[lane];
#pragma unroll
            for (int i = 0; i < 9; ++i) mine.C[i] = in[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) { mine.m1[i] = in[9 + i]; mine.m2[i] = in[12 + i]; }
          }
          float fR[9], ft[3];
          tfc_get_transformation(mine, fR, ft);
          const bool fnan = has_nan12(fR, ft);
          if (p) {
            SlotS& sl = lds.w[lane / kSlots].slot[lane % kSlots];
#pragma unroll
            for (int i = 0; i < 9; ++i) sl.R[i] = fR[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sl.t[i] = ft[i];
            if (fnan) sl.active = 0;  // :1144
          }
          lsync();  // the transforms are in LDS before the flags say so
          if (p) flag_store(&lds.req[lane], 0);
          SP_COUNT(11, __popcll(__ballot(p)))
        }
        lsync();
#if defined(RGBDFE_SPLIT_UNLOCK_ALL_LANES)
        flag_store(&lds.lock, 0);
#else
        if (lane == 0) atomicExch(&lds.lock, 0);
#endif
        SP_COUNT(15, 1)
        SP_MARK(9)
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
      WD_TICK(3)
    }
    WD_RESET
    asm volatile("" ::: "memory");
    SP_MARK(8)
  }
  WD_MARK(99)
#ifdef RGBDFE_PROFILE_PHASES
  SP_MARK(10)
  if (lane == 0) {
    const unsigned row = atomicAdd(&g_split_count, 1u) % kSplitLogWaves;
    for (int i = 0; i < 22; ++i) g_split_log[row][i] = sp[i];
    g_split_log[row][22] = sp_rt0;
    g_split_log[row][23] = __builtin_amdgcn_s_memrealtime();
    g_split_log[row][24] = (unsigned long long)blockIdx.x * 8ull + (unsigned long long)wave;
    g_split_log[row][25] = sp[13] != 0 ? 1ull : 0ull;
  }
#endif
}

void launch_ransac_hyp(const PairWork* work, uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan,
                       hipStream_t stream) {
  if (n_pairs == 0 || rc.ransac_iterations <= 0) return;
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

void launch_ransac_refine(uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan, hipStream_t stream) {
  const uint32_t units = n_pairs * (uint32_t)plan.n_shares;
  if (units == 0) return;
  // persistent workgroups: two per CU (LDS), each streaming units through its three LDS buffers; the units are handed out
  // through plan.unit_counter (zero at launch)
  const int n_cus = ransac_split_init();
  const uint32_t max_wgs = 2u * (uint32_t)n_cus;
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

}  // namespace rgbdfe

#ifdef RGBDFE_PROFILE_PHASES
// diagnostics build only (librgbdfe_prof.so): wall cycles per phase summed over the refinement kernel's waves
extern "C" int rgbdfe_debug_reopened() {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(rgbdfe::g_dbg_reopened), sizeof(n)) != hipSuccess) return -1;
  return (int)n;
}
extern "C" int rgbdfe_debug_split_rows(unsigned long long* out, unsigned max_rows) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(rgbdfe::g_split_count), sizeof(n)) != hipSuccess) return -1;
  if (n > rgbdfe::kSplitLogWaves) n = rgbdfe::kSplitLogWaves;
  if (n > max_rows) n = max_rows;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(rgbdfe::g_split_log), (size_t)n * 26 * 8) != hipSuccess) return -1;
  return (int)n;
}
extern "C" int rgbdfe_debug_split_totals(unsigned long long* out32, int reset) {
  if (out32) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(rgbdfe::g_split_count), sizeof(n)) != hipSuccess) return -1;
    if (n > rgbdfe::kSplitLogWaves) n = rgbdfe::kSplitLogWaves;
    std::vector<unsigned long long> rows((size_t)n * 26);
    if (n && hipMemcpyFromSymbol(rows.data(), HIP_SYMBOL(rgbdfe::g_split_log), rows.size() * 8) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) out32[i] = 0;
    for (unsigned r = 0; r < n; ++r)
      for (int i = 0; i < 26; ++i) out32[i] += rows[(size_t)r * 26 + i];
  }
  if (reset) {
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rgbdfe::g_split_count), &zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// waves of refinement launches that gave up on a spin wait since the process started (the phases concerned were recorded
// by the fallback launch); -1 = could not be read
extern "C" int rgbdfe_debug_split_gave_up() {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(rgbdfe::g_split_gave_up), sizeof(n)) != hipSuccess) return -1;
  return (int)n;
}

// tests: make the next n refinement launches of this device give up at once (their phases then come from the fallback launch)
extern "C" int rgbdfe_debug_split_sabotage(int n) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rgbdfe::g_split_sabotage), &n, sizeof(n)) == hipSuccess ? 0 : -1;
}

#ifdef RGBDFE_SPLIT_WATCHDOG
// diagnostics build only: the watchdog record (word 0 = the site of the wait that never ended, 0 = none), optionally cleared
extern "C" int rgbdfe_debug_watchdog(unsigned int* out64, int reset) {
  if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(rgbdfe::g_wd), 64 * sizeof(unsigned int)) != hipSuccess) return -1;
  if (reset) {
    unsigned int z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rgbdfe::g_wd), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
