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
#define SP_DECL uint64_t sp_t0 = __builtin_readcyclecounter(); uint64_t sp[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SP_MARK(i) { const uint64_t sp_t1 = __builtin_readcyclecounter(); sp[i] += sp_t1 - sp_t0; sp_t0 = sp_t1; }
#define SP_COUNT(i, v) { sp[i] += (uint64_t)(v); }
#else
#define SP_DECL
#define SP_MARK(i)
#define SP_COUNT(i, v)
#endif

#ifndef RGBDFE_SPLIT_UNITS
#define RGBDFE_SPLIT_UNITS 2
#endif
#ifndef RGBDFE_SPLIT_WAVES
#define RGBDFE_SPLIT_WAVES 4
#endif
constexpr int kUnitsPerWg = RGBDFE_SPLIT_UNITS;    // (pair, share) units per workgroup
constexpr int kWavesPerUnit = RGBDFE_SPLIT_WAVES;  // waves that share a unit's match records
constexpr int kSplitWaves = kUnitsPerWg * kWavesPerUnit;
constexpr int kSplitThreads = kSplitWaves * kWave;
constexpr int kSplitSlots = kSplitWaves * kSlots;  // slots of a workgroup: one lane each in the combined SVD
static_assert(kSplitSlots <= kWave, "the combined SVD is lane = slot of the workgroup");
constexpr int kMVec = RGBDFE_MAX_MATCHES * kRec / 4;  // float4s of a pair's match records

// one RANSAC iteration in flight (see select_ransac.hip: Slot)
struct SlotB {
  float R[9], t[3];          // transform to score next (first the 4-point hypothesis, then the refits')
  float rR[9], rt[3];        // refined_transformation (node.cpp:1137,1163)
  uint64_t rmask[kRounds];   // refined_matches = the input of the next refit while the slot is active
  uint64_t cmask[kRounds];   // inlier set of the scoring of the current round
  double rerr;               // refined_error
  double csum;               // sequential sum of the current scoring's inlier errors (node.cpp:1006)
  int rn;                    // refined_matches.size()
  int cn;                    // inliers of the current scoring
  int active;                // still inside the refinement loop (:1140-1169)
  int round;                 // refinement passes done
  int iter;                  // RANSAC iteration held by the slot, -1 = free
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
  SlotB slot[kSlots];
};
struct alignas(16) SplitLds {
  float M[kUnitsPerWg][RGBDFE_MAX_MATCHES * kRec];  // the units' match records (see PairPrep)
  WaveLds w[kSplitWaves];
  float svd_in[kSplitSlots][16];  // mailbox of the combined SVD: C[9], mean1[3], mean2[3] of a slot's refit
  int req[kWave];                 // 1 = the slot's SVD is pending (accessed with relaxed workgroup atomics)
  int lock;                       // 1 = a wave is serving the pending requests
};
static_assert(sizeof(SplitLds) <= 80 * 1024, "two workgroups per CU");

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
    } else if (in_range) {
      sum_pair[k] = IterSum{1e6, 0, 0};
    }
  }
}

// ---------------------------------------------------------------------------------
// The refinement loops (node.cpp:1140-1169) of the viable iterations of [phase_begin, phase_end / spec_end).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kSplitThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void ransac_refine_kernel(
    uint32_t n_pairs, const RansacConst rc, const SplitPlan plan) {
  __shared__ SplitLds lds;
  SP_DECL
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int u = wave / kWavesPerUnit, wu = wave % kWavesPerUnit;
  WaveLds& wl = lds.w[wave];
  const int I = rc.ransac_iterations;

  // ---- this wave's unit: a pair and a share of the launch's iteration range.  Everything here is wave-uniform and kept
  // in scalar registers (readfirstlane); the loads that only need the pair index are issued together.
  const uint32_t unit = blockIdx.x * (uint32_t)kUnitsPerWg + (uint32_t)u;
  const bool in_grid = unit < n_pairs * (uint32_t)plan.n_shares;
  const uint32_t pair = in_grid ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(unit / (uint32_t)plan.n_shares)) : 0u;
  const int share = (int)(unit - pair * (uint32_t)plan.n_shares);
  const PairPrep* __restrict__ pp = plan.prep + pair;
  const uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
  // walk[pair].state >= 0: upper bound of the iterations the pair can still need; < 0: its loop has ended
  const WalkState ws = plan.walk[pair];
  const int batch_class1 = plan.walk[n_pairs].state;
  const int n_all_ld = pp->n_all;
  const float pmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pp->pmax)));
  const bool fast_alpha = __builtin_amdgcn_readfirstlane((int)pp->fast_alpha) != 0;
  uint64_t w_nonzero[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) w_nonzero[r] = uniform_u64(pp->w_nonzero[r]);
  // the first four words of the pair's viable-iteration mask from the range's first block on (256 iterations: the whole
  // range of the default 200); later blocks are fetched when the scan reaches them
  const int blk0 = (plan.phase_begin + share * plan.share_iters) >> 6;
  uint64_t vw0 = 0ull, vw1 = 0ull, vw2 = 0ull, vw3 = 0ull;
  if (blk0 + 0 < plan.vmask_words) vw0 = uniform_u64(vm_pair[blk0 + 0]);
  if (blk0 + 1 < plan.vmask_words) vw1 = uniform_u64(vm_pair[blk0 + 1]);
  if (blk0 + 2 < plan.vmask_words) vw2 = uniform_u64(vm_pair[blk0 + 2]);
  if (blk0 + 3 < plan.vmask_words) vw3 = uniform_u64(vm_pair[blk0 + 3]);
  bool have = in_grid;
  int k_begin = 0, k_end = 0;
  {
    const int pair_state = plan.phase_begin == 0 ? I : __builtin_amdgcn_readfirstlane(ws.state);
    // class 2 (no jump of `it` so far, junk-heavy; class 1 when the batch has few such pairs: effective_class): everything
    // that is left is recorded in this launch
    int cls = plan.phase_begin != 0 ? __builtin_amdgcn_readfirstlane(ws.speculate)
                                    : (plan.first_spec ? __builtin_amdgcn_readfirstlane((int)plan.preclass[pair]) : 0);
    if (cls == 1) cls = ((uint32_t)__builtin_amdgcn_readfirstlane(batch_class1) * 64u <= n_pairs) ? 2 : 0;
    const int end = min(cls == 2 ? plan.spec_end : plan.phase_end, pair_state);
    k_begin = plan.phase_begin + share * plan.share_iters;
    k_end = min(k_begin + plan.share_iters, end);
    have = have && pair_state >= 0 && k_begin < k_end;
  }
  const int n_all = have ? __builtin_amdgcn_readfirstlane(n_all_ld) : 0;
  SP_MARK(17)
  have = have && (n_all > rc.min_matches && n_all >= 4);  // no RANSAC for this pair (node.cpp:1087, :1130)

  // ---- prologue: the unit's match records (into registers first: the loads fly while the first iterations are picked)
  // ---- prologue: the unit's match records go global -> LDS directly (global_load_lds_dwordx4: 64 x 16 bytes per wave
  // instruction, LDS destination = wave-uniform base + lane * 16); the loads fly while the first iterations are picked.
  // (A unit without work reads its pair's records all the same: the loads do not wait for the pair's state.)
  if (in_grid) {
    const char* __restrict__ src = reinterpret_cast<const char*>(pp->M);
    for (int i = wu; i < (kMVec + kWave - 1) / kWave; i += kWavesPerUnit) {
      const int v = i * kWave + lane;
      if (v < kMVec)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)v * 16),
                                         (__attribute__((address_space(3))) void*)(lds.M[u] + i * (kWave * 4)), 16, 0, 0);
    }
  }
  if (lane < kSlots) { wl.slot[lane].active = 0; wl.slot[lane].iter = -1; }
  if (wave == 0) {
    lds.req[lane] = 0;
    if (lane == 0) lds.lock = 0;
  }
  const float* __restrict__ M = lds.M[u];
  uint32_t thr = (uint32_t)rc.min_matches;                                         // :1094
  if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
  thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr);
  IterRec* __restrict__ rec_pair = plan.recs + (size_t)pair * (size_t)I;
  IterSum* __restrict__ sum_pair = plan.sums + (size_t)pair * (size_t)I;

  // ---- the unit's viable iterations, in order; the k-th one belongs to wave k mod kWavesPerUnit
  int blk = (k_begin >> 6) - 1;
  uint64_t word = 0ull;
  uint32_t rank = 0u;
  auto next_item = [&]() -> int {
    for (;;) {
      if (word == 0ull) {
        ++blk;
        const int lo = blk << 6;
        if (lo >= k_end) return -1;
        const int rel = blk - blk0;
        uint64_t wv = rel == 0 ? vw0 : (rel == 1 ? vw1 : (rel == 2 ? vw2 : (rel == 3 ? vw3 : uniform_u64(vm_pair[blk]))));
        if (k_begin > lo) wv &= ~0ull << (k_begin - lo);
        if (k_end - lo < 64) wv &= (1ull << (k_end - lo)) - 1ull;
        word = wv;
        continue;
      }
      const int b = (int)__builtin_ctzll(word);
      word &= word - 1ull;
      const bool mine = (plan.debug_flags & 2) ? wu == 0 : (rank % (uint32_t)kWavesPerUnit) == (uint32_t)wu;
      ++rank;
      if (mine) return (blk << 6) + b;
    }
  };

  // iterations that have left their refinement loop: the outcome record, the slot is free again
  auto close_finished = [&]() {
    if (lane < kSlots) {
      SlotB& sl = wl.slot[lane];
      if (sl.iter >= 0 && !sl.active) {
        IterRec& r = rec_pair[sl.iter];
#pragma unroll
        for (int i = 0; i < 9; ++i) r.rR[i] = sl.rR[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) r.rt[i] = sl.rt[i];
#pragma unroll
        for (int q = 0; q < kRounds; ++q) r.rmask[q] = sl.rmask[q];
        r.rerr = sl.rerr;
        r.rn = sl.rn;
        r.pad = 0;
        sum_pair[sl.iter] = IterSum{sl.rerr, sl.rn, 0};
        sl.iter = -1;
      }
    }
    lsync();
    SP_MARK(1)
  };
  // free slots take the next viable iterations; their hypotheses are fetched together, 6 lanes (float2) each.
  // Returns whether any slot holds an iteration.
  auto refill = [&]() -> bool {
    bool occupied = false;
    int n_open = 0, my_g = -1, my_k = 0;
    for (int g = 0; g < kSlots; ++g) {
      int it_g = __builtin_amdgcn_readfirstlane(wl.slot[g].iter);
      if (it_g < 0) {
        const int k = next_item();
        if (k >= 0) {
          if (lane / 6 == n_open) { my_g = g; my_k = k; }
          ++n_open;
          it_g = k;
        }
      }
      occupied |= it_g >= 0;
    }
    SP_MARK(2)
    SP_COUNT(12, n_open)
    if (my_g >= 0) {
      int lane_here = lane;
      asm volatile("" : "+v"(lane_here));  // the load address is formed here, not carried through the rounds
      const int e2 = lane_here % 6;
      SlotB& sl = wl.slot[my_g];
      const float2 v = reinterpret_cast<const float2*>(rec_pair[my_k].rR)[e2];  // rR[9], rt[3] are contiguous
      reinterpret_cast<float2*>(sl.R)[e2] = v;                                   // ... and so are R[9], t[3]
      if (e2 == 0) {
        const float IR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int i = 0; i < 9; ++i) sl.rR[i] = IR[i];  // :1137 refined = Identity
#pragma unroll
        for (int i = 0; i < 3; ++i) sl.rt[i] = 0.f;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) sl.rmask[r] = 0ull;
        sl.rerr = 1e6;  // :1133
        sl.rn = 0;      // :1134
        sl.active = 1;
        sl.round = 0;
        sl.iter = my_k;
      }
    }
    return occupied;
  };

  lsync();
  SP_MARK(18)
  bool occupied = have ? refill() : false;
  SP_MARK(19)
  // the match records are in LDS behind the first hypotheses' loads; then the only workgroup barrier: from here on the
  // waves run on their own
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_MARK(20)
  __syncthreads();
  SP_MARK(0)

  for (; occupied; close_finished(), occupied = refill()) {
    lsync();
    SP_MARK(3)
    SP_COUNT(13, 1)

    // ================================ one pass of the refinement loop (:1140) for every active slot
    // ---- scorings (:1148), one after the other (lane = match), each followed by its error sum
    for (int g = 0; g < kSlots; ++g) {
      SlotB& sl = wl.slot[g];
      if (__builtin_amdgcn_readfirstlane(sl.active) == 0) continue;
      float curR[9], curt[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) curR[i] = sl.R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) curt[i] = sl.t[i];
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
    if (lane < kSlots) {
      SlotB& sl = wl.slot[lane];
      if (sl.active) {
        const int n_inl = sl.cn, rn = sl.rn;
        const uint32_t need = max(thr, (uint32_t)rn);
        // mean_error = 1e9 below 3 inliers (:1012-1014); a count below `need` is rejected by the count
        const double err_mine = !((uint32_t)n_inl < need || n_inl < 3) ? sqrt(sl.csum / (double)n_inl) : 1e9;  // :1016-1017
        float max_dist_f = rc.max_dist_m;
        asm volatile("" : "+v"(max_dist_f));  // (kept out of the loop-invariant registers: they are scarce)
        if (!((uint32_t)n_inl < thr || err_mine > (double)max_dist_f)) {  // :1154
          if (n_inl >= rn && err_mine <= sl.rerr) {               // :1160
            still = (n_inl != rn);                                // :1166
#pragma unroll
            for (int i = 0; i < 9; ++i) sl.rR[i] = sl.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) sl.rt[i] = sl.t[i];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) sl.rmask[r] = sl.cmask[r];
            sl.rn = n_inl;
            sl.rerr = err_mine;
          }
        }
        if (sl.round == 18) still = false;  // the 19th pass was the last one (:1140)
        sl.round++;
        sl.active = still ? 1 : 0;
      }
    }
    const bool any_active = __ballot(still) != 0ull;
    lsync();
    SP_MARK(5)
    if (!any_active) continue;
    // ---- refits (:1142): the weighted-mean recurrences of the wave's active slots side by side ...
    int n_mine = 0, k256_mine = 0, n_max = 0, n_min = RGBDFE_MAX_MATCHES;
    for (int g = 0; g < kSlots; ++g) {
      SlotB& sl = wl.slot[g];
      if (__builtin_amdgcn_readfirstlane(sl.active) == 0) continue;
      uint64_t m5[kRounds];
#pragma unroll
      for (int r = 0; r < kRounds; ++r) m5[r] = uniform_u64(sl.rmask[r]);
      int k256_g;
      const int n_g = fit_compact(g, m5, w_nonzero, wl.u.fit, k256_g);
      if (lane / 9 == g) { n_mine = n_g; k256_mine = k256_g; }
      n_max = max(n_max, n_g);
      n_min = min(n_min, n_g);
    }
    lsync();
    SP_MARK(6)
    {
      float C, m1, m2;
      if (fast_alpha) fit_recurrence<true>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2);
      else fit_recurrence<false>(n_mine, k256_mine, n_min, n_max, wl.u.fit, M, C, m1, m2);
      // lane 9s+x holds C[x], lane 9s+j mean1[j], lane 9s+3i mean2[i] of slot s: into the slot's mailbox
      int lane_here = lane;
      asm volatile("" : "+v"(lane_here));  // the mailbox address is formed here, not carried through the round
      const int s = min(lane_here / 9, kSlots - 1), x = lane_here % 9;
      if (lane_here < 9 * kSlots && wl.slot[s].active) {
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
    if (lane < kSlots && wl.slot[lane].active) flag_store(&lds.req[my_req], 1);
    lsync();
    for (;;) {
      const bool pending = lane < kSlots && flag_load(&lds.req[my_req]) != 0;
      if (__ballot(pending) == 0ull) break;
      int got = 0;
      if (lane == 0) got = atomicCAS(&lds.lock, 0, 1) == 0 ? 1 : 0;
      got = __builtin_amdgcn_readfirstlane(got);
      if (got) {
        SP_MARK(8)
        asm volatile("" ::: "memory");
        const bool p = lane < kSplitSlots && flag_load(&lds.req[lane]) == 1 && (!(plan.debug_flags & 1) || lane / kSlots == wave);
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
            SlotB& sl = lds.w[lane / kSlots].slot[lane % kSlots];
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
        if (lane == 0) atomicExch(&lds.lock, 0);
        SP_COUNT(15, 1)
        SP_MARK(9)
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
    }
    asm volatile("" ::: "memory");
    SP_MARK(8)
  }
#ifdef RGBDFE_PROFILE_PHASES
  SP_MARK(10)
  if (lane == 0) {
    const unsigned row = atomicAdd(&g_split_count, 1u) % kSplitLogWaves;
    for (int i = 0; i < 24; ++i) g_split_log[row][i] = sp[i];
    g_split_log[row][24] = 1ull;
    g_split_log[row][25] = have ? 1ull : 0ull;
  }
#endif
}

void launch_ransac_hyp(const PairWork* work, uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan,
                       hipStream_t stream) {
  if (n_pairs == 0 || rc.ransac_iterations <= 0) return;
  hipLaunchKernelGGL(ransac_hyp_kernel, dim3(n_pairs), dim3(kHypThreads), 0, stream, work, n_pairs, rc, plan);
}

void launch_ransac_refine(uint32_t n_pairs, const RansacConst& rc, const SplitPlan& plan, hipStream_t stream) {
  const uint32_t units = n_pairs * (uint32_t)plan.n_shares;
  if (units == 0) return;
  hipLaunchKernelGGL(ransac_refine_kernel, dim3((units + kUnitsPerWg - 1) / kUnitsPerWg), dim3(kSplitThreads), 0, stream,
                     n_pairs, rc, plan);
}

int ransac_split_words_per_pair(int ransac_iterations) {
  const int I = ransac_iterations > 0 ? ransac_iterations : 0;
  return 4 * ((I + kHypThreads - 1) / kHypThreads);  // a workgroup of the hypothesis kernel writes 4 words per pass
}
int ransac_split_waves_per_unit() { return kWavesPerUnit; }

}  // namespace rgbdfe

#ifdef RGBDFE_PROFILE_PHASES
// diagnostics build only (librgbdfe_prof.so): wall cycles per phase summed over the refinement kernel's waves
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
