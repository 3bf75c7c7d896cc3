// select_ransac.hip -- per-pair match selection + RANSAC rigid-transform estimation (gfx950).
//
// Replaces, per (newer node, older node) pair:
//   Node::featureMatching's `hd >= 128` gate + keepStrongestMatches     (node.cpp:572, 520-531, 674)
//   Node::getRelativeTransformationTo (RANSAC driver)                   (node.cpp:1074-1277)
//   sample_matches_prefer_by_distance                                   (node.cpp:1024-1047)
//   getTransformFromMatches / pcl::TransformationFromCorrespondences    (transformation_estimation_euclidean.cpp:7-61)
//   Node::computeInliersAndError + errorFunction2                       (node.cpp:968-1020, misc.cpp:697-770)
//   Node::matchNodePair's result assembly                               (node.cpp:1305-1429)
//
// Kernels (one wave64 per workgroup throughout: barriers are free, LDS is private to the wave):
//   pair_prep_kernel    once per pair.  Selection: counting sort of the (hd, queryIdx) keys in LDS -- histogram by
//                       hd, wave prefix scan, then a stable placement pass that ranks equal-hd lanes of a 64-query
//                       chunk with __ballot bit-matching (SIFT: the head of the list sift_sort_kernel ordered); the
//                       matched points become 7-word records (from.xyz, to.xyz, weight) in the pair's PairPrep block.
//   select_ransac_kernel<MODE>  the RANSAC work (12.9 KB LDS, 168 VGPRs: 12 waves per CU), see MODE below:
//     * hypothesis generation: LANE = RANSAC ITERATION.  64 iterations' 4-point samples, weighted fits and 3x3
//       Jacobi SVDs are computed at once (the counter-based generator makes iteration k's sample a pure function
//       of k, D1);
//     * slots: the refinement loops (node.cpp:1140-1169) of 7 iterations run side by side, round by round;
//     * scoring (per slot): LANE = MATCH, two passes -- a float prefilter of the reference's shortcut test with a
//       proven error band (exact double round when a lane is too close to call) + ballot compaction of the
//       candidates, then (LANE = CANDIDATE) the double-precision covariance and Cholesky solve with the divisions /
//       square roots spelled out as their unscaled gfx950 expansions behind an exponent guard;
//     * error sums and the loop's bookkeeping (per round): LANE = SLOT, the sequential sums of the round's scorings
//       side by side from the wave's rows of the error pool;
//     * refits (per round): the PCL weighted-mean recurrences of all active slots share one 63-lane loop (9 state
//       elements per slot), followed by one batched SVD (LANE = SLOT).
//   replay_walk_kernel  the reference's in-order bookkeeping (`it += 10/20`, the 80 % exit) over recorded iterations.
//   The float/double operation order is the oracle's (oracle/rgbd_oracle.c), which restates the reference and is
//   pinned on the reference's own compiled code (oracle/_ref/libref_ransac.so): sequential weighted-mean recurrence,
//   sequential error sum.  Compiled with -ffp-contract=off the results are bit-identical to the CPU restatement, so
//   every discrete RANSAC decision is too.
#include "ransac_device.h"

namespace rgbdfe {

// Every kernel of this file runs ONE wave per workgroup: a barrier orders the wave's own LDS traffic (lanes exchange data
// through LDS).  __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0) before s_barrier); an LDS-only
// variant (RGBDFE_LDS_ONLY_SYNC) was measured and changes nothing -- a wave's wall time is set by the SIMD's issue slots
// it shares with two other waves, not by its own waits -- so the plain barrier stays.
#ifdef RGBDFE_LDS_ONLY_SYNC
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_sync_global() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
#else
__device__ __forceinline__ void wave_sync() { __syncthreads(); }
__device__ __forceinline__ void wave_sync_global() { __syncthreads(); }
#endif


// selection phase
struct SelBuf {
  uint32_t mqt[RGBDFE_MAX_MATCHES];  // queryIdx | trainIdx << 16
  uint32_t mhd[RGBDFE_MAX_MATCHES];
  uint32_t cnt[128];                 // hd histogram / running bin cursors
};
// scoring phase: candidates (matches that survive the cheap shortcut test) and inliers, compacted
struct ScoreBuf {
  uint16_t cand[RGBDFE_MAX_MATCHES]; // k-th candidate match
  uint32_t mbits[2 * kRounds];       // inlier set as bits over matches
};
// error sums: one block of every slot's error list on its way from the error pool (global) to lane = slot
constexpr int kStageBlk = 32;                // list entries per row and block
constexpr int kStageRow = kStageBlk + 2;     // doubles; the rows start 4 banks apart: conflict-free 16-byte reads
struct SumStage {
  double v[kSlots][kStageRow];
};
union Scratch {  // the phases never overlap
  ScoreBuf sc;
  FitBuf fit;
  SumStage sum;
};
// a (transform, inlier set, error) triple kept in LDS (wave-uniform state)
struct Hyp {
  float R[9], t[3];
  uint64_t mask[kRounds];
  int n;
  int nan;
  double err;
};
// One RANSAC iteration in flight.  The iterations of a window are independent (the sample of
// iteration k is a pure function of k, D1), so their refinement loops run side by side: scoring is
// per slot (lane = match), the refits' 3x3 SVDs are batched (lane = slot).
struct Slot {
  float R[9], t[3];          // transform to score next
  float rR[9], rt[3];        // refined_transformation (node.cpp:1137,1163)
  // refined_matches; while the slot is active this is also the inlier set of the last scoring, i.e. the
  // input of the next refit (a slot stays active only through an accepted scoring, :1160-1166)
  uint64_t rmask[kRounds];
  uint64_t cmask[kRounds];   // inlier set of the slot's scoring in the current refinement round
  double rerr;               // refined_error
  int rn;                    // refined_matches.size()
  int cn;                    // inliers of the current scoring
  int active;                // still inside the refinement loop (:1140-1169)
  int round;                 // refinement passes done
  int iter;                  // RANSAC iteration held by the slot, -1 = free
  int pad;
};
struct __attribute__((aligned(16))) RansacLds {
  Scratch u;
  // match m: M[7m..7m+2] newer node's point ("from"), M[7m+3..7m+5] older node's point ("to"),
  // M[7m+6] = 1/(from.z*to.z) (transformation_estimation_euclidean.cpp:25)
  alignas(16) float M[RGBDFE_MAX_MATCHES * kRec];
  Slot slot[kSlots];
  Hyp best;
};



// ---------------------------------------------------------------------------------
// computeInliersAndError (node.cpp:968-1020) with errorFunction2 (misc.cpp:697-770).
// R,t are wave-uniform.  Returns inlier masks, count and rms error.
//   pass 1 (lane = match, 5 rounds): the hypothesis-dependent part that is cheap -- transform the point,
//           squared distance, the reference's shortcut test (misc.cpp:726-735) -- and a ballot compaction
//           of the surviving candidates;
//   pass 2 (lane = candidate): the double-precision covariance + 3x3 Cholesky solve, only for the
//           ceil(n_cand / 64) dense rounds that are needed; inliers are compacted again, in match order;
//   sum   : the reference's strictly sequential error sum over the compacted inliers.
// When fewer candidates than `need` survive pass 1 the caller will reject the hypothesis whatever
// the exact numbers are (node.cpp:1154, :1206): the solve is skipped and (0, 1e9) is returned.
// Optional phase timers (librgbdfe_prof.so, -DRGBDFE_PROFILE_PHASES): wall cycles per phase are
// written over the head of all_q of each result.  Never enabled in the product build.
#ifdef RGBDFE_PROFILE_PHASES
// totals over the recording waves of all launches since the last reset (slot 16 = waves); rgbdfe_debug_phase_totals
__device__ unsigned long long g_phase_totals[24];
#define PH_DECL uint64_t ph_t0 = __builtin_readcyclecounter(); uint64_t ph[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(i) { uint64_t ph_t1 = __builtin_readcyclecounter(); ph[i] += ph_t1 - ph_t0; ph_t0 = ph_t1; }
#define PH_COUNT(i) { ph[i] += 1; }
#define PH_ADDX(i, v) { ph[i] += (uint64_t)(v); }
#define PH_ARG , uint64_t* ph, uint64_t& ph_t0
#define PH_PASS , ph, ph_t0
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_COUNT(i)
#define PH_ADDX(i, v)
#define PH_ARG
#define PH_PASS
#endif
#define PH_ADD(i, v) PH_ADDX(i, v)


// ---------------------------------------------------------------------------------
// The squared errors of a scoring's inliers go, in match order, to a row of the wave's region of the error pool
// (global memory: 7 rows of kEcRow doubles, L2 resident while the wave lives) instead of LDS: the strictly
// sequential error sums of the (up to 7) scorings of a refinement round then run side by side, lane = slot
// (sum_rows), instead of one after the other with all 64 lanes repeating the same addition.
constexpr int kEcRow = RGBDFE_MAX_MATCHES;      // one list (whole staging blocks)
constexpr int kEcRegion = kSlots * kEcRow;      // doubles per wave

// Passes 1 and 2 of computeInliersAndError: the inlier set (mask, n_inl) and the inliers' errors (ec_row).
// n_inl == 0 when fewer than `need` candidates survive pass 1 (the caller rejects the hypothesis whatever the
// exact numbers are).
// (Summing the errors on the spot for short candidate lists -- a scalar v_readlane / v_add_f64 chain over the inlier
// lanes -- was measured: slower on every workload, the chains of a round add up instead of running lane = slot.)
__device__ __forceinline__ void score_passes(const float* R, const float* tr, int n_all, uint32_t need,
                                             const RansacConst& rc, RansacLds& lds, float pmax,
                                             double* __restrict__ ec_row, uint64_t* mask, int& n_inl PH_ARG) {
  const int lane = threadIdx.x;
  double Rd[9], td[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rd[i] = (double)R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) td[i] = (double)tr[i];
  const double rcx = rc.raster_cov_x, rcy = rc.raster_cov_y, dc = rc.depth_cov;
  const double smax = rcx > dc ? rcx : dc;
  const double shortcut = 2.0 * (smax + smax);
  ScoreBuf& sb = lds.u.sc;
  // ---- pass 1.  The shortcut test `dsq > shortcut` (misc.cpp:731) is decided from a float evaluation of dsq
  // whenever that is safe: with u = 2^-24 and P = the largest |coordinate| of the matched points, the fma-evaluated
  // d_i differs from the exact one by at most e_i = 4u ((|R_i0| + |R_i1| + |R_i2| + 1) P + |t_i|), hence the float
  // dsq from the exact one by at most 2 sqrt(S) (e_0 + e_1 + e_2) + (e_0 + e_1 + e_2)^2 + 4u S near the threshold S
  // (and a dsq far above S stays above S - E).  E is doubled for the roundings of the bound itself and of the
  // reference's double evaluation.  Lanes inside [S - E, S + E] -- or with NaN / overflow, where every comparison
  // fails -- make the wave redo the round in double, the reference's own arithmetic.
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
    // stride-7 word addresses: conflict-free across the LDS banks
    const float pxf = lds.M[m * kRec + 0], pyf = lds.M[m * kRec + 1], pzf = lds.M[m * kRec + 2];
    const float qxf = lds.M[m * kRec + 3], qyf = lds.M[m * kRec + 4], qzf = lds.M[m * kRec + 5];
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
  wave_sync();
  n_inl = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) mask[r] = 0ull;
  PH_ADD(13, n_cand)
  if ((uint32_t)n_cand < need) { PH_COUNT(15) return; }  // hopeless: nobody looks at the exact numbers
  // ---- pass 2
  for (int k0 = 0; k0 < n_cand; k0 += kWave) {
    const int k = k0 + lane;
    const bool act = k < n_cand;
    const int m = act ? (int)sb.cand[k] : 0;
    const double a0 = (double)lds.M[m * kRec + 0], a1 = (double)lds.M[m * kRec + 1], a2 = (double)lds.M[m * kRec + 2];
    const double b0 = (double)lds.M[m * kRec + 3], b1 = (double)lds.M[m * kRec + 4], b2 = (double)lds.M[m * kRec + 5];
    double d[3];
    // mu_1_in_frame_2 = (T * x1).head<3>() with x1.w == 1 (misc.cpp:724)
    d[0] = (((Rd[0] * a0 + Rd[1] * a1) + Rd[2] * a2) + td[0]) - b0;
    d[1] = (((Rd[3] * a0 + Rd[4] * a1) + Rd[5] * a2) + td[1]) - b1;
    d[2] = (((Rd[6] * a0 + Rd[7] * a1) + Rd[8] * a2) + td[2]) - b2;
    double e = DBL_MAX;
    {
      const double c1[3] = {rcx * a2, rcy * a2, dc};
      const double c2[3] = {rcx * b2, rcy * b2, dc};
      // S = R^T * cov1 * R + cov2 (misc.cpp:751,760), lower triangle only.
      // S(i,j) = (R(0,i)c1_0*R(0,j) + R(1,i)c1_1*R(1,j)) + R(2,i)c1_2*R(2,j)
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
      ec_row[n_inl + (int)lane_rank(im)] = e;  // candidates ascend in match index: so do the inliers
      atomicOr(&sb.mbits[m >> 5], 1u << (m & 31));
    }
    n_inl += __popcll(im);
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(sb.mbits[2 * r]);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(sb.mbits[2 * r + 1]);
    mask[r] = ((uint64_t)hi << 32) | lo;
  }
  PH_ADD(14, n_inl)
  wave_sync();
}

// mean_error += mahal_dist in match order (node.cpp:1006): a strictly sequential double sum per list.  Lane s
// (< kSlots) sums the first n_mine entries of row s of the wave's error-pool region; every lane walks n_max
// entries.  The additions are one dependent chain per lane, so the cost of a round's (up to 7) sums is that of the
// longest one.  Transport is 64 lanes wide: a block of 32 entries of all 7 rows is fetched with 4 coalesced loads
// per lane (two blocks ahead of the chain: the rows sit in L2), entries behind a list's end are replaced by 0.0
// (x + 0.0 == x for these non-negative sums), the block goes through LDS, and lane s reads row s back.
__device__ __forceinline__ double sum_rows(const double* __restrict__ ec_region, RansacLds& lds, int lane, int n_mine,
                                           int n_max PH_ARG) {
  SumStage& st = lds.u.sum;
  // transport role of this lane: entries q = j * 64 + lane of a block, row = q / 32, column = q % 32
  const int col = lane & (kStageBlk - 1);
  int row[4], n_row[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    row[j] = (j * kWave + lane) / kStageBlk;                  // 0 .. 7; row 7 does not exist
    n_row[j] = __shfl(n_mine, min(row[j], kSlots - 1));
    if (row[j] >= kSlots) n_row[j] = -1;                      // never stored
  }
  const int nb = (n_max + kStageBlk - 1) / kStageBlk;
  auto fetch = [&](double* g, int b) {
    const int idx = max(min(b, nb - 1), 0) * kStageBlk + col;  // reading ahead of the last block re-reads it
#pragma unroll
    // only entries that exist are loaded: the rest of a region is memory nobody has written in this launch (other
    // slots' rows, the tail of a row) -- touching it costs HBM and TLB misses on cold lines, measured at ~7 us per round
    for (int j = 0; j < 4; ++j) {
      g[j] = 0.0;
      // the element offset is formed HERE (the asm keeps the compiler from hoisting four 64-bit lane addresses to the top
      // of the kernel, where they were spilled: their reloads from scratch, each followed by s_waitcnt vmcnt(0), made the
      // four loads of a block four dependent memory round trips -- ~9000 cycles per refinement round)
      uint32_t off = (uint32_t)(((j * kWave + lane) / kStageBlk) * kEcRow + idx);
      asm volatile("" : "+v"(off));
      if (idx < n_row[j]) g[j] = ec_region[off];
    }
  };
  const double* mine = st.v[min(lane, kSlots - 1)];
  double sum = 0.0;
  auto block = [&](double* g, int b) {
    const int idx = b * kStageBlk + col;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n_row[j] >= 0) st.v[row[j]][col] = idx < n_row[j] ? g[j] : 0.0;
    wave_sync();
    asm volatile("" ::: "memory");  // g is dead: its registers take the block after next
    fetch(g, b + 2);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < kStageBlk; i += 2) {
      const double2 v = *reinterpret_cast<const double2*>(mine + i);
      sum += v.x;
      sum += v.y;
    }
    asm volatile("" : "+v"(sum) :: "memory");
    wave_sync();
  };
  double g0[4], g1[4];
  PH_MARK(20)
  wave_sync_global();  // the rows were written by this wave's own stores (score_passes, pass 2)
  PH_MARK(21)
  fetch(g0, 0);
  fetch(g1, 1);
#ifdef RGBDFE_PROFILE_PHASES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the read-back latency on its own
  PH_MARK(0)
#endif
  for (int b = 0; b < nb; b += 2) {
    block(g0, b);
    if (b + 1 < nb) block(g1, b + 1);
  }
  return sum;
}


// computeInliersAndError for ONE wave-uniform transform (the identity fallback): score_passes + lane 0's sum.
__device__ __forceinline__ void score_hypothesis(const float* R, const float* tr, int n_all, uint32_t need,
                                                 const RansacConst& rc, RansacLds& lds, float pmax,
                                                 double* __restrict__ ec_region, uint64_t* mask, int& n_inl,
                                                 double& err PH_ARG) {
  score_passes(R, tr, n_all, need, rc, lds, pmax, ec_region, mask, n_inl PH_PASS);
  err = 1e9;
  if ((uint32_t)n_inl < need || n_inl < 3) return;  // err stays 1e9 (node.cpp:1012-1014); rejected anyway when < need
  const double sum = uniform_f64(sum_rows(ec_region, lds, threadIdx.x, threadIdx.x == 0 ? n_inl : 0, n_inl PH_PASS));
  err = sqrt(sum / (double)n_inl);  // node.cpp:1016-1017
}



__device__ __forceinline__ void hyp_store(Hyp& h, const float* R, const float* t, const uint64_t* mask,
                                          int n, int nan, double err) {
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) h.R[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) h.t[i] = t[i];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) h.mask[r] = mask[r];
    h.n = n;
    h.nan = nan;
    h.err = err;
  }
}


// Match sources: ORB = packed (hd, train row) keys of hamming_nn_kernel; SIFT = the mutual-best
// match list of sift_finish_kernel (queryIdx, trainIdx, L2 distance).
struct SiftMatchList {
  const uint16_t* q;   // [pair][max_kp]
  const uint16_t* t;
  const float* d;
  const int32_t* n;    // [pair]
  float* all_dist;     // [pair][RGBDFE_MAX_MATCHES] output: distances of the selected matches
};

// keepStrongestMatches by the float L2 distance (node.cpp:674) with D2's tie-break, once per pair: the rank of a match
// is the number of matches with a smaller (distance, queryIdx) key (distances are non-negative floats: their bit
// patterns order like the values).  The max_matches best matches are written back, in rank order, over the head of the
// pair's list; the count stays the original one.  One wave per pair.
constexpr int kSortThreads = 256;
__global__ __launch_bounds__(kSortThreads) void sift_sort_kernel(uint16_t* __restrict__ sm_q, uint16_t* __restrict__ sm_t,
                                                                 float* __restrict__ sm_d, const int32_t* __restrict__ sm_n,
                                                                 uint32_t max_kp, uint32_t n_pairs, int max_matches) {
  constexpr int kTile = 2048;  // keys held in LDS at a time
  __shared__ uint64_t s_key[kTile];
  __shared__ uint32_t s_qt[RGBDFE_MAX_MATCHES], s_d[RGBDFE_MAX_MATCHES];
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int tid = threadIdx.x;
  uint16_t* __restrict__ sq = sm_q + (size_t)pair * max_kp;
  uint16_t* __restrict__ st = sm_t + (size_t)pair * max_kp;
  float* __restrict__ sd = sm_d + (size_t)pair * max_kp;
  const int n = sm_n[pair];
  const int n_all = min(n, max_matches);
  if (n <= kTile) {
    // Bitonic sort of (distance, queryIdx, trainIdx) keys -- queryIdx is unique inside a list, so the order is the total
    // order (distance, queryIdx) of the rank count below and the train index rides along in the low bits.
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int j = tid; j < n_pad; j += kSortThreads)
      s_key[j] = j < n ? (((uint64_t)__float_as_uint(sd[j]) << 32) | ((uint64_t)sq[j] << 16) | (uint64_t)st[j]) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= n_pad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < (n_pad >> 1); i += kSortThreads) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
          const uint64_t a = s_key[lo], b = s_key[hi];
          const bool up = (lo & k) == 0;
          if ((a > b) == up) {
            s_key[lo] = b;
            s_key[hi] = a;
          }
        }
        __syncthreads();
      }
    for (int m = tid; m < n_all; m += kSortThreads) {
      const uint64_t key = s_key[m];
      sq[m] = (uint16_t)((key >> 16) & 0xFFFFu);
      st[m] = (uint16_t)(key & 0xFFFFu);
      sd[m] = __uint_as_float((uint32_t)(key >> 32));
    }
    return;
  }
  // longer lists (nodes above 2048 keypoints): rank count, the keys staged again for every 256 ranks
  auto key_of = [&](int i) { return ((uint64_t)__float_as_uint(sd[i]) << 16) | (uint64_t)sq[i]; };  // (distance, queryIdx)
  for (int base = 0; base < n; base += kSortThreads) {
    const int i = base + tid;
    const bool act = i < n;
    const uint64_t ki = act ? key_of(i) : 0ull;
    int rank = 0;
    for (int t0 = 0; t0 < n; t0 += kTile) {
      const int tn = min(kTile, n - t0);
      __syncthreads();
      for (int j = tid; j < tn; j += kSortThreads) s_key[j] = key_of(t0 + j);
      __syncthreads();
      int j = 0;
      for (; j + 8 <= tn; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += s_key[j + u] < ki ? 1 : 0;
      }
      for (; j < tn; ++j) rank += s_key[j] < ki ? 1 : 0;
    }
    if (act && rank < max_matches) {
      s_qt[rank] = (uint32_t)sq[i] | ((uint32_t)st[i] << 16);
      s_d[rank] = __float_as_uint(sd[i]);
    }
  }
  __syncthreads();  // every read of the unsorted list is done
  for (int m = tid; m < n_all; m += kSortThreads) {
    sq[m] = (uint16_t)(s_qt[m] & 0xFFFFu);
    st[m] = (uint16_t)(s_qt[m] >> 16);
    sd[m] = __uint_as_float(s_d[m]);
  }
}

// MODE selects what a wave does with a pair's RANSAC iterations ("record / replay"):
//   kWhole   the whole pair: windows of iterations refined side by side, replayed in order (one wave per pair)
//   kRecord  a share of the iterations of a phase: refine them (7 slots, refilled as iterations finish) and write each
//            iteration's outcome (IterRec) to memory -- several waves share one pair; no bookkeeping, no result
//   kReplay  the result of a pair whose in-order bookkeeping replay_walk_kernel has run over the records: adopt the best
//            record, the identity fallback, the result POD
// kRecord + replay_walk_kernel + kReplay give the same result as kWhole (an iteration's refinement is a pure function of
// its index, D1) with the refinement work of one pair spread over many waves.
constexpr int kWhole = 0, kRecord = 1, kReplay = 2;

// ---------------------------------------------------------------------------------
// Once per pair, before any select+RANSAC wave: the <= max_matches strongest matches in the reference's order
// (keepStrongestMatches + sort, node.cpp:674-676 / :1315) and their 3-D points as the 7-word records the RANSAC
// waves keep in LDS.  Writes the pair's PairPrep and the match lists of its result POD.  One wave per pair.
// ---------------------------------------------------------------------------------
template <bool SIFT>
__global__ __launch_bounds__(kWave) void pair_prep_kernel(
    const float4* __restrict__ xyz_pool, const PairWork* __restrict__ work,
    const uint32_t* __restrict__ keys, uint32_t key_planes, const SiftMatchList sm,
    rgbdfe_match_result* __restrict__ results, uint32_t max_kp, uint32_t n_pairs,
    const RansacConst rc, PairPrep* __restrict__ prep, uint32_t* __restrict__ zero64) {
  __shared__ SelBuf sel;
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x;
  // (the batch's order-bucket counters, SplitPlan::order_cnt: the hypothesis kernel's workgroups add to them)
  if (pair == 0 && zero64 != nullptr) zero64[lane] = 0u;
  const PairWork w = work[pair];
  rgbdfe_match_result* __restrict__ out = results + pair;
  PairPrep* __restrict__ pp = prep + pair;
  const int max_matches = rc.max_matches;
  int n_all;

  if (!SIFT) {
  const uint32_t nq = w.nq;
  const uint32_t* __restrict__ kin = keys + (size_t)pair * key_planes * max_kp;
  // the Hamming kernel may have split the train rows over `key_planes` blocks: min over the planes
  auto load_key = [&](uint32_t i) {
    uint32_t k = kin[i];
    for (uint32_t pl = 1; pl < key_planes; ++pl) k = min(k, kin[(size_t)pl * max_kp + i]);
    return k;
  };
  // ------------------------------------------------------------------ selection
  sel.cnt[lane] = 0;
  sel.cnt[lane + 64] = 0;
  wave_sync();
  for (uint32_t base = 0; base < nq; base += kWave) {
    const uint32_t i = base + lane;
    if (i < nq) {
      const uint32_t hd = load_key(i) >> 16;
      if (hd < 128u) atomicAdd(&sel.cnt[hd], 1u);  // node.cpp:572
    }
  }
  wave_sync();
  uint32_t total;
  uint32_t cut_hd;  // bins >= cut_hd start at or beyond max_matches: never selected
  {
    const uint32_t a = sel.cnt[2 * lane], b = sel.cnt[2 * lane + 1];
    uint32_t incl = a + b;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      uint32_t v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    const uint32_t excl = incl - (a + b);
    total = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t s0 = excl, s1 = excl + a;
    uint32_t c = 128u;
    if (s1 >= (uint32_t)max_matches) c = 2 * lane + 1;
    if (s0 >= (uint32_t)max_matches) c = 2 * lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c = min(c, (uint32_t)__shfl_xor(c, off));
    cut_hd = c;
    wave_sync();
    sel.cnt[2 * lane] = s0;
    sel.cnt[2 * lane + 1] = s1;
  }
  wave_sync();
  n_all = (int)min(total, (uint32_t)max_matches);
  // stable placement: (hd, queryIdx) order == D2's deterministic tie-break
  for (uint32_t base = 0; base < nq; base += kWave) {
    const uint32_t i = base + lane;
    uint32_t key = i < nq ? load_key(i) : 0xFFFFFFFFu;
    const uint32_t hd = key >> 16;
    const bool valid = hd < cut_hd;
    uint64_t same = __ballot(valid);
    if (same == 0ull) continue;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
      const bool bit = (hd >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const uint32_t rank = lane_rank(same);
    const uint32_t cnt_same = __popcll(same);
    uint32_t pos = 0;
    if (valid) pos = sel.cnt[hd] + rank;
    wave_sync();
    if (valid && rank == 0) sel.cnt[hd] = pos + cnt_same;
    if (valid && pos < (uint32_t)max_matches) {
      sel.mqt[pos] = i | ((key & 0xFFFFu) << 16);
      sel.mhd[pos] = hd;
    }
    wave_sync();
  }
  wave_sync();
  } else {
    // ------------------------------------------------------------- selection (SIFT)
    // sift_sort_kernel has left the pair's matches in keepStrongestMatches order (node.cpp:674, D2) at the head of
    // the list
    const uint16_t* __restrict__ sq = sm.q + (size_t)pair * max_kp;
    const uint16_t* __restrict__ st = sm.t + (size_t)pair * max_kp;
    const float* __restrict__ sd = sm.d + (size_t)pair * max_kp;
    n_all = min(sm.n[pair], max_matches);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int m = r * kWave + lane;
      if (m < n_all) {
        sel.mqt[m] = (uint32_t)sq[m] | ((uint32_t)st[m] << 16);
        sel.mhd[m] = __float_as_uint(sd[m]);  // distance bits travel in the hd slot
      }
    }
    wave_sync();
  }

  // ------------------------------------------------- matched 3-D points -> records
  const float4* __restrict__ qxyz = xyz_pool + (size_t)w.q_slot * max_kp;
  const float4* __restrict__ txyz = xyz_pool + (size_t)w.t_slot * max_kp;
  bool w_plain = false;  // a weight outside the window of fit_recurrence<true>
  float pmax = 0.0f;     // largest finite |coordinate| of the matched points (bounds the float prefilter's error)
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int m = r * kWave + lane;
    float4 p = make_float4(0.f, 0.f, 0.f, 1.f), q = make_float4(0.f, 0.f, 0.f, 1.f);
    uint32_t qt = 0, hd = 0;
    if (m < n_all) {
      qt = sel.mqt[m];
      hd = sel.mhd[m];
      p = qxyz[qt & 0xFFFFu];
      q = txyz[qt >> 16];
    }
    pmax = fmaxf(pmax, fmaxf(fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)),
                             fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fabsf(q.z))));  // fmaxf skips NaN
    float* __restrict__ rec = pp->M + m * kRec;
    rec[0] = p.x; rec[1] = p.y; rec[2] = p.z;
    rec[3] = q.x; rec[4] = q.y; rec[5] = q.z;
    // weight = 1.0/(from(2)*to(2)) (transformation_estimation_euclidean.cpp:25): the double
    // divide rounded to float equals the float divide (53 >= 2*24+2)
    const float wgt = 1.0f / (p.z * q.z);
    rec[6] = wgt;
    {
      const uint32_t ex = (__float_as_uint(wgt) >> 23) & 0xFFu;
      w_plain |= (m < n_all) && (wgt != 0.0f) && (ex < 127u - 40u || ex > 127u + 40u);
      const uint64_t nz = __ballot(wgt != 0.0f);
      if (lane == 0) pp->w_nonzero[r] = nz;
    }
    out->all_q[m] = (uint16_t)(qt & 0xFFFFu);
    out->all_t[m] = (uint16_t)(qt >> 16);
    if (SIFT) {
      out->all_hd[m] = 0;
      sm.all_dist[(size_t)pair * RGBDFE_MAX_MATCHES + m] = __uint_as_float(hd);
    } else {
      out->all_hd[m] = (uint8_t)hd;
    }
  }
  const bool fast_alpha = (__ballot(w_plain) == 0ull);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, d));
  if (lane == 0) {
    pp->n_all = n_all;
    pp->pmax = pmax;
    pp->fast_alpha = fast_alpha ? 1u : 0u;
    pp->pad = 0u;
  }
}

#ifdef RGBDFE_PADNOPS  // diagnostics: shifts the code that follows
__global__ void rgbdfe_pad_kernel(int* p) {
#pragma unroll
  for (int i = 0; i < RGBDFE_PADNOPS; ++i) asm volatile("s_nop 0");
  if (p) *p = 1;
}
#endif



template <int MODE>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 3))) void select_ransac_kernel(
    const PairWork* __restrict__ work, rgbdfe_match_result* __restrict__ results, uint32_t n_pairs,
    const RansacConst rc, const RecordPlan plan) {
  __shared__ RansacLds lds;
  // Recording waves: the workgroups of a launch go round-robin over the 8 XCDs, each with its own L2.  The grid is
  // walked in 8 contiguous segments, segment = blockIdx % 8, and a segment owns a contiguous range of pairs, so that the
  // waves sharing a pair (its PairPrep block, its records) run on one XCD.  The long shares of sub-grid B are ordered
  // share-major inside a segment (share 0 of all its pairs, then share 1, ...): consecutive workgroups -- which the
  // dispatcher spreads over the CUs -- then do the same kind of work instead of clustering a pair's long waves on a CU.
  // A recording launch has two sub-grids per pair: n_chunks shares of chunk_iters iterations (sub-grid A) and n_chunks_b
  // shares of a whole hypothesis batch (sub-grid B; only the second phase of a phased plan has one).  Which sub-grid
  // works for a pair, and how far, is the pair's class:
  //   WalkState::speculate 0  the first phase has moved `it` ahead (a hypothesis with > 50 % inliers): the loop may
  //                           end early -> sub-grid A up to the phase's nominal end, phase by phase;
  //                           (also: most iterations give a refined hypothesis, more than 9/14 of them -- such a pair
  //                           finds its > 50 % hypothesis soon: recording ahead of the walk would be wasted)
  //                        2  no jump so far and at most 9/14 of the iterations gave a refined hypothesis: the pair will
  //                           most likely run all its iterations -> sub-grid B records ALL that is left at once, in long
  //                           shares and with the hypothesis pre-screen (no further phases, no launch tails for it).
  //                        (a class "record all that is left, in short shares and without the pre-screen" was measured
  //                           and is slower than class 0 for the pairs it would apply to)
  const uint32_t per_pair = plan.n_chunks + plan.n_chunks_b;
  const uint32_t pps = (n_pairs + 7u) / 8u;  // pairs per segment
  const uint32_t seg = blockIdx.x % 8u, in_seg = blockIdx.x / 8u;
  // sub-grid A first, pair-major (a pair's short shares next to each other: its PairPrep block is read once into L2 and
  // the ordinary phases keep the locality they were tuned with); then sub-grid B, share-major
  const uint32_t units_a = pps * plan.n_chunks;
  const bool in_b = MODE == kRecord && in_seg >= units_a;
  const uint32_t pair = MODE != kRecord ? blockIdx.x
                        : seg * pps + (in_b ? (in_seg - units_a) % pps : (plan.n_chunks ? in_seg / plan.n_chunks : 0u));
  const uint32_t in_pair = MODE != kRecord ? 0u
                           : (in_b ? plan.n_chunks + (in_seg - units_a) / pps : (plan.n_chunks ? in_seg % plan.n_chunks : 0u));
  if (MODE == kRecord && (in_pair >= per_pair || pair >= seg * pps + pps)) return;
  const bool sub_b = MODE == kRecord && in_pair >= plan.n_chunks;
  const uint32_t unit = sub_b ? in_pair - plan.n_chunks : in_pair;  // share index inside the pair's sub-grid
  if (pair >= n_pairs) return;
  // record / replay bookkeeping: walk[pair].state >= 0 is an upper bound of the iterations the pair can still need,
  // < 0 means its loop has ended
  const int pair_state = MODE != kRecord ? 0 : (plan.phase_begin == 0 ? rc.ransac_iterations : plan.walk[pair].state);
  if (MODE == kRecord && pair_state < 0) return;
  const int pair_class = (MODE == kRecord && plan.phase_begin != 0) ? effective_class(plan.walk, pair, n_pairs) : 0;
  if (MODE == kRecord && (pair_class == 2) != sub_b) return;  // the other sub-grid works for this pair
  const int my_chunk_iters = sub_b ? plan.chunk_iters_b : plan.chunk_iters;
  const int recorded_end = MODE == kRecord ? min(sub_b ? plan.spec_end : plan.phase_end, pair_state) : 0;
  // the pre-screen pays where many hypotheses are junk: for the pairs the first phase's walk has put into class 2 (the
  // first phase itself, 14 of 200 iterations, runs without it: a pair of mostly valid hypotheses would only pay for it)
  const bool prescreen = MODE != kRecord || pair_class == 2 || plan.n_phases_total == 1;
  const int k_begin = MODE == kRecord ? plan.phase_begin + (int)unit * my_chunk_iters : 0;
  const int k_end = MODE == kRecord ? min(k_begin + my_chunk_iters, recorded_end) : 0;
  if (MODE == kRecord && k_begin >= k_end) return;  // nothing of this chunk is needed (any more)
  IterRec* __restrict__ rec_pair = MODE == kWhole ? nullptr : plan.recs + (size_t)pair * (size_t)rc.ransac_iterations;
  IterSum* __restrict__ sum_pair = MODE == kWhole ? nullptr : plan.sums + (size_t)pair * (size_t)rc.ransac_iterations;
  const int lane = threadIdx.x;
  const PairWork w = work[pair];
  rgbdfe_match_result* __restrict__ out = results + pair;
  double* __restrict__ ec_region = plan.ec_pool + (size_t)blockIdx.x * kEcRegion;  // this wave's rows of the error pool
  PH_DECL

  // ------------------------------------------------- the pair's matches (pair_prep_kernel) -> LDS
  const PairPrep* __restrict__ pp = plan.prep + pair;
  const int n_all = pp->n_all;
  const float pmax = pp->pmax;
  const bool fast_alpha = pp->fast_alpha != 0u;
  uint64_t w_nonzero[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) w_nonzero[r] = pp->w_nonzero[r];
  auto load_points = [&]() {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pp->M);
    float4* __restrict__ dst = reinterpret_cast<float4*>(lds.M);
    constexpr int kVec = RGBDFE_MAX_MATCHES * kRec / 4;
#pragma unroll
    for (int i = 0; i < (kVec + kWave - 1) / kWave; ++i) {
      const int v = i * kWave + lane;
      if (v < kVec) dst[v] = src[v];
    }
    wave_sync();
  };
  // no RANSAC for this pair (node.cpp:1087, :1130): a recording wave has nothing to record
  if (MODE == kRecord && !(n_all > rc.min_matches && n_all >= 4)) return;
  if (MODE != kReplay) load_points();  // a result wave needs them for the identity fallback only

  PH_MARK(1)
  // ------------------------------------------------------------------ RANSAC
  const float IR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, It[3] = {0, 0, 0};
  const uint64_t zero_mask[kRounds] = {0, 0, 0, 0, 0};
  hyp_store(lds.best, IR, It, zero_mask, 0, 0, 0.0);
  int best_n = 0;
  float rmse = 0.0f;  // MatchingResult() default (matching_result.h:27)
  int valid_iterations = 0, real_iterations = 0;
  bool found = false;

  // matchNodePair: `all_matches.size() < min_matches` -> no RANSAC (node.cpp:1319);
  // getRelativeTransformationTo: `size <= min_matches` -> false      (node.cpp:1087)
  if (n_all > rc.min_matches) {
    uint32_t thr = (uint32_t)rc.min_matches;                                  // :1094
    if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
    // (the double of max_dist_m is formed where it is compared: as a loop-invariant value it sat in a register PAIR for the length
    // of the kernel -- one of the values the allocator sent to scratch)
    auto max_dist_d = [&]() { float f = rc.max_dist_m; asm volatile("" : "+v"(f)); return (double)f; };
    rmse = 1e6f;  // :1112
    const uint32_t seed_uid = mix32(mix32(rc.seed ^ 0x9E3779B9u) + w.uid * 0x85EBCA6Bu);

    float hypR[9], hypt[3];
    bool hyp_nan = true;
    bool hyp_viable = false;     // this lane's hypothesis can reach `thr` candidates at all (pre-screen below)
    uint64_t viable_mask = 0ull;  // ... of the 64 hypotheses of the batch
    int hyp_base = -kWave;  // iteration index of lane 0's hypothesis (none yet)
    bool done = false;

    auto gen_hypotheses = [&](int k0) {
        // ---- LANE = HYPOTHESIS: sample + 4-point fit for iterations k0 .. k0+63
        hyp_base = k0;
        const uint32_t iter = (uint32_t)(k0 + lane);
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
            const bool dup = (cnt > 0 && ids[0] == id1) || (cnt > 1 && ids[1] == id1) ||
                             (cnt > 2 && ids[2] == id1);
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
          if (s4 < cnt) acc.add(lds.M, (int)ids[s4]);
        tfc_get_transformation(acc, hypR, hypt);
        hyp_nan = has_nan12(hypR, hypt);
        // ---- pre-screen, still LANE = HYPOTHESIS.  With inlier ratios of real loop-closure candidates almost every
        // 4-point hypothesis is junk: its first scoring (:1148) finds fewer than `thr` inliers, the refinement loop
        // breaks (:1154) and the iteration ends with refined_matches empty.  An iteration is certainly of that kind when
        // fewer than `thr` matches can pass errorFunction2's shortcut test (misc.cpp:726-735): an upper bound of that
        // count comes from the float evaluation of dsq with score_passes' error band (a match counts unless its dsq_f
        // is provably above the threshold; NaN counts).  Every lane walks all matches for ITS hypothesis -- the match
        // record is the same LDS address for all lanes (broadcast) -- which costs ~1/4 of a slot's pass-1 scoring per
        // hypothesis and spares a junk iteration the whole slot machinery (open, score, bookkeeping round, close).
        if (!prescreen) {  // (wave-uniform) every finite hypothesis takes a slot
          hyp_viable = !hyp_nan;
        } else {
          hyp_viable = !hyp_nan && prescreen_may_pass(hypR, hypt, lds.M, n_all, pmax, rc) >= thr;
        }
        viable_mask = __ballot(hyp_viable);
        PH_MARK(2)
    };
    // slot g <- iteration k: first transform = its 4-point hypothesis
    auto open_slot = [&](int g, int k) {
      // (recording waves get here through next_viable(), which has generated the batch that holds k)
      if (MODE != kRecord && (hyp_base < 0 || k < hyp_base || k >= hyp_base + kWave)) gen_hypotheses(k);
      const int hl = k - hyp_base;
      float R0[9], t0[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) R0[i] = bcast_f(hypR[i], hl);
#pragma unroll
      for (int i = 0; i < 3; ++i) t0[i] = bcast_f(hypt[i], hl);
      // a NaN transform leaves the refinement loop at once (:1144); so does, after its first scoring, a hypothesis the
      // pre-screen has shown to be junk (:1154) -- with the same outcome: refined_matches stays empty
      const bool nan0 = ((viable_mask >> hl) & 1ull) == 0ull;
      if (lane == 0) {
        Slot& sl = lds.slot[g];
#pragma unroll
        for (int i = 0; i < 9; ++i) { sl.R[i] = R0[i]; sl.rR[i] = IR[i]; }  // :1137 refined = Identity
#pragma unroll
        for (int i = 0; i < 3; ++i) { sl.t[i] = t0[i]; sl.rt[i] = 0.f; }
#pragma unroll
        for (int r = 0; r < kRounds; ++r) sl.rmask[r] = 0ull;
        sl.rerr = 1e6;          // :1133
        sl.rn = 0;              // :1134
        sl.active = nan0 ? 0 : 1;  // a NaN transform leaves the refinement loop (:1144)
        sl.round = 0;
        sl.iter = k;
      }
    };
    // One pass of the refinement loop (`for refinements = 1 .. 19`, :1140) for every active slot; false when no
    // slot stays active.
    auto refine_round = [&]() -> bool {
        // ---- scorings (:1148) of the active slots, one after the other (lane = match) ...
        int cn_mine = 0;  // lane s: inliers of slot s's scoring when its error sum is looked at
        int n_sum_max = 0;
        for (int g = 0; g < kSlots; ++g) {
          Slot& sl = lds.slot[g];
          if (__builtin_amdgcn_readfirstlane(sl.active) == 0) continue;
          float curR[9], curt[3];
#pragma unroll
          for (int i = 0; i < 9; ++i) curR[i] = sl.R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) curt[i] = sl.t[i];
          uint64_t inl_mask[kRounds];
          int n_inl;
          PH_MARK(5)
          // a scoring with fewer inliers than max(threshold, refined_matches.size()) is rejected whatever
          // its error is (:1154, :1160): the scorer may stop counting as soon as that is certain
          const uint32_t need = max(thr, (uint32_t)__builtin_amdgcn_readfirstlane(sl.rn));
          score_passes(curR, curt, n_all, need, rc, lds, pmax, ec_region + g * kEcRow, inl_mask, n_inl PH_PASS);
          PH_MARK(3)
          PH_COUNT(6)
          if (lane == 0) {
#pragma unroll
            for (int r = 0; r < kRounds; ++r) sl.cmask[r] = inl_mask[r];
            sl.cn = n_inl;
          }
          if (!((uint32_t)n_inl < need || n_inl < 3)) {  // else mean_error = 1e9 (:1012-1014) / rejected anyway
            if (lane == g) cn_mine = n_inl;
            n_sum_max = max(n_sum_max, n_inl);
          }
        }
        wave_sync();
        PH_MARK(5)
        PH_COUNT(18)
        PH_ADDX(19, n_sum_max)
#ifdef RGBDFE_FENCE_SUMS  // diagnostics build: drop this CU's L1 before the error rows are read back
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        // ... then their sequential error sums side by side and the bookkeeping (:1154-1166), lane = slot
        const double sum_mine = n_sum_max > 0 ? sum_rows(ec_region, lds, lane, cn_mine, n_sum_max PH_PASS) : 0.0;  // wave-uniform
        const double err_mine = cn_mine > 0 ? sqrt(sum_mine / (double)cn_mine) : 1e9;  // :1016-1017
        PH_MARK(11)
        bool still = false;
        if (lane < kSlots) {
          Slot& sl = lds.slot[lane];
          if (sl.active) {
            const int n_inl = sl.cn, rn = sl.rn;
            if (!((uint32_t)n_inl < thr || err_mine > max_dist_d())) {   // :1154
              if (n_inl >= rn && err_mine <= sl.rerr) {                 // :1160
                still = (n_inl != rn);                                 // :1166
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
        wave_sync();
        if (!any_active) return false;
        // ---- refits (:1142): the weighted-mean recurrences of all active slots side by side, then
        // ONE batched 3x3 SVD with lane = slot
        int n_mine = 0, k256_mine = 0, n_max = 0, n_min = RGBDFE_MAX_MATCHES;
        PH_MARK(5)
        for (int g = 0; g < kSlots; ++g) {
          Slot& sl = lds.slot[g];
          if (__builtin_amdgcn_readfirstlane(sl.active) == 0) continue;
          uint64_t m5[kRounds];
#pragma unroll
          for (int r = 0; r < kRounds; ++r) m5[r] = uniform_u64(sl.rmask[r]);
          int k256_g;
          const int n_g = fit_compact(g, m5, w_nonzero, lds.u.fit, k256_g, lane);
          if (lane / 9 == g) { n_mine = n_g; k256_mine = k256_g; }
          n_max = max(n_max, n_g);
          n_min = min(n_min, n_g);
          PH_COUNT(7)
        }
        wave_sync();
        PH_MARK(8)
        PH_COUNT(9)
        PH_ADD(10, n_max)
        Tfc mine;
        mine.reset();
        {
          float C, m1, m2;
          if (fast_alpha) fit_recurrence<true>(n_mine, k256_mine, n_min, n_max, lds.u.fit, lds.M, C, m1, m2, lane);
          else fit_recurrence<false>(n_mine, k256_mine, n_min, n_max, lds.u.fit, lds.M, C, m1, m2, lane);
          PH_MARK(4)
          // lane s (< G) collects the state of slot s
          // (the lane made opaque here: derived from the kernel's `lane` the shuffle addresses were formed at the top of the
          // kernel, kept for its whole length and spilled)
          int lane_here = lane;
          asm volatile("" : "+v"(lane_here));
          const int src = min(lane_here, kSlots - 1) * 9;
#pragma unroll
          for (int x = 0; x < 9; ++x) mine.C[x] = __shfl(C, src + x);
#pragma unroll
          for (int j = 0; j < 3; ++j) mine.m1[j] = __shfl(m1, src + j);
#pragma unroll
          for (int i = 0; i < 3; ++i) mine.m2[i] = __shfl(m2, src + 3 * i);
        }
        wave_sync();
        {
          float fR[9], ft[3];
          tfc_get_transformation(mine, fR, ft);
          const bool fnan = has_nan12(fR, ft);
          if (lane < kSlots) {
            Slot& sl = lds.slot[lane];
            if (sl.active) {
#pragma unroll
              for (int i = 0; i < 9; ++i) sl.R[i] = fR[i];
#pragma unroll
              for (int i = 0; i < 3; ++i) sl.t[i] = ft[i];
              if (fnan) sl.active = 0;  // :1144
            }
          }
          PH_MARK(12)
        }
        wave_sync();
      return true;
    };

    if (lane < kSlots) { lds.slot[lane].active = 0; lds.slot[lane].iter = -1; }
    wave_sync();

    int it = 0;
    if (MODE == kRecord) {
      // Recording: the iterations of the chunk are independent and their outcomes are addressed by iteration index,
      // so a slot whose iteration has left its refinement loop is written out and refilled with the next
      // iteration at once -- the batched rounds stay full instead of waiting for a window's slowest slot.
      int k_next = k_begin;
      // next viable iteration of the chunk (or -1); generates hypothesis batches as needed and writes the (empty)
      // records of their junk iterations, 64 at a time, lane = iteration
      auto next_viable = [&]() -> int {
        while (k_next < k_end) {
          if (hyp_base < 0 || k_next < hyp_base || k_next >= hyp_base + kWave) {
            gen_hypotheses(k_next);
            const int k = hyp_base + lane;
            if (k < k_end && !hyp_viable) sum_pair[k] = IterSum{1e6, 0, 0};  // refined_matches stays empty (:1133-1134)
          }
          const int off0 = k_next - hyp_base;
          const uint64_t rest = viable_mask >> off0;
          if (rest == 0ull) { k_next = min(k_end, hyp_base + kWave); continue; }
          const int k = k_next + (int)__builtin_ctzll(rest);
          if (k >= k_end) { k_next = k_end; break; }
          k_next = k + 1;
          return k;
        }
        return -1;
      };
      while (n_all >= 4) {
        if (lane < kSlots) {
          Slot& sl = lds.slot[lane];
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
        wave_sync();
        bool occupied = false;
        for (int g = 0; g < kSlots; ++g) {
          int it_g = __builtin_amdgcn_readfirstlane(lds.slot[g].iter);
          if (it_g < 0) {
            const int k = next_viable();
            if (k >= 0) {
              open_slot(g, k);
              it_g = k;
            }
          }
          occupied |= it_g >= 0;
        }
        wave_sync();
        if (!occupied) break;
        refine_round();
      }
#ifdef RGBDFE_PROFILE_PHASES
      PH_MARK(5)
      if (lane == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_totals[i], (unsigned long long)ph[i]);
        atomicAdd(&g_phase_totals[16], 1ull);
        atomicAdd(&g_phase_totals[17], (unsigned long long)(k_end - k_begin));
        atomicAdd(&g_phase_totals[18], (unsigned long long)ph[18]);
        atomicAdd(&g_phase_totals[19], (unsigned long long)ph[19]);
        atomicAdd(&g_phase_totals[20], (unsigned long long)ph[20]);
        atomicAdd(&g_phase_totals[21], (unsigned long long)ph[21]);
      }
#endif
    } else if (MODE == kReplay) {
      // the in-order bookkeeping: adopt its outcome and the record of the best iteration.  One-kernel recording stage: it
      // ran in replay_walk_kernel, phase by phase.  Split path (plan.final_walk): the refinement kernel has walked what it
      // had to know between a pair's windows (nothing at all for a pair recorded in one window) and left the end of the
      // pair's recorded range in WalkState::speculate; the rest of the walk is this wave's.
      const WalkState ws = plan.walk[pair];
      WalkRegs wr{ws.it, ws.real_iterations, ws.valid_iterations, ws.best_idx, ws.best_n, ws.rmse, ws.state < 0};
      if (plan.final_walk && n_all >= 4) {
        const uint64_t* __restrict__ vm_pair = plan.vmask + (size_t)pair * (size_t)plan.vmask_words;
        walk_records<false>(wr, min(ws.speculate, rc.ransac_iterations), rc.ransac_iterations, n_all, thr, sum_pair,
                            [&](int k) { return ((vm_pair[k >> 6] >> (k & 63)) & 1ull) != 0ull; }, lane);
      }
      valid_iterations = wr.valid_iterations;
      real_iterations = wr.real_iterations;
      best_n = wr.best_n;
      rmse = wr.rmse;
      if (wr.best_idx >= 0 && lane == 0) {
        const IterRec& r = rec_pair[wr.best_idx];
        Hyp& b = lds.best;
#pragma unroll
        for (int i = 0; i < 9; ++i) b.R[i] = r.rR[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) b.t[i] = r.rt[i];
#pragma unroll
        for (int q = 0; q < kRounds; ++q) b.mask[q] = r.rmask[q];
        b.n = r.rn;
        b.nan = 0;
        b.err = r.rerr;
      }
      wave_sync();
    } else {
    for (; !done && it < rc.ransac_iterations && n_all >= 4;) {  // :1130
      const int k0 = real_iterations;
      // The first iteration runs alone: an easy pair leaves the loop right after it (:1188) and must
      // not pay for a speculative window.
      const int G = (k0 == 0) ? 1 : kSlots;
      // ---- open the window: slot g <- iteration k0 + g; refine all of them to the end
      for (int g = 0; g < G; ++g) open_slot(g, k0 + g);
      wave_sync();
      while (refine_round()) {}
      // ---- replay the window in iteration order (:1171-1190)
      for (int g = 0; g < G; ++g) {
        if (!(it < rc.ransac_iterations)) { done = true; break; }
        real_iterations++;  // :1139
        const Slot& sl = lds.slot[g];
        const int refined_n = __builtin_amdgcn_readfirstlane(sl.rn);
        const double refined_error = sl.rerr;
        if (refined_n > 0) {  // :1171
          valid_iterations++;
          if (refined_error <= (double)rmse && refined_n >= best_n && (uint32_t)refined_n >= thr) {  // :1177
            rmse = (float)refined_error;  // :1182
            wave_sync();
            if (lane == 0) {
              Hyp& b = lds.best;
#pragma unroll
              for (int i = 0; i < 9; ++i) b.R[i] = sl.rR[i];
#pragma unroll
              for (int i = 0; i < 3; ++i) b.t[i] = sl.rt[i];
#pragma unroll
              for (int r = 0; r < kRounds; ++r) b.mask[r] = sl.rmask[r];
              b.n = refined_n;
              b.nan = 0;
              b.err = refined_error;
            }
            wave_sync();
            best_n = refined_n;
            // (n_all's double formed here, from a value the compiler cannot see through: as three loop-invariant products they sat
            // in register pairs for the length of the kernel -- the values the allocator sent to scratch)
            int n_all_here = n_all;
            asm volatile("" : "+v"(n_all_here));
            const double n_all_d = (double)n_all_here;
            if ((double)refined_n > n_all_d * 0.5) it += 10;   // :1186
            if ((double)refined_n > n_all_d * 0.75) it += 10;  // :1187
            if ((double)refined_n > n_all_d * 0.8) { done = true; break; }  // :1188
          }
        }
        ++it;
      }
    }
    }  // kWhole
    if (MODE != kRecord && valid_iterations == 0) {  // :1192 identity hypothesis
      if (MODE == kReplay) load_points();
      uint64_t inl_mask[kRounds];
      int n_inl;
      double inlier_error;
      score_hypothesis(IR, It, n_all, thr + 1u, rc, lds, pmax, ec_region, inl_mask, n_inl, inlier_error PH_PASS);  // needs > thr (:1206)
      if ((uint32_t)n_inl > thr && inlier_error < max_dist_d()) {  // :1206
        hyp_store(lds.best, IR, It, inl_mask, n_inl, 0, inlier_error);
        best_n = n_inl;
        rmse = (float)inlier_error;
        valid_iterations++;
      }
    }
    found = (uint32_t)best_n >= thr;  // :1275
  }
  wave_sync();

  // ------------------------------------------------------------------ result POD
  if (MODE != kRecord && lane == 0) {
    const Hyp& b = lds.best;
    out->n_all = n_all;
    out->n_inl = best_n;
    out->rmse = rmse;
    // Eigen::Matrix4f column-major
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) out->trafo[j * 4 + i] = b.R[i * 3 + j];
      out->trafo[12 + i] = b.t[i];
      out->trafo[i * 4 + 3] = 0.0f;
    }
    out->trafo[15] = 1.0f;
    out->pad0 = 0;
#ifdef RGBDFE_DEBUG_CLASS  // diagnostics build: the pair's class of the record / replay plan
    if (MODE == kReplay) out->pad0 = (uint32_t)plan.walk[pair].speculate;
#endif
    out->valid_iterations = valid_iterations;
    out->real_iterations = real_iterations;
    if (found) {
      out->id1 = w.tid;  // node.cpp:1337
      out->id2 = w.qid;  // node.cpp:1338
      out->info_scale = (double)((float)best_n / (rmse * rmse));  // node.cpp:1335
    } else {
      out->id1 = -1;  // node.cpp:1419-1422
      out->id2 = -1;
      out->info_scale = 0.0;
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) out->inlier_mask[r] = b.mask[r];
#ifdef RGBDFE_PROFILE_PHASES
    PH_MARK(5)
    uint64_t* dbg = reinterpret_cast<uint64_t*>(out->all_q);
    for (int i = 0; i < 16; ++i) dbg[i] = ph[i];
#endif
  }
}

void launch_select_ransac(const float4* xyz_pool, const PairWork* work, const uint32_t* keys,
                          uint32_t key_planes, rgbdfe_match_result* results, uint32_t max_kp,
                          uint32_t n_pairs, const RansacConst& rc, PairPrep* prep, double* ec_pool, hipStream_t stream) {
  if (n_pairs == 0) return;
  SiftMatchList none{};
  hipLaunchKernelGGL(pair_prep_kernel<false>, dim3(n_pairs), dim3(kWave), 0, stream, xyz_pool, work, keys, key_planes,
                     none, results, max_kp, n_pairs, rc, prep, (uint32_t*)nullptr);
  RecordPlan plan{};
  plan.prep = prep;
  plan.ec_pool = ec_pool;
  hipLaunchKernelGGL(select_ransac_kernel<kWhole>, dim3(n_pairs), dim3(kWave), 0, stream, work, results, n_pairs, rc,
                     plan);
}

// SIFT: sift_sort_kernel leaves each pair's match list in keepStrongestMatches order, pair_prep_kernel<true> takes
// its head
static void launch_sift_prep(const float4* xyz_pool, const PairWork* work, uint16_t* sm_q, uint16_t* sm_t, float* sm_d,
                             const int32_t* sm_n, float* all_dist, rgbdfe_match_result* results, uint32_t max_kp,
                             uint32_t n_pairs, const RansacConst& rc, PairPrep* prep, uint32_t* zero64, hipStream_t stream) {
  hipLaunchKernelGGL(sift_sort_kernel, dim3(n_pairs), dim3(kSortThreads), 0, stream, sm_q, sm_t, sm_d, sm_n, max_kp, n_pairs,
                     rc.max_matches);
  SiftMatchList sm{sm_q, sm_t, sm_d, sm_n, all_dist};
  hipLaunchKernelGGL(pair_prep_kernel<true>, dim3(n_pairs), dim3(kWave), 0, stream, xyz_pool, work,
                     (const uint32_t*)nullptr, 1u, sm, results, max_kp, n_pairs, rc, prep, zero64);
}

void launch_select_ransac_sift(const float4* xyz_pool, const PairWork* work, uint16_t* sm_q,
                               uint16_t* sm_t, float* sm_d, const int32_t* sm_n,
                               float* all_dist, rgbdfe_match_result* results, uint32_t max_kp,
                               uint32_t n_pairs, const RansacConst& rc, PairPrep* prep, double* ec_pool,
                               hipStream_t stream) {
  if (n_pairs == 0) return;
  launch_sift_prep(xyz_pool, work, sm_q, sm_t, sm_d, sm_n, all_dist, results, max_kp, n_pairs, rc, prep, nullptr, stream);
  RecordPlan plan{};
  plan.prep = prep;
  plan.ec_pool = ec_pool;
  hipLaunchKernelGGL(select_ransac_kernel<kWhole>, dim3(n_pairs), dim3(kWave), 0, stream, work, results, n_pairs, rc,
                     plan);
}

// The reference's in-order bookkeeping (node.cpp:1130-1191: `it += 10 / 20`, the 80 % exit, best-so-far) over the records
// of iterations [real_iterations, min(phase_end, state)): one wave per pair, 64 records fetched per step, the sequential
// decisions on wave-uniform values.  Leaves either "finished" (state < 0) or a tighter bound on the iterations the pair
// can still need; resumes where the previous phase stopped.
__global__ __launch_bounds__(kWave) void replay_walk_kernel(const IterSum* __restrict__ sums, WalkState* __restrict__ walk,
                                                            const PairPrep* __restrict__ prep, uint32_t n_pairs,
                                                            const RansacConst rc, int phase_begin, int phase_end,
                                                            int spec_end, int may_speculate,
                                                            const uint8_t* __restrict__ preclass, int phase_index,
                                                            const uint64_t* __restrict__ vmask, int vmask_words) {
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x;
  WalkState ws = walk[pair];
  const int I = rc.ransac_iterations;
  if (phase_begin == 0) {
    ws.state = I;
    ws.it = 0; ws.real_iterations = 0; ws.valid_iterations = 0;
    ws.best_idx = -1; ws.best_n = 0;
    ws.rmse = 1e6f;  // :1112
    ws.speculate = 0;
  } else if (ws.state < 0) {
    return;
  }
  const int n_all = prep[pair].n_all;
  // the records of a speculating pair reach as far as its recording waves were allowed to go
  // (first phase of a split plan: the pairs the hypothesis kernel has pre-classified junk-heavy were recorded to the end)
  const bool to_spec_end = spec_end > phase_end && (phase_begin != 0 ? effective_class(walk, pair, n_pairs) == 2
                                                                     : (preclass != nullptr && preclass[pair] == 2));
  const int recorded_end = min(to_spec_end ? spec_end : phase_end, ws.state);
  uint32_t thr = (uint32_t)rc.min_matches;                                         // :1094
  if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);  // :1095-1098
  const IterSum* __restrict__ sum_pair = sums + (size_t)pair * (size_t)(I > 0 ? I : 0);
  WalkRegs wr{ws.it, ws.real_iterations, ws.valid_iterations, ws.best_idx, ws.best_n, ws.rmse, false};
  const bool runs = n_all > rc.min_matches && n_all >= 4;  // :1087, :1130
  // split path: an iteration the pre-screen rejected has no summary at all -- the pair's viable mask says so (its bit is
  // clear) and it counts as {1e6, 0} (the hypothesis kernel used to write 16 bytes for every one of them: 12.8 MB per batch
  // of configs[1] written and read back up to four times)
  const uint64_t* __restrict__ vm_pair = vmask != nullptr ? vmask + (size_t)pair * (size_t)vmask_words : nullptr;
  if (runs)
    walk_records<false>(wr, recorded_end, I, n_all, thr, sum_pair,
                        [&](int k) { return vm_pair == nullptr || ((vm_pair[k >> 6] >> (k & 63)) & 1ull) != 0ull; }, lane);
  const int it = wr.it, real_iterations = wr.real_iterations, valid_iterations = wr.valid_iterations;
  const int best_idx = wr.best_idx, best_n = wr.best_n;
  const float rmse = wr.rmse;
  const bool done = wr.done;
  if (lane == 0) {
    // records ran out before the loop ended: at most (I - it) more iterations can follow
    ws.state = (runs && !done && it < I) ? real_iterations + (I - it) : -1;
    // after the first phase: nothing has jumped `it` ahead yet -> record the rest of this pair in one go
    // class of the pair for the rest of the plan (see select_ransac_kernel): 1 = no jump so far, 2 = ... and junk-heavy
#ifdef RGBDFE_NO_CLASS2  // diagnostics build
    may_speculate = 0;
#endif
    if (may_speculate) {
      const bool no_jump = ws.state >= 0 && it == real_iterations;
      const bool junk_heavy = valid_iterations * kClass2Den <= real_iterations * kClass2Num;
      ws.speculate = no_jump ? (junk_heavy ? 2 : 1) : 0;
      // class-1 pairs of the batch are counted: when there are only a few of them they are recorded like class 2
      // (see effective_class) instead of keeping the later phases' launches alive for a handful of long waves
      if (ws.speculate == 1) atomicAdd(&walk[n_pairs].state, 1);
    }
    ws.it = it; ws.real_iterations = real_iterations; ws.valid_iterations = valid_iterations;
    ws.best_idx = best_idx; ws.best_n = best_n; ws.rmse = rmse;
    walk[pair] = ws;
    // "a pair is still running after phase phase_index": the next refinement launch ends at once when nobody says so
    if (ws.state >= 0) walk[n_pairs].best_n = phase_index + 1;
  }
}

// Which recording stage a batch takes.  ransac_split.hip's hypothesis + refinement kernels (bit-identical to the
// one-kernel stage select_ransac_kernel<kRecord>) are the default for every record / replay plan; the refinement kernel's
// waves synchronise through one hardware barrier per half-round and nothing else, so there is no batch shape it has to be
// kept away from (round 4's streaming version, with LDS spin locks, stalled on small batches and was gated to phased plans).
//   RGBDFE_RANSAC_SPLIT unset or = 1: the split path; = 0: the one-kernel stage (A/B runs).
static int ransac_split_mode() {
  static const int mode = getenv("RGBDFE_RANSAC_SPLIT") ? (atoi(getenv("RGBDFE_RANSAC_SPLIT")) != 0 ? 1 : 0) : -1;
  return mode;
}

// Record / replay schedule.  The iteration range is covered in `n_phases` phases ending at phase_ends[]: each phase
// launches the recording waves (the phase in equal shares of at most chunk_iters iterations per wave; waves of finished
// pairs and waves beyond a pair's remaining need return at once) and the walk (one small wave per pair), which either
// ends the pair's loop or tightens the bound on the iterations it can still need.  One launch of result waves follows
// the last phase.  One phase = full speculation (lowest latency); several phases stop recording where the reference's
// bookkeeping stops iterating.  recs: n_pairs x rc.ransac_iterations records; walk: n_pairs states; prep: filled.
static void launch_record_replay(const PairWork* work, rgbdfe_match_result* results, uint32_t n_pairs,
                                 const RansacConst& rc, const PairPrep* prep, IterRec* recs, WalkState* walk,
                                 double* ec_pool, int chunk_iters, const int* phase_ends, int n_phases,
                                 hipStream_t stream) {
  int begin = 0;
  RecordPlan plan{};
  plan.recs = recs; plan.walk = walk; plan.prep = prep; plan.ec_pool = ec_pool;
  const size_t n_recs = (size_t)n_pairs * (size_t)(rc.ransac_iterations > 0 ? rc.ransac_iterations : 0);
  plan.sums = reinterpret_cast<IterSum*>(recs + n_recs);
  plan.n_phases_total = n_phases;  // a single-phase plan (small batches: full speculation) always pre-screens
  const int I = rc.ransac_iterations;
  const int split_mode = ransac_split_mode();
  const bool split = split_mode != 0;
  // walk[n_pairs]: the batch's counters (class-1 pairs; split path: unit counters, "still running" flag), zero at the start
  // (the split path's hypothesis kernel does it itself)
  if (!split) (void)hipMemsetAsync(walk + n_pairs, 0, sizeof(WalkState), stream);
  SplitPlan sp{};
  if (split) {
    // every iteration's hypothesis + pre-screen, all pairs, one launch (lane = iteration); the phases below refine the
    // viable ones
    sp.recs = recs; sp.sums = plan.sums; sp.walk = walk; sp.prep = prep;
    sp.vmask = reinterpret_cast<uint64_t*>(plan.sums + n_recs);
    sp.vmask_words = ransac_split_words_per_pair(I);
    sp.preclass = reinterpret_cast<uint8_t*>(sp.vmask + (size_t)n_pairs * (size_t)sp.vmask_words);
    sp.order_cnt = ransac_split_order_cnt(recs, n_pairs, I);
    sp.order = sp.order_cnt + kOrderBuckets;
    // phased plans: pairs the pre-screen alone shows to be junk-heavy skip the first phase's launch + walk and record
    // everything at once (the classes only schedule the recording: the walk decides the outcome either way)
    static const bool no_pre = getenv("RGBDFE_NO_PRECLASS") && atoi(getenv("RGBDFE_NO_PRECLASS")) != 0;  // A/B runs
    sp.preclass_iters = (n_phases > 1 && !no_pre) ? phase_ends[0] : 0;
    // ONE refinement launch for the whole batch.  Phased plans: a unit is a pair, whose range the kernel records in windows
    // with the in-order walk between them (SplitPlan); latency batches (one phase, full speculation): a pair's range in
    // shares of 4 x chunk_iters iterations so that a handful of pairs still fills the chip.  The result waves below finish
    // the walk.
    sp.phased = n_phases > 1 ? 1 : 0;
    sp.n_phases = n_phases < 4 ? n_phases : 4;
    for (int p = 0; p < 4; ++p) sp.phase_ends[p] = p < sp.n_phases ? (p == sp.n_phases - 1 ? I : phase_ends[p]) : I;
    // Full speculation: a pair's range in shares of 4 x chunk_iters iterations so that a handful of pairs still fills the chip
    // -- but no more shares than it takes to give every unit buffer of the launch (workgroups x resident units) a unit: a
    // pair cut into shares is loaded once per share, a pair in one piece is one workgroup's chain of passes (0.002 z^2, 512
    // pairs: whole pairs 2.66 ms, shares of 28 1.83 ms; 0.01 z^2: 0.397 / 0.427 ms).
    int share = chunk_iters >= 28 ? (I > 0 ? I : 1) : chunk_iters * 4;
    {
      const int buffers = ransac_split_wgs() * 4;
      const int want = (int)((buffers + (int)n_pairs - 1) / (int)(n_pairs > 0 ? n_pairs : 1));   // shares per pair that fill them
      const int by_want = I > 0 ? (I + want - 1) / (want > 0 ? want : 1) : 1;
      if (by_want > share) share = by_want;
    }
    if (share > ransac_split_max_share()) share = ransac_split_max_share();   // (a unit's list of viable iterations)
    sp.n_shares = sp.phased ? 1 : (I > 0 ? (I + share - 1) / share : 1);
    sp.share_iters = sp.phased ? I : (I > 0 ? (I + sp.n_shares - 1) / sp.n_shares : share);
    // (the launch's unit counter: a spare word of walk[n_pairs], zeroed by the hypothesis kernel)
    sp.unit_counter = reinterpret_cast<uint32_t*>(&walk[n_pairs].it);
    launch_ransac_hyp(work, n_pairs, rc, sp, stream);
    if (I > 0) launch_ransac_refine(n_pairs, rc, sp, stream);
    plan.final_walk = 1;
    plan.vmask = sp.vmask;
    plan.vmask_words = sp.vmask_words;
  } else {
  for (int p = 0; p < n_phases; ++p) {
    const int end = phase_ends[p];
    // The second phase of a phased plan covers everything that is left for the pairs of class 2 (see replay_walk_kernel);
    // waves that have nothing to do for their pair return at once.
#ifdef RGBDFE_NO_SUBGRID_B  // diagnostics build
    const bool spec = false;
#else
    const bool spec = n_phases > 2 && p == 1 && I > end;
#endif
    const int cover = spec ? I : end;
    {
      // the one-kernel recording launch of this phase
      // sub-grid A: the phase in ceil(length / chunk) equal shares (a short last wave would be the launch's straggler)
      const int n_chunks = (end - begin + chunk_iters - 1) / chunk_iters;
      plan.n_chunks = (uint32_t)n_chunks;
      plan.chunk_iters = n_chunks > 0 ? (end - begin + n_chunks - 1) / n_chunks : chunk_iters;
      const int cover_rec = spec ? I : end;
      const int n_chunks_b = spec ? (cover_rec - begin + kWave - 1) / kWave : 0;
      plan.n_chunks_b = (uint32_t)n_chunks_b;
      plan.chunk_iters_b = n_chunks_b > 0 ? (cover_rec - begin + n_chunks_b - 1) / n_chunks_b : kWave;
      plan.phase_begin = begin;
      plan.phase_end = end;
      plan.spec_end = cover_rec;
      if (cover_rec > begin)
        hipLaunchKernelGGL(select_ransac_kernel<kRecord>,
                           dim3(8u * ((n_pairs + 7u) / 8u) * (plan.n_chunks + plan.n_chunks_b)), dim3(kWave), 0, stream, work,
                           results, n_pairs, rc, plan);  // 8 XCD segments x pairs per segment x shares per pair
    }
    hipLaunchKernelGGL(replay_walk_kernel, dim3(n_pairs), dim3(kWave), 0, stream, plan.sums, walk, prep, n_pairs, rc, begin,
                       end, cover, (n_phases > 2 && p == 0) ? 1 : 0, (const uint8_t*)nullptr, p,
                       (const uint64_t*)nullptr, 0);
    begin = end;
  }
  }
  plan.n_chunks = 1;
  plan.n_chunks_b = 0;
  hipLaunchKernelGGL(select_ransac_kernel<kReplay>, dim3(n_pairs), dim3(kWave), 0, stream, work, results, n_pairs, rc,
                     plan);
}

void launch_select_ransac_latency(const float4* xyz_pool, const PairWork* work, const uint32_t* keys,
                                  uint32_t key_planes, rgbdfe_match_result* results, uint32_t max_kp,
                                  uint32_t n_pairs, const RansacConst& rc, PairPrep* prep, IterRec* recs, WalkState* walk,
                                  double* ec_pool, int chunk_iters, const int* phase_ends, int n_phases,
                                  hipStream_t stream) {
  if (n_pairs == 0) return;
  SiftMatchList none{};
  hipLaunchKernelGGL(pair_prep_kernel<false>, dim3(n_pairs), dim3(kWave), 0, stream, xyz_pool, work, keys, key_planes,
                     none, results, max_kp, n_pairs, rc, prep, ransac_split_order_cnt(recs, n_pairs, rc.ransac_iterations));
  launch_record_replay(work, results, n_pairs, rc, prep, recs, walk, ec_pool, chunk_iters, phase_ends, n_phases, stream);
}

void launch_select_ransac_sift_latency(const float4* xyz_pool, const PairWork* work, uint16_t* sm_q,
                                       uint16_t* sm_t, float* sm_d, const int32_t* sm_n, float* all_dist,
                                       rgbdfe_match_result* results, uint32_t max_kp, uint32_t n_pairs,
                                       const RansacConst& rc, PairPrep* prep, IterRec* recs, WalkState* walk,
                                       double* ec_pool, int chunk_iters, const int* phase_ends, int n_phases,
                                       hipStream_t stream) {
  if (n_pairs == 0) return;
  launch_sift_prep(xyz_pool, work, sm_q, sm_t, sm_d, sm_n, all_dist, results, max_kp, n_pairs, rc, prep,
                   ransac_split_order_cnt(recs, n_pairs, rc.ransac_iterations), stream);
  launch_record_replay(work, results, n_pairs, rc, prep, recs, walk, ec_pool, chunk_iters, phase_ends, n_phases, stream);
}

// =====================================================================================================================
// "G2O Refinement" (node.cpp:1222-1268 + getTransformFromMatchesG2O, transformation_estimation.cpp:37-170), after the
// RANSAC result of a pair exists: a two-view bundle adjustment over the inlier matches -- camera 2 = the newer node, fixed
// at the identity; camera 1 = the earlier node, free, started at the RANSAC estimate; one free 3-D point per match started
// at the newer node's position; two (u, v, depth) edges per match, information diag(1, 1, 1 / depth_covariance);
// K = (521, 521, 319.5, 239.5) (:56); `g2o_transformation_refinement` Gauss-Newton steps -- then the re-scoring and the
// accept / refine-again / adopt rules of :1233-1260.
// g2o itself is not in the reference tree: the optimiser is the oracle's restatement (orc_g2o_refine: normal equations
// with the 3x3 point blocks eliminated, the reduced 6x6 system by Cholesky, VertexSE3's multiplicative update), followed
// operation by operation.  One wave per pair: LANE = MATCH for the per-match blocks (each lane owns matches l, l+64, ...),
// the 27 reduced sums by an xor butterfly (the oracle adds in the same order), the 6x6 solve redundantly in every lane.
// =====================================================================================================================
struct GnShared {
  double X[RGBDFE_MAX_MATCHES][3];     // the point vertices
  float2 kq[RGBDFE_MAX_MATCHES];       // KeyPoint.pt of the selected matches in the newer / the earlier node
  float2 kt[RGBDFE_MAX_MATCHES];
  uint16_t sel[RGBDFE_MAX_MATCHES];    // k-th selected match
};

__device__ __forceinline__ void gn_proj(const double* Y, double* e, double* J) {
  const double fx = 521.0, fy = 521.0, cx = 319.5, cy = 239.5;  // transformation_estimation.cpp:56
  const double iz = 1.0 / Y[2];
  e[0] = fx * (Y[0] * iz) + cx;
  e[1] = fy * (Y[1] * iz) + cy;
  e[2] = Y[2];
  J[0] = fx * iz; J[1] = 0.0;     J[2] = -(fx * (Y[0] * iz)) * iz;
  J[3] = 0.0;     J[4] = fy * iz; J[5] = -(fy * (Y[1] * iz)) * iz;
  J[6] = 0.0;     J[7] = 0.0;     J[8] = 1.0;
}

// the blocks of one match: adds its Schur terms to acc[27] (upper triangle of S, then g); returns Hpp^-1, bp, Hcp
__device__ __forceinline__ void gn_match_terms(const double* X, const double* R1, const double* t1, const double* m1,
                                               const double* m2, double wz, double* acc, double* Hpp_inv, double* bp,
                                               double* Hcp) {
  double e2[3], J2[9], e1[3], Jp1[9];
  gn_proj(X, e2, J2);
#pragma unroll
  for (int i = 0; i < 3; ++i) e2[i] = e2[i] - m2[i];
  const double dX[3] = {X[0] - t1[0], X[1] - t1[1], X[2] - t1[2]};
  double Y[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Y[i] = (R1[0 * 3 + i] * dX[0] + R1[1 * 3 + i] * dX[1]) + R1[2 * 3 + i] * dX[2];
  gn_proj(Y, e1, Jp1);
#pragma unroll
  for (int i = 0; i < 3; ++i) e1[i] = e1[i] - m1[i];
  double J1p[9], J1c[18];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      J1p[r * 3 + c] = (Jp1[r * 3 + 0] * R1[c * 3 + 0] + Jp1[r * 3 + 1] * R1[c * 3 + 1]) + Jp1[r * 3 + 2] * R1[c * 3 + 2];
  const double Yx[9] = {0.0, -Y[2], Y[1], Y[2], 0.0, -Y[0], -Y[1], Y[0], 0.0};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) J1c[r * 6 + c] = -Jp1[r * 3 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      J1c[r * 6 + 3 + c] = 2.0 * ((Jp1[r * 3 + 0] * Yx[0 * 3 + c] + Jp1[r * 3 + 1] * Yx[1 * 3 + c]) + Jp1[r * 3 + 2] * Yx[2 * 3 + c]);
  }
  const double w[3] = {1.0, 1.0, wz};
  double Hpp[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double s2 = 0.0, s1 = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) { s2 += (J2[r * 3 + a] * w[r]) * J2[r * 3 + b]; s1 += (J1p[r * 3 + a] * w[r]) * J1p[r * 3 + b]; }
      Hpp[a * 3 + b] = s2 + s1;
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double s2 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) { s2 += (J2[r * 3 + a] * w[r]) * e2[r]; s1 += (J1p[r * 3 + a] * w[r]) * e1[r]; }
    bp[a] = s2 + s1;
  }
  double Hcc[36], bc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) sacc += (J1c[r * 6 + a] * w[r]) * J1c[r * 6 + b];
      Hcc[a * 6 + b] = sacc;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < 3; ++r) sacc += (J1c[r * 6 + a] * w[r]) * J1p[r * 3 + b];
      Hcp[a * 3 + b] = sacc;
    }
    double sacc = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) sacc += (J1c[r * 6 + a] * w[r]) * e1[r];
    bc[a] = sacc;
  }
  {
    const double c00 = Hpp[4] * Hpp[8] - Hpp[5] * Hpp[7], c01 = Hpp[5] * Hpp[6] - Hpp[3] * Hpp[8], c02 = Hpp[3] * Hpp[7] - Hpp[4] * Hpp[6];
    const double det = (Hpp[0] * c00 + Hpp[1] * c01) + Hpp[2] * c02;
    const double id = 1.0 / det;
    Hpp_inv[0] = c00 * id; Hpp_inv[3] = c01 * id; Hpp_inv[6] = c02 * id;
    Hpp_inv[1] = (Hpp[2] * Hpp[7] - Hpp[1] * Hpp[8]) * id;
    Hpp_inv[4] = (Hpp[0] * Hpp[8] - Hpp[2] * Hpp[6]) * id;
    Hpp_inv[7] = (Hpp[1] * Hpp[6] - Hpp[0] * Hpp[7]) * id;
    Hpp_inv[2] = (Hpp[1] * Hpp[5] - Hpp[2] * Hpp[4]) * id;
    Hpp_inv[5] = (Hpp[2] * Hpp[3] - Hpp[0] * Hpp[5]) * id;
    Hpp_inv[8] = (Hpp[0] * Hpp[4] - Hpp[1] * Hpp[3]) * id;
  }
  double W[18];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      W[a * 3 + b] = (Hcp[a * 3 + 0] * Hpp_inv[0 * 3 + b] + Hcp[a * 3 + 1] * Hpp_inv[1 * 3 + b]) + Hcp[a * 3 + 2] * Hpp_inv[2 * 3 + b];
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = a; b < 6; ++b)
      acc[k++] += Hcc[a * 6 + b] - ((W[a * 3 + 0] * Hcp[b * 3 + 0] + W[a * 3 + 1] * Hcp[b * 3 + 1]) + W[a * 3 + 2] * Hcp[b * 3 + 2]);
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[21 + a] += bc[a] - ((W[a * 3 + 0] * bp[0] + W[a * 3 + 1] * bp[1]) + W[a * 3 + 2] * bp[2]);
}

// Eigen::Quaterniond(Matrix3d), normalised (g2o::SE3Quat), back to a rotation matrix: the start estimate of camera 1
__device__ __forceinline__ void gn_rot_via_quaternion(const double* Rin, double* Rout) {
  double q[4];
  const double t = Rin[0] + Rin[4] + Rin[8];
  if (t > 0.0) {
    double tt = sqrt(t + 1.0);
    q[3] = 0.5 * tt;
    tt = 0.5 / tt;
    q[0] = (Rin[7] - Rin[5]) * tt;
    q[1] = (Rin[2] - Rin[6]) * tt;
    q[2] = (Rin[3] - Rin[1]) * tt;
  } else {
    // the largest diagonal element picks (i, j, k); each case is written out with constant indices -- indexed by a run-time i
    // the nine doubles of Rin (and q) lived in an 80-byte private array in scratch
    const bool i1 = Rin[4] > Rin[0];
    const bool i2 = Rin[8] > (i1 ? Rin[4] : Rin[0]);
    auto pick = [&](auto I, auto J, auto K) {
      constexpr int i = decltype(I)::value, j = decltype(J)::value, k = decltype(K)::value;
      double tt = sqrt(Rin[i * 3 + i] - Rin[j * 3 + j] - Rin[k * 3 + k] + 1.0);
      q[i] = 0.5 * tt;
      tt = 0.5 / tt;
      q[3] = (Rin[k * 3 + j] - Rin[j * 3 + k]) * tt;
      q[j] = (Rin[j * 3 + i] + Rin[i * 3 + j]) * tt;
      q[k] = (Rin[k * 3 + i] + Rin[i * 3 + k]) * tt;
    };
    using std::integral_constant;
    if (i2) pick(integral_constant<int, 2>(), integral_constant<int, 0>(), integral_constant<int, 1>());
    else if (i1) pick(integral_constant<int, 1>(), integral_constant<int, 2>(), integral_constant<int, 0>());
    else pick(integral_constant<int, 0>(), integral_constant<int, 1>(), integral_constant<int, 2>());
  }
  const double nrm = sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] = q[i] / nrm;
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  Rout[0] = 1.0 - (tyy + tzz); Rout[1] = txy - twz;          Rout[2] = txz + twy;
  Rout[3] = txy + twz;          Rout[4] = 1.0 - (txx + tzz); Rout[5] = tyz - twx;
  Rout[6] = txz - twy;          Rout[7] = tyz + twx;          Rout[8] = 1.0 - (txx + tyy);
}

__device__ __forceinline__ double shfl_xor_f64(double v, int d) {
  return __hiloint2double(__shfl_xor(__double2hiint(v), d), __shfl_xor(__double2loint(v), d));
}

// getTransformFromMatchesG2O over the matches whose bits are set in `mask`; R, t (float, newer -> earlier) in and out
__device__ void gn_two_view(const uint64_t* mask, const RansacLds& lds, GnShared& gs, const rgbdfe_match_result* out,
                            const float2* __restrict__ qkp, const float2* __restrict__ tkp, int iterations, double wz,
                            float* R, float* tr) {
  const int lane = threadIdx.x;
  // the selected matches in match order
  int nsel = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const uint64_t pm = mask[r];
    if ((pm >> lane) & 1ull) gs.sel[nsel + (int)lane_rank(pm)] = (uint16_t)(r * kWave + lane);
    nsel += __popcll(pm);
  }
  wave_sync();
  for (int s = lane; s < nsel; s += kWave) {
    const int m = gs.sel[s];
    gs.kq[s] = qkp[out->all_q[m]];
    gs.kt[s] = tkp[out->all_t[m]];
    const float px = lds.M[m * kRec + 0], py = lds.M[m * kRec + 1], pz = lds.M[m * kRec + 2];
    if (!__builtin_isnan(pz)) {
      gs.X[s][0] = (double)px; gs.X[s][1] = (double)py; gs.X[s][2] = (double)pz;
    } else {  // :118
      gs.X[s][0] = (double)(px * 10); gs.X[s][1] = (double)(py * 10); gs.X[s][2] = 10.0;
    }
  }
  wave_sync();
  double R1[9], t1[3];
  {
    double Rin[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rin[i] = (double)R[i];
    gn_rot_via_quaternion(Rin, R1);
#pragma unroll
    for (int i = 0; i < 3; ++i) t1[i] = (double)tr[i];
  }
  auto measurements = [&](int s, double* m1, double* m2) {
    const int m = gs.sel[s];
    const float qz = lds.M[m * kRec + 2], tz = lds.M[m * kRec + 5];
    m1[0] = (double)gs.kt[s].x; m1[1] = (double)gs.kt[s].y; m1[2] = __builtin_isnan(tz) ? 10.0 : (double)tz;
    m2[0] = (double)gs.kq[s].x; m2[1] = (double)gs.kq[s].y; m2[2] = __builtin_isnan(qz) ? 10.0 : (double)qz;
  };
  bool ok = true;
  for (int it = 0; it < iterations && ok; ++it) {
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    double Hinv[9], bp[3], Hcp[18];
    for (int s = lane; s < nsel; s += kWave) {
      double m1[3], m2[3];
      measurements(s, m1, m2);
      const double X[3] = {gs.X[s][0], gs.X[s][1], gs.X[s][2]};
      gn_match_terms(X, R1, t1, m1, m2, wz, acc, Hinv, bp, Hcp);
    }
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1)
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = acc[k] + shfl_xor_f64(acc[k], d);
    // S dc = -g by Cholesky, redundantly in every lane (all lanes hold the same sums)
    double S[36], L[36], y[6], dc[6];
    {
      int k = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) { S[a * 6 + b] = acc[k]; S[b * 6 + a] = acc[k]; ++k; }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = S[j * 6 + j];
#pragma unroll
      for (int kk = 0; kk < 6; ++kk)
        if (kk < j) d -= L[j * 6 + kk] * L[j * 6 + kk];
      if (!(d > 0.0)) ok = false;
      L[j * 6 + j] = sqrt(d);
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i > j) {
          double v = S[i * 6 + j];
#pragma unroll
          for (int kk = 0; kk < 6; ++kk)
            if (kk < j) v -= L[i * 6 + kk] * L[j * 6 + kk];
          L[i * 6 + j] = v / L[j * 6 + j];
        }
    }
    if (!ok) break;  // the linear solver failed: the optimiser stops (wave-uniform)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double v = -acc[21 + i];
#pragma unroll
      for (int kk = 0; kk < 6; ++kk)
        if (kk < i) v -= L[i * 6 + kk] * y[kk];
      y[i] = v / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double v = y[i];
#pragma unroll
      for (int kk = 0; kk < 6; ++kk)
        if (kk > i) v -= L[kk * 6 + i] * dc[kk];
      dc[i] = v / L[i * 6 + i];
    }
    // the points, from the state before the update
    for (int s = lane; s < nsel; s += kWave) {
      double m1[3], m2[3], dummy[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) dummy[k] = 0.0;
      measurements(s, m1, m2);
      const double X[3] = {gs.X[s][0], gs.X[s][1], gs.X[s][2]};
      gn_match_terms(X, R1, t1, m1, m2, wz, dummy, Hinv, bp, Hcp);
      double rr[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        double v = bp[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) v += Hcp[a * 3 + b] * dc[a];
        rr[b] = v;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const double dp = -((Hinv[a * 3 + 0] * rr[0] + Hinv[a * 3 + 1] * rr[1]) + Hinv[a * 3 + 2] * rr[2]);
        gs.X[s][a] = X[a] + dp;
      }
    }
    // the pose: estimate = estimate * (dt, q(dq))
    {
      double Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      const double qx = dc[3], qy = dc[4], qz = dc[5];
      double ww = 1.0 - ((qx * qx + qy * qy) + qz * qz);
      if (!(ww < 0.0)) {
        ww = sqrt(ww);
        const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
        const double twx = tx * ww, twy = ty * ww, twz = tz * ww;
        const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
        const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        Rd[0] = 1.0 - (tyy + tzz); Rd[1] = txy - twz;          Rd[2] = txz + twy;
        Rd[3] = txy + twz;          Rd[4] = 1.0 - (txx + tzz); Rd[5] = tyz - twx;
        Rd[6] = txz - twy;          Rd[7] = tyz + twx;          Rd[8] = 1.0 - (txx + tyy);
      }
      double tn[3], Rn[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) tn[r] = t1[r] + ((R1[r * 3 + 0] * dc[0] + R1[r * 3 + 1] * dc[1]) + R1[r * 3 + 2] * dc[2]);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          Rn[r * 3 + c] = (R1[r * 3 + 0] * Rd[0 * 3 + c] + R1[r * 3 + 1] * Rd[1 * 3 + c]) + R1[r * 3 + 2] * Rd[2 * 3 + c];
#pragma unroll
      for (int i = 0; i < 9; ++i) R1[i] = Rn[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) t1[i] = tn[i];
    }
    wave_sync();
  }
  // estimate.cast<float>().inverse() (:169)
  float Rf[9], tf[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rf[i] = (float)R1[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) tf[i] = (float)t1[i];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rf[c * 3 + r];
    const float v = (Rf[0 * 3 + r] * tf[0] + Rf[1 * 3 + r] * tf[1]) + Rf[2 * 3 + r] * tf[2];
    tr[r] = -v;
  }
  wave_sync();
}

__global__ __launch_bounds__(kWave) void g2o_refine_kernel(const PairWork* __restrict__ work,
                                                           rgbdfe_match_result* __restrict__ results, uint32_t n_pairs,
                                                           const RansacConst rc, const PairPrep* __restrict__ prep,
                                                           const float2* __restrict__ kp_pool, uint32_t max_kp,
                                                           double* __restrict__ ec_pool) {
  __shared__ RansacLds lds;
  __shared__ GnShared gs;
  const uint32_t pair = blockIdx.x;
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x;
  const PairWork w = work[pair];
  rgbdfe_match_result* __restrict__ out = results + pair;
  const PairPrep* __restrict__ pp = prep + pair;
  const int n_all = pp->n_all;
  if (!(n_all > rc.min_matches)) return;  // getRelativeTransformationTo returned at :1087 (or was never called, :1319)
  PH_DECL
  uint32_t thr = (uint32_t)rc.min_matches;
  if ((double)thr > 0.75 * (double)n_all) thr = (uint32_t)(0.75 * (double)n_all);
  int n_matches = out->n_inl;
  if (!((uint32_t)n_matches > thr)) return;  // :1226
  double* __restrict__ ec_region = ec_pool + (size_t)blockIdx.x * kEcRegion;
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(pp->M);
    float4* __restrict__ dst = reinterpret_cast<float4*>(lds.M);
    constexpr int kVec = RGBDFE_MAX_MATCHES * kRec / 4;
    for (int v = lane; v < kVec; v += kWave) dst[v] = src[v];
    wave_sync();
  }
  const float pmax = pp->pmax;
  const float2* __restrict__ qkp = kp_pool + (size_t)w.q_slot * max_kp;
  const float2* __restrict__ tkp = kp_pool + (size_t)w.t_slot * max_kp;
  const double wz = 1.0 / rc.depth_cov;  // point_information_matrix (misc2.h:44), depth_covariance frozen (D3)
  float R[9], tr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = out->trafo[j * 4 + i];
    tr[i] = out->trafo[12 + i];
  }
  uint64_t mask[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) mask[r] = out->inlier_mask[r];
  float rmse = out->rmse;
  gn_two_view(mask, lds, gs, out, qkp, tkp, rc.g2o_iterations, wz, R, tr);  // :1229
  uint64_t inl[kRounds];
  int n_inl;
  double inlier_error;
  score_hypothesis(R, tr, n_all, 0u, rc, lds, pmax, ec_region, inl, n_inl, inlier_error PH_PASS);  // :1233
  bool adopt = false;
  if (n_inl >= n_matches || ((uint32_t)n_inl >= thr && inlier_error < (double)rmse)) {  // :1239
    if (n_inl > n_matches) {                                                             // :1241
      gn_two_view(inl, lds, gs, out, qkp, tkp, rc.g2o_iterations, wz, R, tr);             // :1243
      score_hypothesis(R, tr, n_all, 0u, rc, lds, pmax, ec_region, inl, n_inl, inlier_error PH_PASS);  // :1244
    }
    adopt = n_inl >= n_matches;  // :1252
  }
  if (adopt && lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) out->trafo[j * 4 + i] = R[i * 3 + j];
      out->trafo[12 + i] = tr[i];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) out->inlier_mask[r] = inl[r];
    out->n_inl = n_inl;
    const float new_rmse = (float)inlier_error;  // :1258
    out->rmse = new_rmse;
    out->valid_iterations = out->valid_iterations + 1;  // :1259
    // found stays true (n_inl >= n_matches > thr); the edge's information follows the new numbers (node.cpp:1335)
    out->info_scale = (double)((float)n_inl / (new_rmse * new_rmse));
  }
}

void launch_g2o_refine(const PairWork* work, rgbdfe_match_result* results, uint32_t n_pairs, const RansacConst& rc,
                       const PairPrep* prep, const float* kp_pool, uint32_t max_kp, double* ec_pool, hipStream_t stream) {
  if (n_pairs == 0 || rc.g2o_iterations <= 0) return;
  hipLaunchKernelGGL(g2o_refine_kernel, dim3(n_pairs), dim3(kWave), 0, stream, work, results, n_pairs, rc, prep,
                     reinterpret_cast<const float2*>(kp_pool), max_kp, ec_pool);
}

size_t select_ransac_ec_region_bytes() { return sizeof(double) * (size_t)kEcRegion; }

}  // namespace rgbdfe

#ifdef RGBDFE_PROFILE_PHASES
// diagnostics build only (librgbdfe_prof.so): wall cycles per phase summed over the recording waves
extern "C" int rgbdfe_debug_phase_totals(unsigned long long* out20, int reset) {
  unsigned long long zero[24] = {0};
  if (out20 && hipMemcpyFromSymbol(out20, HIP_SYMBOL(rgbdfe::g_phase_totals), sizeof(zero)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(rgbdfe::g_phase_totals), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
#endif
