// ransac_device.h -- device-side arithmetic shared by the RANSAC kernels (select_ransac.hip: one wave per pair,
// result waves, g2o refinement; ransac_split.hip: the hypothesis and refinement kernels of the record / replay schedule).
// Every function follows the operation order of oracle/rgbd_oracle.c (which restates the reference and is pinned on
// the reference's own compiled code): with -ffp-contract=off the results are bit-identical to the CPU restatement.
#pragma once
#include <float.h>
#include <stddef.h>

#include <type_traits>

#include "rgbdfe_internal.h"

namespace rgbdfe {

constexpr int kWave = 64;
constexpr int kRounds = RGBDFE_MAX_MATCHES / kWave;  // 5

// RANSAC iterations refined side by side: the refits of a round share ONE recurrence loop (9 lanes per
// slot, 7 x 9 = 63 lanes) and ONE batched SVD (lane = slot).
constexpr int kSlots = 7;
constexpr int kFitUnroll = 4;   // recurrence steps per trip; loads run two trips ahead
// bytes per slot list: 320 entries + two trips of read-ahead, an odd number of words so that the
// slots' k-th entries sit in different LDS banks
constexpr int kOrdStride = RGBDFE_MAX_MATCHES + 2 * kFitUnroll + 4;
static_assert(kOrdStride % 4 == 0 && (kOrdStride / 4) % 2 == 1, "list rows: word aligned, odd word count");
static_assert(RGBDFE_MAX_MATCHES <= 512, "list entries keep the low 8 bits of a match index + one threshold");
// one match in LDS: from.xyz, to.xyz, weight (7 words: a lane = match access is bank-conflict free)
constexpr int kRec = 7;
constexpr uint32_t kRecBytes = kRec * 4;

// refit phase: the inlier set compacted in match order
struct FitBuf {
  // per slot: its participating matches in match order, low 8 bits of the match index (the lists ascend, so
  // "index >= 256" is one threshold position per list, kept in a register): 1/2 of the u16 footprint, which
  // is what lets 12 instead of 10 waves share a CU's LDS
  uint8_t ord[kSlots][kOrdStride];
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// D1: counter-based replacement of rand() (node.cpp:1033-1034); same integer function as
// the oracle's orc_rand31.
__device__ __forceinline__ uint32_t rand31(uint32_t seed_mixed_uid, uint32_t iter, uint32_t k) {
  uint32_t h = mix32(seed_mixed_uid ^ (iter * 0xC2B2AE35u + 0x165667B1u));
  h = mix32(h + k * 0x27D4EB2Fu);
  return h >> 1;
}

__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {
  // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ float bcast_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---------------------------------------------------------------------------------
// pcl::TransformationFromCorrespondences accumulator (float, sequential recurrence)
// ---------------------------------------------------------------------------------
struct Tfc {
  float W;
  float m1[3], m2[3];
  float C[9];  // row-major C[i*3+j]
  __device__ __forceinline__ void reset() {
    W = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) m1[i] = m2[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = 0.0f;
  }
  // transformation_estimation_euclidean.cpp:20-25,56 + tfc.add()
  __device__ __forceinline__ void add(const float* __restrict__ M, int m) {
    float f[3] = {M[m * kRec + 0], M[m * kRec + 1], M[m * kRec + 2]};
    float t[3] = {M[m * kRec + 3], M[m * kRec + 4], M[m * kRec + 5]};
    if (__builtin_isnan(f[2]) || __builtin_isnan(t[2])) return;
    // weight = 1.0/(from(2)*to(2)): double divide rounded to float == float divide
    // (53 >= 2*24+2: double rounding is innocuous for division)
    float w = 1.0f / (f[2] * t[2]);
    if (w == 0.0f) return;
    W += w;
    float alpha = w / W;
    float d1[3], d2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) d1[j] = f[j] - m1[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) d2[i] = t[i] - m2[i];
    float oma = 1.0f - alpha;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float outer = d2[i] * d1[j];
        float scaled = alpha * outer;
        float sum = C[i * 3 + j] + scaled;
        C[i * 3 + j] = oma * sum;
      }
#pragma unroll
    for (int j = 0; j < 3; ++j) m1[j] = m1[j] + alpha * d1[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) m2[i] = m2[i] + alpha * d2[i];
  }
};

// ---------------------------------------------------------------------------------
// 3x3 two-sided Jacobi SVD (Eigen::JacobiSVD<Matrix3f> as published), row-major.
// Same operation order as the oracle's orc_svd3.
// ---------------------------------------------------------------------------------
template <int p, int q>
__device__ __forceinline__ bool jacobi_pair(float* W, float* U, float* V, float& max_diag) {
  const float precision = 2.0f * FLT_EPSILON;
  float threshold = precision * max_diag;
  if (FLT_MIN > threshold) threshold = FLT_MIN;
  if (!(fabsf(W[p * 3 + q]) > threshold || fabsf(W[q * 3 + p]) > threshold)) return false;
  float m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
  float t = m00 + m11;
  float d = m10 - m01;
  float c1, s1;
  if (fabsf(d) < FLT_MIN) {
    c1 = 1.0f; s1 = 0.0f;
  } else {
    float u = t / d;
    float tmp = sqrtf(1.0f + u * u);
    s1 = 1.0f / tmp;
    c1 = u / tmp;
  }
  float n00 = c1 * m00 + s1 * m10;
  float n01 = c1 * m01 + s1 * m11;
  float n11 = (-s1) * m01 + c1 * m11;
  float cr, sr;
  float deno = 2.0f * fabsf(n01);
  if (deno < FLT_MIN) {
    cr = 1.0f; sr = 0.0f;
  } else {
    float tau = (n00 - n11) / deno;
    float w = sqrtf(tau * tau + 1.0f);
    float tt = (tau > 0.0f) ? 1.0f / (tau + w) : 1.0f / (tau - w);
    float sign_t = (tt > 0.0f) ? 1.0f : -1.0f;
    float nn = 1.0f / sqrtf(tt * tt + 1.0f);
    sr = -sign_t * (n01 / fabsf(n01)) * fabsf(tt) * nn;
    cr = nn;
  }
  float cl = c1 * cr + s1 * sr;
  float sl = s1 * cr - c1 * sr;
  // Each rotation updates a pair (x, y) from its own old values.  Written so that the results can land in the registers
  // of x and y themselves (the lanes that skip this pair keep theirs): the four products first -- the last one into y --
  // then the two sums; as two assignments of full expressions the compiler computed into temporaries and copied.
  const float nsl = -sl;
  auto rot_l = [&](float& x, float& y) {   // x' = cl x + sl y, y' = (-sl) x + cl y
    const float a = cl * x, b = sl * y, c = nsl * x;
    y = cl * y;
    y = c + y;
    x = a + b;
  };
  auto rot_r = [&](float& x, float& y) {   // x' = cr x - sr y, y' = sr x + cr y
    const float a = cr * x, b = sr * y, c = sr * x;
    y = cr * y;
    y = c + y;
    x = a - b;
  };
#pragma unroll
  for (int k = 0; k < 3; ++k) rot_l(W[p * 3 + k], W[q * 3 + k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) rot_l(U[k * 3 + p], U[k * 3 + q]);
#pragma unroll
  for (int k = 0; k < 3; ++k) rot_r(W[k * 3 + p], W[k * 3 + q]);
#pragma unroll
  for (int k = 0; k < 3; ++k) rot_r(V[k * 3 + p], V[k * 3 + q]);
  float a = fabsf(W[p * 3 + p]), b = fabsf(W[q * 3 + q]);
  if (b > a) a = b;
  if (a > max_diag) max_diag = a;
  return true;
}

template <int a, int b>
__device__ __forceinline__ void swap_cols(float* S, float* U, float* V) {
  float ts = S[a]; S[a] = S[b]; S[b] = ts;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float tu = U[k * 3 + a]; U[k * 3 + a] = U[k * 3 + b]; U[k * 3 + b] = tu;
    float tv = V[k * 3 + a]; V[k * 3 + a] = V[k * 3 + b]; V[k * 3 + b] = tv;
  }
}

__device__ __forceinline__ float det3(const float* m) {
  float h0 = m[0] * (m[4] * m[8] - m[5] * m[7]);
  float h1 = m[1] * (m[3] * m[8] - m[5] * m[6]);
  float h2 = m[2] * (m[3] * m[7] - m[4] * m[6]);
  return h0 - h1 + h2;
}

// tfc.getTransformation(): R (row-major 9) and t (3)
__device__ __forceinline__ void tfc_get_transformation(const Tfc& s, float* R, float* tr) {
  float W[9], U[9], V[9], S[3];
  float scale = 0.0f;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    float a = fabsf(s.C[i]);
    if (a > scale) scale = a;
  }
  if (scale == 0.0f) scale = 1.0f;
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = s.C[i] / scale;
#pragma unroll
  for (int i = 0; i < 9; ++i) U[i] = V[i] = (i % 4 == 0) ? 1.0f : 0.0f;
  float max_diag = fabsf(W[0]);
  if (fabsf(W[4]) > max_diag) max_diag = fabsf(W[4]);
  if (fabsf(W[8]) > max_diag) max_diag = fabsf(W[8]);
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool any = false;
    any |= jacobi_pair<1, 0>(W, U, V, max_diag);
    any |= jacobi_pair<2, 0>(W, U, V, max_diag);
    any |= jacobi_pair<2, 1>(W, U, V, max_diag);
    if (!any) break;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float w = W[i * 3 + i];
    S[i] = fabsf(w);
    if (w < 0.0f) {
#pragma unroll
      for (int k = 0; k < 3; ++k) U[k * 3 + i] = -U[k * 3 + i];
    }
    S[i] = S[i] * scale;
  }
  // selection sort, descending, first maximum wins; stop at an all-zero tail
  {
    int pos = 0;
    float best = S[0];
    if (S[1] > best) { best = S[1]; pos = 1; }
    if (S[2] > best) { best = S[2]; pos = 2; }
    if (best != 0.0f) {
      if (pos == 1) swap_cols<0, 1>(S, U, V);
      if (pos == 2) swap_cols<0, 2>(S, U, V);
      if (S[2] > S[1]) {  // i = 1: best = S[2] != 0 here since S[2] > S[1] >= 0
        swap_cols<1, 2>(S, U, V);
      }
    }
  }
  float s22 = 1.0f;
  if (det3(U) * det3(V) < 0.0f) s22 = -1.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float us2 = U[i * 3 + 2] * s22;
      R[i * 3 + j] = (U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1]) + us2 * V[j * 3 + 2];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float rm = (R[i * 3 + 0] * s.m1[0] + R[i * 3 + 1] * s.m1[1]) + R[i * 3 + 2] * s.m1[2];
    tr[i] = s.m2[i] - rm;
  }
}

__device__ __forceinline__ bool has_nan12(const float* R, const float* t) {
  bool n = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) n |= (R[i] != R[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) n |= (t[i] != t[i]);
  return n;
}

// ---------------------------------------------------------------------------------
// d^T S^-1 d through the unblocked Cholesky factorisation and the two triangular solves of
// Eigen's llt().solve() (misc.cpp:763); S given by its lower triangle.  Same operation order as
// the oracle's orc_error_function2; `ok` = all pivots positive.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double mahal_llt_ieee(double S00, double S10, double S11, double S20, double S21,
                                                 double S22, const double* d, bool& ok) {
  ok = S00 > 0.0;
  const double l00 = sqrt(S00);
  const double l10 = S10 / l00;
  const double l20 = S20 / l00;
  const double x1 = S11 - l10 * l10;
  ok = ok && (x1 > 0.0);
  const double l11 = sqrt(x1);
  const double l21 = (S21 - l20 * l10) / l11;
  const double x2 = S22 - (l20 * l20 + l21 * l21);
  ok = ok && (x2 > 0.0);
  const double l22 = sqrt(x2);
  const double y0 = d[0] / l00;
  const double y1 = (d[1] - l10 * y0) / l11;
  const double y2 = (d[2] - (l20 * y0 + l21 * y1)) / l22;
  const double z2 = y2 / l22;
  const double z1 = (y1 - l21 * z2) / l11;
  const double z0 = (y0 - (l10 * z1 + l20 * z2)) / l00;
  return (d[0] * z0 + d[1] * z1) + d[2] * z2;
}

// The same arithmetic with the IEEE divisions and square roots spelled out as the gfx950 expansion of
// `/` and `sqrt` WITHOUT its range scaling (v_div_scale / v_div_fmas / v_div_fixup, v_ldexp), and with
// the refined reciprocal of a pivot shared by the 2-4 divisions that use it: 9 divisions cost
// 3 x 5 + 9 x 3 instructions instead of 9 x 11.  Bit-identical to the expansion whenever the scaling
// would have been the identity, i.e. every numerator and radicand is a normal number with a binary
// exponent within +-200 (no zero, denormal, inf, NaN).  `unsafe` reports a lane outside that window;
// the caller then recomputes with mahal_llt_ieee.
struct ExpWindow {
  uint32_t lo = 0x7FF00000u, hi = 0u;
  __device__ __forceinline__ void see(double v) {
    const uint32_t e = (uint32_t)__double2hiint(v) & 0x7FF00000u;
    lo = min(lo, e);
    hi = max(hi, e);
  }
  __device__ __forceinline__ bool outside() const {
    return lo < ((1023u - 200u) << 20) || hi > ((1023u + 200u) << 20);
  }
};
__device__ __forceinline__ double rcp_refined(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_by(double a, double b, double rb) {
  const double q = a * rb;
  const double res = __builtin_fma(-b, q, a);
  return __builtin_fma(res, rb, q);
}
__device__ __forceinline__ double sqrt_unscaled(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double t = __builtin_fma(-g, g, x);
  g = __builtin_fma(t, h, g);
  t = __builtin_fma(-g, g, x);
  g = __builtin_fma(t, h, g);
  return g;
}
__device__ __forceinline__ double mahal_llt_fast(double S00, double S10, double S11, double S20, double S21,
                                                 double S22, const double* d, bool& ok, bool& unsafe) {
  ExpWindow win;
  ok = S00 > 0.0;
  win.see(S00);
  const double l00 = sqrt_unscaled(S00);
  const double r00 = rcp_refined(l00);
  win.see(S10); win.see(S20); win.see(d[0]);
  const double l10 = div_by(S10, l00, r00);
  const double l20 = div_by(S20, l00, r00);
  const double y0 = div_by(d[0], l00, r00);
  const double x1 = S11 - l10 * l10;
  ok = ok && (x1 > 0.0);
  win.see(x1);
  const double l11 = sqrt_unscaled(x1);
  const double r11 = rcp_refined(l11);
  const double n21 = S21 - l20 * l10;
  const double ny1 = d[1] - l10 * y0;
  win.see(n21); win.see(ny1);
  const double l21 = div_by(n21, l11, r11);
  const double y1 = div_by(ny1, l11, r11);
  const double x2 = S22 - (l20 * l20 + l21 * l21);
  ok = ok && (x2 > 0.0);
  win.see(x2);
  const double l22 = sqrt_unscaled(x2);
  const double r22 = rcp_refined(l22);
  const double ny2 = d[2] - (l20 * y0 + l21 * y1);
  win.see(ny2);
  const double y2 = div_by(ny2, l22, r22);
  win.see(y2);
  const double z2 = div_by(y2, l22, r22);
  const double nz1 = y1 - l21 * z2;
  win.see(nz1);
  const double z1 = div_by(nz1, l11, r11);
  const double nz0 = y0 - (l10 * z1 + l20 * z2);
  win.see(nz0);
  const double z0 = div_by(nz0, l00, r00);
  unsafe = win.outside();
  return (d[0] * z0 + d[1] * z1) + d[2] * z2;
}

__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------
// getTransformFromMatches for the active slots of a window round, all at once.
// The PCL recurrence (Tfc::add) is strictly sequential in its state, so one refit can keep only
// 9 lanes busy (lane l: C[i][j], mean2[i], mean1[j], i = l/3, j = l%3).  The refits of different
// slots are independent: slot s runs on lanes 9s .. 9s+8, each walking its own compacted inlier list.
//   fit_compact     (lane = match): k-th participating match of slot s -> ord[s][k]
//   fit_recurrence  (lane = slot x element): W += w; alpha = w / W; the 15 state elements advance
// Every float operation is the one the sequential code performs, in the same order on the same
// operands: bit-identical to Tfc::add over the same matches.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int fit_compact(int s, const uint64_t* mask, const uint64_t* w_nonzero, FitBuf& fit,
                                           int& n_below_256, int lane) {
  uint32_t base = 0;
  n_below_256 = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    // tfc.add skips weight == 0; NaN depths never reach an inlier set (misc.cpp:712-717)
    const uint64_t pm = mask[r] & w_nonzero[r];  // wave-uniform: scalar ALU
    // lane's bit of the wave-uniform mask as the execution mask itself (s_and_saveexec), no per-lane shift
    if (__builtin_amdgcn_inverse_ballot_w64(pm)) fit.ord[s][base + lane_rank(pm)] = (uint8_t)(r * kWave + lane);
    base += (uint32_t)__popcll(pm);
    if (r == 256 / kWave - 1) n_below_256 = (int)base;
  }
  // the recurrence reads two trips past the end of a list: keep those entries valid match indices
  if (lane < 2 * kFitUnroll) fit.ord[s][base + lane] = 0;
  return (int)base;
}

// Keeps v in a register at this point: the compiler may not sink the computation of v into a
// conditionally executed block (which would serialise the LDS loads feeding it again).
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// n_mine: list length of this lane's slot (0: nothing to do), n_min / n_max: the shortest / longest list among the
// slots that take part in the round.
// On return lane 9s+x holds C[x], lane 9s+j mean1[j], lane 9s+3i mean2[i] of slot s.
// Software pipeline per trip of kFitUnroll steps: list entries are read two trips ahead, the match
// records one trip ahead; every load is unconditional (lists are padded two trips past their end with a
// valid index) and a finished slot keeps its state through selects.
// FAST_DIV: alpha = w / W through the gfx950 expansion of the f32 division without its range scaling
// (v_div_scale / v_div_fmas / v_div_fixup): bit-identical when the scaling is the identity, which the
// caller guarantees by checking once per pair that every weight lies in [2^-40, 2^40] (W is a sum of at
// most 320 of them).
template <bool FAST_DIV>
__device__ __forceinline__ void fit_recurrence(int n_mine, int k256_mine, int n_min, int n_max, const FitBuf& fit,
                                               const float* __restrict__ M, float& C, float& m1, float& m2, int lane) {
  constexpr int U = kFitUnroll;
  const int sl = min(lane / 9, kSlots - 1);
  const int l9 = lane % 9;
  const int ci = l9 / 3, cj = l9 % 3;
  const uint8_t* __restrict__ ord = fit.ord[sl];
  // byte offset of the k-th list entry's record: (low 8 bits + 256 from position k256 on) * 28
  auto rec_off = [&](int k) {
    const uint32_t hi = (k >= k256_mine) ? 256u * kRecBytes : 0u;
    return (uint32_t)ord[k] * kRecBytes + hi;
  };
  const char* __restrict__ recs = reinterpret_cast<const char*>(M);
  const char* __restrict__ recP = recs + cj * 4;        // from[cj]
  const char* __restrict__ recQ = recs + 12 + ci * 4;   // to[ci]
  float W = 0.0f;
  C = 0.0f; m1 = 0.0f; m2 = 0.0f;
  uint32_t off[U];
  float wv[U], fv[U], tv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) off[u] = rec_off(u);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    wv[u] = *reinterpret_cast<const float*>(recs + off[u] + 24);
    fv[u] = *reinterpret_cast<const float*>(recP + off[u]);
    tv[u] = *reinterpret_cast<const float*>(recQ + off[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) off[u] = rec_off(U + u);
  auto trip = [&](int k0, auto with_select) {
    float wc[U], fc[U], tc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { wc[u] = wv[u]; fc[u] = fv[u]; tc[u] = tv[u]; }
    // records of the next trip, list entries of the trip after it
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wv[u] = *reinterpret_cast<const float*>(recs + off[u] + 24);
      fv[u] = *reinterpret_cast<const float*>(recP + off[u]);
      tv[u] = *reinterpret_cast<const float*>(recQ + off[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) off[u] = rec_off(k0 + 2 * U + u);
    // W_k = W_{k-1} + w_k, alpha_k = w_k / W_k: off the state's dependence chain
    float al[U], om[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      W = W + wc[u];
      if (FAST_DIV) {
        float r = __builtin_amdgcn_rcpf(W);
        const float e = __builtin_fmaf(-W, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        float q = wc[u] * r;
        float res = __builtin_fmaf(-W, q, wc[u]);
        q = __builtin_fmaf(res, r, q);
        res = __builtin_fmaf(-W, q, wc[u]);
        al[u] = __builtin_fmaf(res, r, q);
      } else {
        al[u] = wc[u] / W;
      }
      om[u] = 1.0f - al[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool act = (k0 + u) < n_mine;
      const float d1 = fc[u] - m1;
      const float d2 = tc[u] - m2;
      const float outer = d2 * d1;
      const float scaled = al[u] * outer;
      const float sum = C + scaled;
      float Cn = om[u] * sum;
      float m1n = m1 + al[u] * d1;
      float m2n = m2 + al[u] * d2;
      if (decltype(with_select)::value) {  // a slot whose list has ended keeps its state
        pin(Cn); pin(m1n); pin(m2n);
        C = act ? Cn : C;
        m1 = act ? m1n : m1;
        m2 = act ? m2n : m2;
      } else {
        C = Cn; m1 = m1n; m2 = m2n;
      }
    }
  };
  // While every participating slot still has entries (k < n_min) no lane needs the selects; lanes of slots that
  // do not take part in this round (n_mine == 0) compute garbage that nobody reads.
  int k0 = 0;
  for (; k0 + U <= n_min; k0 += U) trip(k0, std::false_type());
  for (; k0 < n_max; k0 += U) trip(k0, std::true_type());
}

// Upper bound of the number of matches that can pass errorFunction2's shortcut test (misc.cpp:726-735) under THIS LANE's
// hypothesis (lane = hypothesis): the float evaluation of dsq with score_passes' error band -- a match counts unless its
// dsq_f is provably above the threshold, NaN counts.  Every lane walks all matches; the match record is the same LDS
// address for all lanes (a broadcast read).
__device__ __forceinline__ uint32_t prescreen_may_pass(const float* hypR, const float* hypt, const float* __restrict__ M,
                                                    int n_all, float pmax, const RansacConst& rc) {
  const float u4 = 4.0f * 5.9604645e-8f;
  float es = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    es += u4 * (((fabsf(hypR[3 * i]) + fabsf(hypR[3 * i + 1])) + fabsf(hypR[3 * i + 2]) + 1.0f) * pmax + fabsf(hypt[i]));
  const double smax = rc.raster_cov_x > rc.depth_cov ? rc.raster_cov_x : rc.depth_cov;
  const float S = (float)(2.0 * (smax + smax));
  const float E = 2.0f * ((2.0f * sqrtf(S) * 1.001f) * es + es * es + u4 * S) + 1e-30f;
  const float hi_f = S * 1.000001f + E;
  uint32_t may_pass = 0;
  // Four matches per trip (their 24 LDS reads are in flight together), evaluated two at a time with packed f32
  // arithmetic (v_pk_fma_f32 / v_pk_add_f32: the hypothesis' coefficients feed both halves).  No validity test per
  // match: the count is an UPPER bound, and a match without depth (zero or NaN z; none survive removeDepthless in
  // practice) or one of the up to three records behind n_all can only add to it -- the outcome of the iteration is the
  // scoring's either way, the pre-screen merely fails to skip it.  The records behind n_all exist: PairPrep holds
  // RGBDFE_MAX_MATCHES of them, a multiple of 4.
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f R2[9], t2[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R2[i] = v2f{hypR[i], hypR[i]};
#pragma unroll
  for (int i = 0; i < 3; ++i) t2[i] = v2f{hypt[i], hypt[i]};
#pragma unroll 1
  for (int m0 = 0; m0 < n_all; m0 += 4) {
    float rec[4][6];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 6; ++c) rec[u][c] = M[(m0 + u) * kRec + c];
#pragma unroll
    for (int u = 0; u < 4; u += 2) {
      const v2f px = {rec[u][0], rec[u + 1][0]}, py = {rec[u][1], rec[u + 1][1]}, pz = {rec[u][2], rec[u + 1][2]};
      const v2f qx = {rec[u][3], rec[u + 1][3]}, qy = {rec[u][4], rec[u + 1][4]}, qz = {rec[u][5], rec[u + 1][5]};
      const v2f f0 = __builtin_elementwise_fma(R2[0], px, __builtin_elementwise_fma(R2[1], py, __builtin_elementwise_fma(R2[2], pz, t2[0]))) - qx;
      const v2f f1 = __builtin_elementwise_fma(R2[3], px, __builtin_elementwise_fma(R2[4], py, __builtin_elementwise_fma(R2[5], pz, t2[1]))) - qy;
      const v2f f2 = __builtin_elementwise_fma(R2[6], px, __builtin_elementwise_fma(R2[7], py, __builtin_elementwise_fma(R2[8], pz, t2[2]))) - qz;
      const v2f dsq = __builtin_elementwise_fma(f0, f0, __builtin_elementwise_fma(f1, f1, f2 * f2));
      may_pass += !(dsq.x > hi_f) ? 1u : 0u;
      may_pass += !(dsq.y > hi_f) ? 1u : 0u;
    }
  }
  return may_pass;
}

// The same count from a transposed copy of the records: blocks of four matches, component-major -- S4[block][component 0..5]
// [match 0..3] -- so that one 16-byte LDS read (the same address in every lane: a broadcast) brings one component of four
// matches, and the halves of its four registers ARE the operand pairs of the packed arithmetic: 6 reads and no register
// shuffling per four matches, against 24 reads and 19 moves from the 7-word records.  Same pairs, same operations, same
// count.  (The copy costs a workgroup 7 LDS writes per thread once.)
constexpr int kS4Block = 24;   // floats of a block of four matches
__device__ __forceinline__ void transpose_records(const float* __restrict__ M, float* __restrict__ S4, int tid, int n_threads) {
  for (int v = tid; v < (RGBDFE_MAX_MATCHES / 4) * kS4Block; v += n_threads) {
    const int blk = v / kS4Block, r = v - blk * kS4Block;
    S4[v] = M[(blk * 4 + (r & 3)) * kRec + (r >> 2)];
  }
}
__device__ __forceinline__ uint32_t prescreen_may_pass_s4(const float* hypR, const float* hypt, const float* __restrict__ S4,
                                                       int n_all, float pmax, const RansacConst& rc) {
  const float u4 = 4.0f * 5.9604645e-8f;
  float es = 0.0f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    es += u4 * (((fabsf(hypR[3 * i]) + fabsf(hypR[3 * i + 1])) + fabsf(hypR[3 * i + 2]) + 1.0f) * pmax + fabsf(hypt[i]));
  const double smax = rc.raster_cov_x > rc.depth_cov ? rc.raster_cov_x : rc.depth_cov;
  const float S = (float)(2.0 * (smax + smax));
  const float E = 2.0f * ((2.0f * sqrtf(S) * 1.001f) * es + es * es + u4 * S) + 1e-30f;
  const float hi_f = S * 1.000001f + E;
  uint32_t may_pass = 0;
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  v2f R2[9], t2[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R2[i] = v2f{hypR[i], hypR[i]};
#pragma unroll
  for (int i = 0; i < 3; ++i) t2[i] = v2f{hypt[i], hypt[i]};
  const v4f* __restrict__ blocks = reinterpret_cast<const v4f*>(S4);
#pragma unroll 1
  for (int m0 = 0; m0 < n_all; m0 += 4) {
    const v4f* __restrict__ b = blocks + (m0 >> 2) * 6;
    const v4f px = b[0], py = b[1], pz = b[2], qx = b[3], qy = b[4], qz = b[5];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f x = h ? px.zw : px.xy, y = h ? py.zw : py.xy, z = h ? pz.zw : pz.xy;
      const v2f f0 = __builtin_elementwise_fma(R2[0], x, __builtin_elementwise_fma(R2[1], y, __builtin_elementwise_fma(R2[2], z, t2[0]))) - (h ? qx.zw : qx.xy);
      const v2f f1 = __builtin_elementwise_fma(R2[3], x, __builtin_elementwise_fma(R2[4], y, __builtin_elementwise_fma(R2[5], z, t2[1]))) - (h ? qy.zw : qy.xy);
      const v2f f2 = __builtin_elementwise_fma(R2[6], x, __builtin_elementwise_fma(R2[7], y, __builtin_elementwise_fma(R2[8], z, t2[2]))) - (h ? qz.zw : qz.xy);
      const v2f dsq = __builtin_elementwise_fma(f0, f0, __builtin_elementwise_fma(f1, f1, f2 * f2));
      may_pass += !(dsq.x > hi_f) ? 1u : 0u;
      may_pass += !(dsq.y > hi_f) ? 1u : 0u;
    }
  }
  return may_pass;
}

// ---------------------------------------------------------------------------------
// The reference's in-order bookkeeping (node.cpp:1130-1191: `it += 10 / 20`, the 80 % exit, best-so-far) over the recorded
// outcomes of iterations [w.real_iterations, recorded_end): one wave, 64 summaries fetched per step, the sequential decisions
// on wave-uniform values.  Stops when the loop ends (w.done) or the records run out.  listed(k): iteration k passed the
// pre-screen and has a summary (an iteration that did not counts as {1e6, 0} and leaves nothing in memory).
// COHERENT: the summaries were written by other waves of the SAME launch (the refinement kernel's walk between two windows
// of a pair): agent-scope loads that do not stop at this CU's vector cache.
// ---------------------------------------------------------------------------------
struct WalkRegs {
  int it, real_iterations, valid_iterations, best_idx, best_n;
  float rmse;
  bool done;
};
template <bool COHERENT, class Listed>
__device__ __forceinline__ void walk_records(WalkRegs& w, int recorded_end, int I, int n_all, uint32_t thr,
                                             const IterSum* __restrict__ sum_pair, Listed listed, int lane) {
  while (!w.done && w.it < I && w.real_iterations < recorded_end) {
    const int k0 = w.real_iterations;
    const int G = min(kWave, recorded_end - k0);
    int rn_l = 0;
    double rerr_l = 0.0;
    if (lane < G && listed(k0 + lane)) {
      if (COHERENT) {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(sum_pair + (k0 + lane));
        const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rerr_l = __longlong_as_double((long long)a);
        rn_l = (int)(uint32_t)b;
      } else {
        const IterSum su = sum_pair[k0 + lane];
        rn_l = su.rn;
        rerr_l = su.rerr;
      }
    }
    // Iterations whose refinement left refined_matches empty (:1171 fails) only count: `if (!(it < I)) break;
    // real_iterations++; ++it` -- a run of them is taken in one step; only the others are looked at one by one.
    const uint64_t with_matches = __ballot(lane < G && rn_l > 0);
    for (int g = 0; g < G;) {
      const uint64_t rest = with_matches >> g;
      const int run = rest != 0ull ? (int)__builtin_ctzll(rest) : G - g;  // empty iterations before the next one with matches
      if (run > 0) {
        const int can = min(run, max(I - w.it, 0));  // (`it` may have jumped beyond ransac_iterations, :1186-1187)
        w.real_iterations += can;          // :1139
        w.it += can;
        if (can < run) { w.done = true; break; }  // the next check of `it < ransac_iterations` fails (:1130)
        g += run;
        if (g >= G) break;
      }
      if (!(w.it < I)) { w.done = true; break; }
      w.real_iterations++;  // :1139
      const int refined_n = __builtin_amdgcn_readlane(rn_l, g);
      const double refined_error = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rerr_l), g),
                                                    __builtin_amdgcn_readlane(__double2loint(rerr_l), g));
      // refined_n > 0 (:1171)
      w.valid_iterations++;
      if (refined_error <= (double)w.rmse && refined_n >= w.best_n && (uint32_t)refined_n >= thr) {  // :1177
        w.rmse = (float)refined_error;  // :1182
        w.best_idx = k0 + g;
        w.best_n = refined_n;
        if ((double)refined_n > (double)n_all * 0.5) w.it += 10;   // :1186
        if ((double)refined_n > (double)n_all * 0.75) w.it += 10;  // :1187
        if ((double)refined_n > (double)n_all * 0.8) { w.done = true; break; }  // :1188
      }
      ++w.it;
      ++g;
    }
  }
}

// a pair is "junk-heavy" (class 2 of the record / replay plan) when at most kClass2Num / kClass2Den of the first phase's
// iterations produced a refined hypothesis
#ifndef RGBDFE_CLASS2_NUM
#define RGBDFE_CLASS2_NUM 9
#define RGBDFE_CLASS2_DEN 14
#endif
constexpr int kClass2Num = RGBDFE_CLASS2_NUM, kClass2Den = RGBDFE_CLASS2_DEN;

// The class a pair is treated as from the second phase on.  WalkState::speculate: 0 = `it` has jumped ahead, 1 = no
// jump and mostly valid hypotheses, 2 = no jump and junk-heavy.  Class 1 behaves like class 0 (phase by phase) unless the
// batch has very few such pairs (walk[n_pairs].state counts them, < 1/64 of the batch): then keeping the third and fourth
// phase's launches alive for a handful of long waves costs more than recording those pairs to the end like class 2.
__device__ __forceinline__ int effective_class(const WalkState* __restrict__ walk, uint32_t pair, uint32_t n_pairs) {
  const int c = walk[pair].speculate;
  if (c != 1) return c;
  return ((uint32_t)walk[n_pairs].state * 64u <= n_pairs) ? 2 : 0;
}

}  // namespace rgbdfe
