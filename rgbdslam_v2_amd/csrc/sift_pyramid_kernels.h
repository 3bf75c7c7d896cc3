// sift_pyramid_kernels.h -- the shape-static device half of the SIFT extraction (sift_extract.hip includes it; so does the
// CPU emulation of tests/emu/emu_sift.cpp, which runs these kernel SOURCES on the host against SiftGPU's own kernels):
// image in, the x2 base, one Gaussian level per launch, the 2:1 decimation between octaves, the extremum flags, the per-level
// scan of the row counts and the ordered emit into the candidate lists -- with their launch chains (launch_pyramid,
// launch_key_flags, launch_key_lists) and the extractor's geometry (SiftExtractor::plan_geometry / bind_levels / init_params).
// Reference: external/SiftGPU/src/SiftGPU/ProgramCU.cu:113-688, PyramidCU.cpp:86-306, 738-850, 946-1066.
#pragma once
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "rgbdfe.h"
#include "sift_extract.h"

namespace rgbdfe {

namespace {

constexpr int kMaxTaps = 33;  // KERNEL_MAX_WIDTH (ProgramCU.cu:40)
struct Taps { float k[kMaxTaps]; int fw; };

// ---- image in: bytes -> luminance / 255 (GLTexInput::DownSamplePixelDataI2F, GLTexImage.cpp:808-831), width cut to w4 ----
// (every kernel of this file serves a BATCH of frames: one grid dimension is the frame, each buffer has a per-frame stride)
__global__ __launch_bounds__(256) void sift_convert_kernel(const uint8_t* __restrict__ gray, int cols, int w4, int rows,
                                                           float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w4 * rows) return;
  gray += (size_t)blockIdx.y * rows * cols;
  out += (size_t)blockIdx.y * w4 * rows;
  const int y = i / w4, x = i - y * w4;
  out[i] = (float)(int)gray[(size_t)y * cols + x] / 255.0f;
}

// The "-fo -1" first octave: the input at twice its size (behaviour of UpsampleKernel<1>, ProgramCU.cu:221-265).  One thread
// makes the two output pixels above source pixel (x, y2 / 2): even output rows repeat the source row, odd ones are the mean of
// the rows above and below; even output columns take that value, odd ones the mean with the right-hand neighbour.  The
// source is addressed as ONE linear array, as the reference's linear texture is: the neighbour of a row's last pixel is the
// first pixel of the next row, and anything past the last pixel reads as 0.
__global__ __launch_bounds__(128) void sift_upsample2_kernel(const float* __restrict__ src, int w, int h,
                                                             float* __restrict__ dst, size_t dst_stride) {
  const int x = blockIdx.x * 128 + threadIdx.x;
  if (x >= w) return;
  const int pixels = w * h;
  src += (size_t)blockIdx.z * pixels;
  dst += (size_t)blockIdx.z * dst_stride;
  auto px = [&](int i) -> float { return i < pixels ? src[i] : 0.0f; };
  const int y2 = blockIdx.y;               // output row
  const int at = (y2 >> 1) * w + x;        // the source pixel above-left of the output pair
  float here = px(at), right = px(at + 1);
  if (y2 & 1) {                            // between two source rows: half of each
    here = px(at + w) * 0.5f + 0.5f * here;
    right = px(at + w + 1) * 0.5f + 0.5f * right;
  }
  float* __restrict__ out = dst + (size_t)(w * y2 + x) * 2;
  out[0] = here;
  out[1] = here * 0.5f + right * 0.5f;
}

// DownsampleKernel<1> (ProgramCU.cu:283-294)
__global__ __launch_bounds__(128) void sift_downsample2_kernel(const float* __restrict__ src, int src_w, int dst_w, int dst_h,
                                                               float* __restrict__ dst, size_t frame_stride) {
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c >= dst_w) return;
  src += (size_t)blockIdx.z * frame_stride;
  dst += (size_t)blockIdx.z * frame_stride;
  const int r = blockIdx.y;
  const int sc = min(c << 1, src_w - 1);
  dst[r * dst_w + c] = src[(size_t)(r << 1) * src_w + sc];
}

// One Gaussian level: FilterH then FilterV (ProgramCU.cu:113-218) in one launch.  A block owns a TW x TH output tile: it
// stages the (TH + 2R) x (TW + 2R) source patch in LDS (rows / columns clamped to the image as the two reference kernels
// clamp their fetches), filters it horizontally into a second LDS plane, then vertically into the output.  value starts
// at 0 and the taps are added in ascending order, multiply and add unfused: the reference's sums, bit for bit.
// Two tile shapes: 64 x 16 outputs per block for the large planes, 16 x 16 for planes of a few thousand pixels, where the
// level-after-level dependency makes the latency of one block the cost of the launch.  FW is a template parameter so
// that the tap loops unroll.
template <int FW, int TW, int TH>
__global__ __launch_bounds__(256) void sift_filter_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                          Taps taps, size_t src_stride, size_t dst_stride) {
  constexpr int R = FW >> 1;
  src += (size_t)blockIdx.z * src_stride;
  dst += (size_t)blockIdx.z * dst_stride;
  constexpr int pw = TW + 2 * R, ph = TH + 2 * R;
  __shared__ float patch[ph * pw];   // source rows / columns clamped to the image
  __shared__ float hrow[ph * TW];    // horizontally filtered
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int tid = threadIdx.x;
  for (int i = tid; i < ph * pw; i += 256) {
    const int py = i / pw, px = i - py * pw;
    int gy = y0 - R + py, gx = x0 - R + px;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
    patch[i] = src[(size_t)gy * w + gx];
  }
  __syncthreads();
  for (int i = tid; i < ph * TW; i += 256) {
    const int py = i / TW, px = i - py * TW;
    const float* p = patch + py * pw + px;
    float value = 0.f;
#pragma unroll
    for (int t = 0; t < FW; ++t) value += p[t] * taps.k[t];
    hrow[i] = value;
  }
  __syncthreads();
  for (int i = tid; i < TH * TW; i += 256) {
    const int ty = i / TW, tx = i - ty * TW;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx >= w || gy >= h) continue;
    const float* p = hrow + ty * TW + tx;
    float value = 0.f;
#pragma unroll
    for (int t = 0; t < FW; ++t) value += p[t * TW] * taps.k[t];
    dst[(size_t)gy * w + gx] = value;
  }
}

// The same level for the LARGE planes (octaves 0 - 2 of a VGA frame, where nearly all of the pyramid's pixels are): a 64 x TH
// output tile per workgroup, TH = 64 or 32 -- the horizontal pass runs over TH + 2R rows, so its overhead falls from (16 + 2R)
// / 16 to (TH + 2R) / TH -- and both passes keep their inputs in REGISTERS instead of re-reading LDS once per tap:
//   horizontal: a work item = 4 neighbouring outputs of a patch row; their 4 + 2R inputs arrive as (4 + 2R + 3) / 4 aligned
//               16-byte LDS reads (consecutive lanes read consecutive 16-byte chunks: conflict-free), the 4 results leave as one;
//   vertical:   a work item = TH / 8 rows of two neighbouring columns; the horizontally filtered values stream through one
//               register pair, each read feeding every output whose window holds it, as packed f32 multiplies / adds (two
//               IEEE operations per instruction: v_pk_mul_f32 / v_pk_add_f32).  Lanes run along the row: LDS and the global
//               store see whole 256-byte rows.
// Every output is still 0 + in[0] * k[0] + in[1] * k[1] + ... in that order, products and sums rounded separately: the planes
// do not change by a bit (tests/test_emu_sift_kernels.py runs this source on the host against SiftGPU's kernels;
// tests/test_gpu_sift_extract.py the compiled kernel).  The patch is staged row by row -- one wave per row, lane = column, so
// the row clamp is scalar work and the address a lane offset -- with the 2R right-hand columns of four rows sharing a load.
typedef float pair_f32 __attribute__((vector_size(8)));
template <int FW, int TH>
__global__ __launch_bounds__(256) void sift_filter_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                               Taps taps, size_t src_stride, size_t dst_stride) {
  constexpr int R = FW >> 1;
  constexpr int NCH = (4 + 2 * R + 3) / 4;       // 16-byte chunks a horizontal work item reads
  constexpr int PW = 60 + 4 * NCH;               // patch row: 64 + 2R columns, rounded up to whole chunks
  constexpr int PH = TH + 2 * R;
  constexpr int RV = TH / 8;                     // rows per vertical work item (x 2 columns): 256 items, one per thread
  __shared__ __attribute__((aligned(16))) float patch[PH * PW];
  __shared__ __attribute__((aligned(16))) float hrow[PH * 64];
  src += (size_t)blockIdx.z * src_stride;
  dst += (size_t)blockIdx.z * dst_stride;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TH;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {  // ---- the source patch, rows and columns clamped to the image (FilterH / FilterV clamp their fetches).  Fixed trip
     //      counts, loads first, LDS writes after: all of a thread's loads are in flight together ------------------------
    constexpr int KM = (PH + 3) / 4, KT = (PH * 2 * R + 255) / 256;
    float vm[KM], vt[KT];
    int gx = x0 - R + lane;
    gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      int gy = y0 - R + wave + 4 * k;      // (a row past the patch re-reads the image's last row: loaded, not stored)
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      vm[k] = src[(size_t)gy * w + gx];
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int i = tid + 256 * k;
      const int py = i / (2 * R), px = 64 + (i - py * (2 * R));
      int gy = y0 - R + py, tx = x0 - R + px;
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      tx = tx > w - 1 ? w - 1 : tx;
      vt[k] = src[(size_t)gy * w + tx];
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (wave + 4 * k < PH) patch[(wave + 4 * k) * PW + lane] = vm[k];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int i = tid + 256 * k;
      const int py = i / (2 * R), px = 64 + (i - py * (2 * R));
      if (i < PH * 2 * R) patch[py * PW + px] = vt[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < PH * 16; i += 256) {     // ---- horizontal ------------------------------------------------------
    const int py = i >> 4, t4 = (i & 15) * 4;
    float in[4 * NCH];
    const float4* p = reinterpret_cast<const float4*>(patch + py * PW + t4);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 v = p[c];
      in[4 * c] = v.x; in[4 * c + 1] = v.y; in[4 * c + 2] = v.z; in[4 * c + 3] = v.w;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < FW; ++t) {
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += in[o + t] * taps.k[t];
    }
    float4 r;
    r.x = acc[0]; r.y = acc[1]; r.z = acc[2]; r.w = acc[3];
    *reinterpret_cast<float4*>(hrow + py * 64 + t4) = r;
  }
  __syncthreads();
  for (int i = tid; i < (TH / RV) * 32; i += 256) {   // ---- vertical: two columns x RV rows per work item --------------
    const int tx = (i & 31) * 2, ty = (i >> 5) * RV;
    const float* p = hrow + ty * 64 + tx;
    pair_f32 acc[RV];
#pragma unroll
    for (int o = 0; o < RV; ++o) acc[o] = pair_f32{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < RV + 2 * R; ++k) {
      const pair_f32 v = *reinterpret_cast<const pair_f32*>(p + k * 64);
#pragma unroll
      for (int o = 0; o < RV; ++o)
        if (k - o >= 0 && k - o < FW) acc[o] = acc[o] + v * pair_f32{taps.k[k - o], taps.k[k - o]};
    }
    const int gx = x0 + tx;   // w is a multiple of 4: a pair is inside the image or outside it as a whole
    if (gx < w) {
#pragma unroll
      for (int o = 0; o < RV; ++o)
        if (y0 + ty + o < h) *reinterpret_cast<pair_f32*>(dst + (size_t)(y0 + ty + o) * w + gx) = acc[o];
    }
  }
}

struct FilterArgs { const float* src; float* dst; int w, h, nf; size_t src_stride, dst_stride; };
// which kernel a level's launch takes: 0 = 16 x 16 tiles, 1 = 64 x 16, 2 = 64 x 32 (register-window kernel), 3 = 64 x 64
inline int filter_tile_choice(int w, int h, int nf) {
  static const int forced = [] { const char* e = getenv("RGBDFE_SIFT_FILTER_TILE"); return e ? atoi(e) : -1; }();
  if (forced >= 0 && forced <= 3) return forced;
  if ((size_t)w * h * nf <= (size_t)160 * 120) return 0;   // few thousand pixels in the whole launch: small tiles, more workgroups
  if ((size_t)w * h < (size_t)320 * 240) return 1;
  const size_t tiles64 = (size_t)((w + 63) / 64) * ((h + 63) / 64) * nf;
  return tiles64 >= 1024 ? 3 : 2;                          // 64 rows per tile once that still leaves four workgroups per CU
}
template <int FW>
void launch_filter(const FilterArgs& a, const Taps& t, hipStream_t s, int choice) {
  switch (choice >= 0 ? choice : filter_tile_choice(a.w, a.h, a.nf)) {
    case 0:
      hipLaunchKernelGGL((sift_filter_kernel<FW, 16, 16>), dim3((a.w + 15) / 16, (a.h + 15) / 16, a.nf), dim3(256), 0, s, a.src,
                         a.dst, a.w, a.h, t, a.src_stride, a.dst_stride);
      break;
    case 1:
      hipLaunchKernelGGL((sift_filter_kernel<FW, 64, 16>), dim3((a.w + 63) / 64, (a.h + 15) / 16, a.nf), dim3(256), 0, s, a.src,
                         a.dst, a.w, a.h, t, a.src_stride, a.dst_stride);
      break;
    case 2:
      hipLaunchKernelGGL((sift_filter_tile_kernel<FW, 32>), dim3((a.w + 63) / 64, (a.h + 31) / 32, a.nf), dim3(256), 0, s, a.src,
                         a.dst, a.w, a.h, t, a.src_stride, a.dst_stride);
      break;
    default:
      hipLaunchKernelGGL((sift_filter_tile_kernel<FW, 64>), dim3((a.w + 63) / 64, (a.h + 63) / 64, a.nf), dim3(256), 0, s, a.src,
                         a.dst, a.w, a.h, t, a.src_stride, a.dst_stride);
      break;
  }
}

// choice: the tile shape (filter_tile_choice's codes), -1 = by plane size
void launch_filter_any(const FilterArgs& a, const Taps& t, hipStream_t s, int choice = -1) {
  switch (t.fw) {   // ProgramCU::FilterImage's switch over the odd widths 5 .. 33 (ProgramCU.cu:430-448)
    case 5: launch_filter<5>(a, t, s, choice); break;
    case 7: launch_filter<7>(a, t, s, choice); break;
    case 9: launch_filter<9>(a, t, s, choice); break;
    case 11: launch_filter<11>(a, t, s, choice); break;
    case 13: launch_filter<13>(a, t, s, choice); break;
    case 15: launch_filter<15>(a, t, s, choice); break;
    case 17: launch_filter<17>(a, t, s, choice); break;
    case 19: launch_filter<19>(a, t, s, choice); break;
    case 21: launch_filter<21>(a, t, s, choice); break;
    case 23: launch_filter<23>(a, t, s, choice); break;
    case 25: launch_filter<25>(a, t, s, choice); break;
    case 27: launch_filter<27>(a, t, s, choice); break;
    case 29: launch_filter<29>(a, t, s, choice); break;
    case 31: launch_filter<31>(a, t, s, choice); break;
    default: launch_filter<33>(a, t, s, choice); break;
  }
}

// ---- extrema -----------------------------------------------------------------------------------------------------------
struct KeyEval { float result, dx, dy, ds; };

// one row [c0 c1 c2 | rhs] of the 3 x 3 system of the sub-pixel fit
struct FitRow {
  float c0, c1, c2, rhs;
  // the row with a non-negative leading coefficient
  static __device__ __forceinline__ FitRow oriented(float c0, float c1, float c2, float rhs) {
    return c0 > 0 ? FitRow{c0, c1, c2, rhs} : FitRow{-c0, -c1, -c2, -rhs};
  }
};
__device__ __forceinline__ void exchange(FitRow& a, FitRow& b) { const FitRow t = a; a = b; b = t; }

// Is the pixel `at` of D[l] = G[l] - G[l-1] a keypoint candidate, and where does the fitted extremum lie?  (Behaviour of
// ComputeKEY_Kernel, ProgramCU.cu:524-640, for an interior pixel.)  g[0..3] = the Gaussian planes G[l-2] .. G[l+1]: the three
// DoG levels involved are differences of neighbouring planes, formed here instead of being stored (ComputeDOG_Kernel's
// `v - vp`, :457-489).  The tests, cheapest first -- each one only ever rejects:
//   contrast gate      |D| above 0.8 * threshold;
//   extremum           strictly above (below) its two row neighbours, not below (above) any of the other 24 neighbours in
//                      scale space (ties: see `holds`); `rim` follows the neighbour closest to the centre value on the side
//                      that matters;
//   edge response      principal-curvature ratio of the 2 x 2 spatial Hessian;
//   sub-pixel fit      Newton step (dx, dy, ds) from the 3 x 3 Hessian system, solved by elimination with the reference's
//                      pivot choices; rejected when the step leaves the pixel / level or the fitted contrast is too low.
// result = +1 for a maximum that is strictly above all 26 neighbours, -1 for every other candidate, 0 = no candidate.
__device__ __forceinline__ KeyEval key_eval(const float* const g[4], int w, int at, float gate, float contrast_threshold,
                                            float edge_threshold) {
  const KeyEval none{0.f, 0.f, 0.f, 0.f};
  auto same = [&](int i) -> float { return g[2][i] - g[1][i]; };    // D[l]
  auto below = [&](int i) -> float { return g[1][i] - g[0][i]; };   // D[l-1]
  auto above = [&](int i) -> float { return g[3][i] - g[2][i]; };   // D[l+1]
  const float v = same(at);
  if (fabsf(v) <= gate) return none;
  const float west = same(at - 1), east = same(at + 1);
  const bool peak = v > fmaxf(west, east);
  if (!peak && !(v < fminf(west, east))) return none;   // between its row neighbours (or level with one of them)
  float rim = peak ? fmaxf(west, east) : fminf(west, east);
  // three more neighbours: does the centre still stand out?  A valley may be level with any neighbour; a peak may only be
  // level with one of the LAST three looked at (it is then reported as -1): a peak found level with an earlier neighbour
  // is dropped when the next three are looked at -- the reference's behaviour, kept.
  auto holds = [&](float a, float b, float c) -> bool {
    if (peak) {
      if (!(v > rim)) return false;
      rim = fmaxf(fmaxf(fmaxf(rim, a), b), c);
      return !(v < rim);
    }
    rim = fminf(fminf(fminf(rim, a), b), c);
    return !(v > rim);
  };
  const int up = at - w, down = at + w;
  const float nw = same(up - 1), north = same(up), ne = same(up + 1);
  if (!holds(nw, north, ne)) return none;
  const float sw = same(down - 1), south = same(down), se = same(down + 1);
  if (!holds(sw, south, se)) return none;
  // edge response: det(H) > 0 and trace(H)^2 / det(H) within the threshold
  const float twice = v * 2.0f;
  const float hxx = west + east - twice;
  const float hyy = north + south - twice;
  const float hxy = 0.25f * (se + nw - sw - ne);
  const float det = hxx * hyy - hxy * hxy;
  const float trace_sq = (hxx + hyy) * (hxx + hyy);
  if (det <= 0 || trace_sq > edge_threshold * det) return none;
  // the 9 + 9 neighbours in the levels below and above
  const float b_nw = below(up - 1), b_n = below(up), b_ne = below(up + 1);
  if (!holds(b_nw, b_n, b_ne)) return none;
  const float b_w = below(at - 1), b_c = below(at), b_e = below(at + 1);
  if (!holds(b_w, b_c, b_e)) return none;
  const float b_sw = below(down - 1), b_s = below(down), b_se = below(down + 1);
  if (!holds(b_sw, b_s, b_se)) return none;
  const float a_nw = above(up - 1), a_n = above(up), a_ne = above(up + 1);
  if (!holds(a_nw, a_n, a_ne)) return none;
  const float a_w = above(at - 1), a_c = above(at), a_e = above(at + 1);
  if (!holds(a_w, a_c, a_e)) return none;
  const float a_sw = above(down - 1), a_s = above(down), a_se = above(down + 1);
  if (!holds(a_sw, a_s, a_se)) return none;
  (void)b_nw; (void)b_ne; (void)b_sw; (void)b_se; (void)a_nw; (void)a_ne; (void)a_sw; (void)a_se;
  // sub-pixel fit ("-s 1"): H * step = -gradient by central differences over (x, y, scale)
  KeyEval out{0.f, 0.f, 0.f, 0.f};
  bool keep = true;
  {
    const float gx = 0.5f * (east - west);
    const float gy = 0.5f * (south - north);
    const float gs = 0.5f * (a_c - b_c);
    const float hss = (a_c + b_c - twice);
    const float hxs = 0.25f * (a_e + b_w - a_w - b_e);
    const float hys = 0.25f * (a_s + b_n - a_n - b_s);
    FitRow r0 = FitRow::oriented(hxx, hxy, hxs, -gx);
    FitRow r1 = FitRow::oriented(hxy, hyy, hys, -gy);
    FitRow r2 = FitRow::oriented(hxs, hys, hss, -gs);
    const float lead = fmaxf(fmaxf(r0.c0, r1.c0), r2.c0);
    if (lead >= 1e-10) {
      // first pivot: the row with the largest leading coefficient (the second row wins a tie, then the third)
      if (lead == r1.c0) exchange(r0, r1);
      else if (lead == r2.c0) exchange(r0, r2);
      r0.c1 /= r0.c0; r0.c2 /= r0.c0; r0.rhs /= r0.c0;
      r1.c1 -= r1.c0 * r0.c1; r1.c2 -= r1.c0 * r0.c2; r1.rhs -= r1.c0 * r0.rhs;
      r2.c1 -= r2.c0 * r0.c1; r2.c2 -= r2.c0 * r0.c2; r2.rhs -= r2.c0 * r0.rhs;
      if (fabsf(r2.c1) > fabsf(r1.c1)) exchange(r1, r2);   // second pivot
      if (fabsf(r1.c1) >= 1e-10) {
        r1.c2 /= r1.c1; r1.rhs /= r1.c1;
        r2.c2 -= r2.c1 * r1.c2; r2.rhs -= r2.c1 * r1.rhs;
        if (fabsf(r2.c2) >= 1e-10) {   // back substitution
          out.ds = r2.rhs / r2.c2;
          out.dy = r1.rhs - out.ds * r1.c2;
          out.dx = r0.rhs - out.ds * r0.c2 - out.dy * r0.c1;
          keep = fabsf(v + 0.5f * (out.dx * gx + out.dy * gy + out.ds * gs)) > contrast_threshold &&
                 fabsf(out.ds) < 1.0f && fabsf(out.dx) < 1.0f && fabsf(out.dy) < 1.0f;
        }
      }
    }
  }
  if (keep) out.result = (peak && v > rim) ? 1.0f : -1.0f;
  return out;
}

// per-frame strides of the batch: frame f's planes / flags / row counters / level totals / candidates start f strides
// behind frame 0's, which is what the LevelDesc records point at
struct FrameStrides { size_t planes, flags, cand; int rows, lvltot; };
__device__ __forceinline__ SiftExtractor::LevelDesc level_of_frame(SiftExtractor::LevelDesc L, const FrameStrides& st, int f) {
#pragma unroll
  for (int k = 0; k < 4; ++k) L.g[k] += (size_t)f * st.planes;
  L.flags += (size_t)f * st.flags;
  return L;
}

// DetectKeypointsEX for one 64 x 16 pixel tile of an OCTAVE, all its kDogLevels key levels at once.  The tile's eight
// Gaussian planes (one pixel of margin) are staged in LDS once -- whole rows, lane = column: every plane of the pyramid is read
// from HBM once, coalesced -- and everything after that is LDS work: the contrast gate |D| > 0.8 * threshold that most
// pixels fail (two reads per level), then key_eval's neighbour tests for the few that pass, each a dependent round trip
// that used to go to L2 (the per-row form of rounds 3 - 4 spent its time waiting for those: 49 us per VGA frame against 10
// us of plane traffic).  A flag byte per pixel and level; the rows' counts are gathered in LDS and leave as one global add
// per (level, row) that has any.
// Counted = what InitHist_Kernel (ProgramCU.cu:665-688) counts: rows 1 .. h-2, columns 1 .. w-2 with a non-zero key.
using KeyTile = SiftExtractor::KeyTile;
constexpr int kKeyTileH = SiftExtractor::kKeyTileH, kKeyPW = 68;   // LDS row: 66 columns, padded to a multiple of 4
__global__ __launch_bounds__(256) void sift_key_flag_kernel(const float* __restrict__ planes, int8_t* __restrict__ flags,
                                                            const SiftExtractor::OctDesc* __restrict__ octs,
                                                            const KeyTile* __restrict__ tiles, int* __restrict__ rowcnt,
                                                            float dog_threshold0, float dog_threshold, float edge_threshold,
                                                            FrameStrides st) {
  constexpr int kDog = SiftExtractor::kDogLevels, kLv = SiftExtractor::kLevels;
  constexpr int PH = kKeyTileH + 2;
  __shared__ float G[kLv][PH * kKeyPW];
  __shared__ int cnt[kDog * kKeyTileH];
  const KeyTile T = tiles[blockIdx.x];
  const SiftExtractor::OctDesc O = octs[T.oct];
  // (planes and flags are addressed from the kernel's own pointer arguments, not through pointers kept in a table: the
  //  compiler then knows the loads are global, which lets it issue all of a thread's 36 loads before the first LDS write --
  //  through table pointers they were flat loads, each ordered behind the LDS write before it: 90 us per frame instead of 49)
  planes += (size_t)blockIdx.y * st.planes + O.plane_off;
  flags += (size_t)blockIdx.y * st.flags + O.flag_off;
  rowcnt += (size_t)blockIdx.y * st.rows + O.row0;
  const int w = O.w, h = O.h;
  const size_t plane = (size_t)w * h;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < kDog * kKeyTileH) cnt[tid] = 0;
  {  // the planes G[0] .. G[7] of the octave: every load of the thread first, the LDS writes after
    constexpr int KM = (PH + 3) / 4, KT = (kLv * PH * 2 + 255) / 256;
    float vm[kLv][KM], vt[KT];
    int gx = T.x0 - 1 + lane;
    gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      int gy = T.y0 - 1 + wave + 4 * k;    // (a row past the tile re-reads the image's last row: loaded, not stored)
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
#pragma unroll
      for (int p = 0; p < kLv; ++p) vm[p][k] = planes[plane * p + (size_t)gy * w + gx];
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {          // columns 64, 65 of every row
      const int i = tid + 256 * k;
      const int p = i / (PH * 2), q = i - p * (PH * 2), r = q >> 1, c = 64 + (q & 1);
      int gy = T.y0 - 1 + r, tx = T.x0 - 1 + c;
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      tx = tx > w - 1 ? w - 1 : tx;
      vt[k] = i < kLv * PH * 2 ? planes[plane * p + (size_t)gy * w + tx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KM; ++k)
      if (wave + 4 * k < PH) {
#pragma unroll
        for (int p = 0; p < kLv; ++p) G[p][(wave + 4 * k) * kKeyPW + lane] = vm[p][k];
      }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int i = tid + 256 * k;
      const int p = i / (PH * 2), q = i - p * (PH * 2), r = q >> 1, c = 64 + (q & 1);
      if (i < kLv * PH * 2) G[p][r * kKeyPW + c] = vt[k];
    }
  }
  __syncthreads();
  const int col = T.x0 + lane;
#pragma unroll 1
  for (int u = 0; u < kKeyTileH / 4; ++u) {   // (rolled: five inlined key_eval bodies instead of twenty)
    const int lr = wave * (kKeyTileH / 4) + u, row = T.y0 + lr;
    if (row >= h) break;
    const bool interior = col < w && row > 0 && col > 0 && row < h - 1 && col < w - 1;
    const int at = (lr + 1) * kKeyPW + lane + 1;
    float c[kDog + 1];   // G[1] .. G[6] at the pixel: level j's centre value is G[j + 2] - G[j + 1]
#pragma unroll
    for (int j = 0; j <= kDog; ++j) c[j] = G[j + 1][at];
#pragma unroll
    for (int j = 0; j < kDog; ++j) {
      int8_t flag = 0;
      if (interior && fabsf(c[j + 1] - c[j]) > dog_threshold0) {
        const float* const g[4] = {G[j], G[j + 1], G[j + 2], G[j + 3]};
        const KeyEval e = key_eval(g, kKeyPW, at, dog_threshold0, dog_threshold, edge_threshold);
        flag = e.result > 0.f ? 1 : (e.result < 0.f ? -1 : 0);
        if (flag) atomicAdd(&cnt[j * kKeyTileH + lr], 1);
      }
      if (col < w) flags[plane * j + (size_t)row * w + col] = flag;
    }
  }
  __syncthreads();
  if (tid < kDog * kKeyTileH) {
    const int j = tid / kKeyTileH, lr = tid - j * kKeyTileH;
    if (cnt[tid]) atomicAdd(&rowcnt[j * h + T.y0 + lr], cnt[tid]);
  }
}

// per level: exclusive scan of its rows' counts, the level's total
__global__ __launch_bounds__(64) void sift_row_scan_kernel(const SiftExtractor::LevelDesc* __restrict__ levels,
                                                           const int* __restrict__ rowcnt, int* __restrict__ rowoff,
                                                           int* __restrict__ lvltot, FrameStrides st,
                                                           int* __restrict__ host_lvltot) {   // (pinned host copy of lvltot, or nullptr)
  const SiftExtractor::LevelDesc L = levels[blockIdx.x];
  rowcnt += (size_t)blockIdx.y * st.rows;
  rowoff += (size_t)blockIdx.y * st.rows;
  lvltot += (size_t)blockIdx.y * st.lvltot;
  int base = 0;
  for (int r0 = 0; r0 < L.h; r0 += 64) {
    const int r = r0 + (int)threadIdx.x;
    const int c = r < L.h ? rowcnt[L.row0 + r] : 0;
    int incl = c;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if ((int)threadIdx.x >= d) incl += o;
    }
    if (r < L.h) rowoff[L.row0 + r] = base + incl - c;
    base += __shfl(incl, 63);
  }
  if (threadIdx.x == 0) {
    lvltot[blockIdx.x] = base;
    if (host_lvltot != nullptr) host_lvltot[(size_t)blockIdx.y * st.lvltot + blockIdx.x] = base;
  }
}

// one wave per row: the row's extrema in column order -> the level's candidate list (x, y, sign, dx, dy, ds).  A lane takes
// EIGHT neighbouring flag bytes per step (512 columns per step: three steps for the widest plane of a VGA frame, where the
// byte-per-lane form of rounds 3 - 4 took twenty dependent load + ballot rounds), counts its non-zero bytes, an exclusive
// wave scan of the counts gives every flagged pixel its rank in the row, and the few lanes that hold one evaluate it.
// (Sixteen rows per workgroup, a wave walking four of them, was measured too: 77 instead of 44 us per 8 frames -- a row with
// extrema is a chain of dependent loads, and the launch lives on rows in flight, not on workgroup dispatch.)
__global__ __launch_bounds__(64) void sift_key_emit_kernel(const SiftExtractor::LevelDesc* __restrict__ levels,
                                                           const int* __restrict__ row2lvl, int* __restrict__ rowcnt,
                                                           const int* __restrict__ rowoff, const int* __restrict__ lvltot,
                                                           float* __restrict__ cand, int cand_cap, float dog_threshold0,
                                                           float dog_threshold, float edge_threshold, FrameStrides st) {
  const int grow = blockIdx.x;
  rowcnt += (size_t)blockIdx.y * st.rows;
  rowoff += (size_t)blockIdx.y * st.rows;
  lvltot += (size_t)blockIdx.y * st.lvltot;
  cand += (size_t)blockIdx.y * st.cand;
  if (rowcnt[grow] == 0) return;
  // the row's count is zero again for the next batch's flag kernel (atomicAdd): this wave is its last reader -- no memset
  // between batches, in particular no memset NODE when the chain is captured into a hipGraph.  (The barrier: every thread has
  // read the count before one of them resets it -- a wave does that in lockstep anyway, the CPU emulation of tests/emu does not.)
  __syncthreads();
  if (threadIdx.x == 0) rowcnt[grow] = 0;
  const int lvl = row2lvl[grow];
  const SiftExtractor::LevelDesc L = level_of_frame(levels[lvl], st, blockIdx.y);
  const int row = grow - L.row0;
  const int lane = threadIdx.x;
  int base = rowoff[grow];
  for (int l = 0; l < lvl; ++l) base += lvltot[l];
  const int8_t* __restrict__ frow = L.flags + (size_t)row * L.w;   // (w is a multiple of 4: the row starts dword-aligned)
  for (int c0 = 0; c0 < L.w; c0 += 512) {
    const int col0 = c0 + lane * 8;
    uint32_t lo = 0, hi = 0;
    if (col0 < L.w) lo = *reinterpret_cast<const uint32_t*>(frow + col0);
    if (col0 + 4 < L.w) hi = *reinterpret_cast<const uint32_t*>(frow + col0 + 4);
    if (__ballot((lo | hi) != 0) == 0) continue;
    // non-zero bytes of the eight: bit 8 i + 7 of `nz` set for byte i
    const uint64_t bytes = ((uint64_t)hi << 32) | lo;
    const uint64_t nz = (((bytes & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | bytes) & 0x8080808080808080ull;
    const int mine = (int)__popcll(nz);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    int rank = base + incl - mine;
    uint64_t todo = nz;
    while (todo) {
      const int i = (__ffsll((long long)todo) - 1) >> 3;
      todo &= todo - 1;
      const int col = col0 + i;
      if (rank < cand_cap) {
        const KeyEval e = key_eval(L.g, L.w, row * L.w + col, dog_threshold0, dog_threshold, edge_threshold);
        float* o = cand + (size_t)rank * 6;
        o[0] = (float)col; o[1] = (float)row; o[2] = e.result; o[3] = e.dx; o[4] = e.dy; o[5] = e.ds;
      }
      ++rank;
    }
    base += __shfl(incl, 63);
  }
}

// The taps of a Gaussian level: exp(-d^2 / 2 sigma^2) for d = -half .. half, normalised to sum 1 (f32 throughout, summed
// left to right -- the values of ProgramCU::CreateFilterKernel, ProgramCU.cu:370-398, with its width factor 4): half =
// ceil(4 sigma - 1/2) taps either side, kept between 2 and 16 (widths 5 .. 33).
Taps make_taps(float sigma) {
  Taps t{};
  const int half = std::min(std::max((int)ceil(4.0f * sigma - 0.5), 2), kMaxTaps / 2);
  const float inv_var = 1.0f / (sigma * sigma);
  t.fw = 2 * half + 1;
  float sum = 0.f;
  for (int j = 0; j < t.fw; ++j) {
    const int d = j - half;
    const float gd = expf(-0.5f * d * d * inv_var);
    t.k[j] = gd;
    sum += gd;
  }
  const float norm = 1.0f / sum;
  for (int j = 0; j < t.fw; ++j) t.k[j] *= norm;
  return t;
}


// BuildPyramid (PyramidCU.cpp:946-998) for the nf frames whose bytes lie in E.d_gray: the launch chain, nothing else (the
// extractor's enqueue_pyramid and the CPU emulation of tests/emu/emu_sift.cpp both run THIS).  filter_choice: the tile shape
// of every level's launch (filter_tile_choice's codes), -1 = by plane size.
// Two ways to shorten this chain of 56 dependent launches were built and measured in round 5, and dropped:
//   * the octaves that fit LDS (3, 4, 5 of a VGA frame) whole in one launch, a 1024-thread workgroup per frame: 33 launches
//     and the same planes, but 133 us against the 127 us of the 24 launches it replaced, single calls 10 % slower
//     (profiles/r05_logs/sift_octave_tail.txt);
//   * levels 6, 7 of each octave -- only the extremum launch reads them -- on a side stream beside the next octave's chain,
//     40 dependent launches: as plain launches no change (6989 vs 7132 frames/s through the batch entry point), as a captured
//     graph with those edges 4850 -- multi-branch graphs replay slowly on this runtime.
inline void launch_pyramid(const SiftExtractor& E, int nf, hipStream_t s, int filter_choice = -1) {
  constexpr int kLevels = SiftExtractor::kLevels, kDogLevels = SiftExtractor::kDogLevels;
  const int rows = E.H, cols = E.W, w4 = E.w4;
  const unsigned NF = (unsigned)nf;
  const uint8_t* d_gray = E.d_gray;
  float* d_input = E.d_input;
  float* d_up = E.d_up;
  const size_t planes_floats = E.planes_floats, input_floats = E.input_floats;
  hipLaunchKernelGGL(sift_convert_kernel, dim3((w4 * rows + 255) / 256, NF), dim3(256), 0, s, d_gray, cols, w4, rows, d_input);
  auto filter = [&](const float* src, size_t src_stride, float* dst, int w, int h, float sg) {
    launch_filter_any(FilterArgs{src, dst, w, h, nf, src_stride, planes_floats}, make_taps(sg), s, filter_choice);
  };
  for (int i = 0; i < E.octave_num; ++i) {
    const SiftExtractor::Octave& o = E.oct[i];
    if (i == 0) {
      const float sg = E.initial_smooth_sigma(E.octave_min);
      if (E.octave_min < 0) {  // SampleImageU + FilterImage in place through the buffer plane
        hipLaunchKernelGGL(sift_upsample2_kernel, dim3((w4 + 127) / 128, rows << 1, NF), dim3(128), 0, s, d_input, w4, rows, d_up,
                           o.plane);
        filter(d_up, o.plane, o.g[0], o.w, o.h, sg);
      } else {
        filter(d_input, input_floats, o.g[0], o.w, o.h, sg);
      }
    } else {  // SampleImageD from level_ds of the octave below (index level_ds - level_min = 5); sigma_skip1 = 0
      const SiftExtractor::Octave& p = E.oct[i - 1];
      hipLaunchKernelGGL(sift_downsample2_kernel, dim3((o.w + 127) / 128, o.h, NF), dim3(128), 0, s, p.g[kDogLevels], p.w, o.w, o.h,
                         o.g[0], planes_floats);
    }
    for (int l = 1; l < kLevels; ++l) filter(o.g[l - 1], planes_floats, o.g[l], o.w, o.h, E.sigma[l - 1]);
  }
}

// DetectKeypointsEX: the extremum flags + row counts of every octave and level of the nf frames (rowcnt is zero: at allocation, and sift_key_emit_kernel leaves it so)
inline void launch_key_flags(const SiftExtractor& E, int nf, const FrameStrides& st, hipStream_t s) {
  const float tdog = E.dog_threshold, tdog1 = 0.8f * tdog;
  const float tedge = (E.edge_threshold + 1) * (E.edge_threshold + 1) / E.edge_threshold;
  hipLaunchKernelGGL(sift_key_flag_kernel, dim3((unsigned)E.n_key_tiles, (unsigned)nf), dim3(256), 0, s, E.d_planes, E.d_flags,
                     static_cast<const SiftExtractor::OctDesc*>(E.d_octs), static_cast<const KeyTile*>(E.d_key_tiles),
                     E.d_rowcnt, tdog1, tdog, tedge, st);
}

// the list part of GenerateFeatureList: per-level scan of the row counts, then the ordered emit into the candidate lists
inline void launch_key_lists(const SiftExtractor& E, int nf, const FrameStrides& st, hipStream_t s) {
  const float tdog = E.dog_threshold, tdog1 = 0.8f * tdog;
  const float tedge = (E.edge_threshold + 1) * (E.edge_threshold + 1) / E.edge_threshold;
  const int nlv = E.octave_num * SiftExtractor::kDogLevels;
  const int* d_row2lvl = E.d_rowcnt + (size_t)E.total_rows * 2 * E.frames_cap;
  hipLaunchKernelGGL(sift_row_scan_kernel, dim3((unsigned)nlv, (unsigned)nf), dim3(64), 0, s, E.d_levels, E.d_rowcnt, E.d_rowoff,
                     E.d_lvltot, st, E.host_write ? E.h_counts : (int*)nullptr);
  hipLaunchKernelGGL(sift_key_emit_kernel, dim3((unsigned)E.total_rows, (unsigned)nf), dim3(64), 0, s, E.d_levels, d_row2lvl,
                     E.d_rowcnt, E.d_rowoff, E.d_lvltot, E.d_cand, (int)E.cand_cap, tdog1, tdog, tedge, st);
}

}  // namespace

// ---- parameters (SiftParam::ParseSiftParam, SiftGPU.cpp:433-473) with "-d 5 -e 10.0" ------------------------------------------
inline void SiftExtractor::init_params() {  // SiftParam::ParseSiftParam (SiftGPU.cpp:433-473) with "-d 5 -e 10.0"
  if (params_ready) return;
  const int dog_level_num = kDogLevels, level_min = -1, level_max = kDogLevels + 1;
  sigma0 = 1.6f * powf(2.0f, 1.0f / dog_level_num);
  sigmak = powf(2.0f, 1.0f / dog_level_num);
  dsigma0 = sigma0 * sqrtf(1.0f - 1.0f / (sigmak * sigmak));
  for (int i = level_min + 1; i <= level_max; i++) sigma[i - level_min - 1] = dsigma0 * powf(sigmak, float(i));
  dog_threshold = 0.02f / dog_level_num;
  edge_threshold = 10.0f;
  params_ready = true;
}

inline float SiftExtractor::initial_smooth_sigma(int om) const {
  const float sa = sigma0 * powf(2.0f, float(-1) / float(kDogLevels));
  const float sb = 0.5f / powf(2.0f, float(om));
  return sa > sb + 0.001 ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

inline float SiftExtractor::level_sigma(int lev) const { return sigma0 * powf(2.0f, float(lev) / float(kDogLevels)); }

// PyramidCU::InitPyramid / ResizePyramid / FitPyramid (PyramidCU.cpp:86-306): the geometry a frame of this size gets
inline int SiftExtractor::plan_geometry(int rows, int cols, std::string& err) {
  const int tw = cols & 0xfffffffc;  // GLTexInput::TruncateWidthCU (GLTexImage.h:125)
  if (tw < 16 || rows < 16) { err = "image too small for SIFT extraction"; return RGBDFE_ERR_INVALID_ARG; }
  int om = -1;  // "-fo -1"
  int wp = tw << 1, hp = rows << 1;
  while (wp > 3200 || hp > 3200) { om++; wp >>= 1; hp >>= 1; }  // GlobalUtil::_texMaxDim (GlobalUtil.cpp:86)
  if (om > 0) { err = "images beyond 3200 x 3200 pixels are not supported"; return RGBDFE_ERR_CAPACITY; }
  int on = (int)floor(log(double(std::min(wp, hp))) / log(2.0)) - 3;
  if (on < 1) on = 1;
  if (on > kMaxOctaves) on = kMaxOctaves;
  W = cols; H = rows; w4 = tw; octave_min = om; octave_num = on;
  size_t total = 0, foff = 0;
  int w = wp, h = hp;
  total_rows = 0;
  for (int i = 0; i < on; ++i) {
    oct[i].w = ((w + 3) / 4) * 4;
    oct[i].h = h;
    oct[i].plane = (size_t)oct[i].w * h;
    total += oct[i].plane * kLevels;
    foff += oct[i].plane * kDogLevels;
    total_rows += h * kDogLevels;
    w >>= 1; h >>= 1;
  }
  planes_floats = total;
  flags_bytes = foff;
  input_floats = (size_t)tw * rows;
  return RGBDFE_OK;
}

// frame 0's planes / flag planes inside d_planes / d_flags; the level records, the row -> level table and the 64 x 16 pixel
// tiles of every octave (the extremum launch's work list) on the host
inline void SiftExtractor::bind_levels() {
  const int on = octave_num;
  size_t off = 0, foff = 0;
  for (int i = 0; i < on; ++i)
    for (int l = 0; l < kLevels; ++l) { oct[i].g[l] = d_planes + off; off += oct[i].plane; }
  h_levels.assign((size_t)on * kDogLevels, LevelDesc{});
  h_row2lvl.assign((size_t)total_rows, 0);
  int row0 = 0;
  for (int i = 0; i < on; ++i)
    for (int j = 0; j < kDogLevels; ++j) {
      LevelDesc& L = h_levels[(size_t)i * kDogLevels + j];
      const int l = j + 2;  // key level: DoG l - 1, l, l + 1 = Gaussian l - 2 .. l + 1
      for (int k = 0; k < 4; ++k) L.g[k] = oct[i].g[l - 2 + k];
      L.flags = d_flags + foff;
      L.w = oct[i].w; L.h = oct[i].h; L.row0 = row0;
      for (int r = 0; r < oct[i].h; ++r) h_row2lvl[(size_t)row0 + r] = i * kDogLevels + j;
      row0 += oct[i].h;
      foff += oct[i].plane;
    }
  h_octs.assign((size_t)on, OctDesc{});
  for (int i = 0; i < on; ++i) {   // the same facts per OCTAVE, as offsets: what the extremum kernel addresses with
    const LevelDesc& L0 = h_levels[(size_t)i * kDogLevels];
    h_octs[(size_t)i] = OctDesc{(size_t)(oct[i].g[0] - d_planes), (size_t)(L0.flags - d_flags), oct[i].w, oct[i].h, L0.row0, 0};
  }
  h_key_tiles.clear();
  for (int i = 0; i < on; ++i)
    for (int y0 = 0; y0 < oct[i].h; y0 += kKeyTileH)
      for (int x0 = 0; x0 < oct[i].w; x0 += 64) h_key_tiles.push_back(KeyTile{i, x0, y0});
}

}  // namespace rgbdfe
