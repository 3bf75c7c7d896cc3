// sift_extract.hip -- SIFT extraction on CDNA4: SiftGPUWrapper::detect (src/sift_gpu_wrapper.cpp:113-167), i.e. the
// pipeline of the SiftGPU the reference vendors (external/SiftGPU/src/SiftGPU/), CUDA flavour:
//   PyramidCU::BuildPyramid (PyramidCU.cpp:946-998)          up-sample x2 ("-fo -1"), 8 Gaussian levels per octave
//   PyramidCU::DetectKeypointsEX (:1000-1066)                 DoG, extrema + edge test + sub-pixel solve (ComputeKEY_Kernel,
//                                                             ProgramCU.cu:524-640)
//   PyramidCU::GenerateFeatureList (:738-850)                 raster-ordered lists, coarse octaves first, "-tc2" limit
//   PyramidCU::GetFeatureOrientations (:1145-1172)            36-bin histograms, two orientations (ProgramCU.cu:774-935)
//   PyramidCU::ReshapeFeatureListCPU (:501-585)               one feature per orientation, level -> image coordinates
//   PyramidCU::GetFeatureDescriptors (:393-432)               4x4x8 histograms, unnormalised ("-unn", ProgramCU.cu:967-1046)
// This is a new design, not a translation of those kernels:
//   * a Gaussian level is ONE launch (horizontal + vertical pass fused through LDS) instead of two, with the same
//     per-tap accumulation order, so every plane equals the reference's bit for bit;
//   * no DoG, gradient or keypoint planes exist: the extremum test recomputes D = G[l] - G[l-1] from the Gaussian planes
//     (the same subtraction), all octaves and levels in one launch that leaves one flag byte per pixel + per-row counts;
//     an ordered ballot compaction (scan + emit) replaces the reference's 4-ary histogram pyramid and yields the same
//     raster-ordered lists; the orientation and descriptor kernels take gradients from the Gaussian plane on the fly
//     (same differences, sqrt, atan2) -- the 45 floats per pixel the reference keeps shrink to 8;
//   * a keypoint's orientation histogram and each of its 16 descriptor cells are wave-wide jobs (the reference gives each
//     one thread); per-sample arithmetic is the reference's, the sums run lane-parallel in a fixed order.
// Arithmetic without transcendental functions (pyramid, extrema, sub-pixel offsets, lists) is exact against the
// reference's kernels compiled on the CPU emulation (oracle/_ref/libref_siftgpu.so); exp / atan2 / pow / sincos come
// from the device's libm and differ from glibc's by ulps: tests/test_gpu_sift_extract.py states the tolerances.
#include "sift_extract.h"

#include <math.h>
#include <string.h>

#include <algorithm>

#include "rgbdfe_internal.h"

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

struct FilterArgs { const float* src; float* dst; int w, h, nf; size_t src_stride, dst_stride; };
template <int FW>
void launch_filter(const FilterArgs& a, const Taps& t, hipStream_t s) {
  if ((size_t)a.w * a.h * a.nf <= (size_t)160 * 120)   // few thousand pixels in the whole launch: small tiles, more workgroups
    hipLaunchKernelGGL((sift_filter_kernel<FW, 16, 16>), dim3((a.w + 15) / 16, (a.h + 15) / 16, a.nf), dim3(256), 0, s, a.src, a.dst,
                       a.w, a.h, t, a.src_stride, a.dst_stride);
  else
    hipLaunchKernelGGL((sift_filter_kernel<FW, 64, 16>), dim3((a.w + 63) / 64, (a.h + 15) / 16, a.nf), dim3(256), 0, s, a.src, a.dst,
                       a.w, a.h, t, a.src_stride, a.dst_stride);
}

void launch_filter_any(const FilterArgs& a, const Taps& t, hipStream_t s) {
  switch (t.fw) {   // ProgramCU::FilterImage's switch over the odd widths 5 .. 33 (ProgramCU.cu:430-448)
    case 5: launch_filter<5>(a, t, s); break;
    case 7: launch_filter<7>(a, t, s); break;
    case 9: launch_filter<9>(a, t, s); break;
    case 11: launch_filter<11>(a, t, s); break;
    case 13: launch_filter<13>(a, t, s); break;
    case 15: launch_filter<15>(a, t, s); break;
    case 17: launch_filter<17>(a, t, s); break;
    case 19: launch_filter<19>(a, t, s); break;
    case 21: launch_filter<21>(a, t, s); break;
    case 23: launch_filter<23>(a, t, s); break;
    case 25: launch_filter<25>(a, t, s); break;
    case 27: launch_filter<27>(a, t, s); break;
    case 29: launch_filter<29>(a, t, s); break;
    case 31: launch_filter<31>(a, t, s); break;
    default: launch_filter<33>(a, t, s); break;
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

// one wave (= one workgroup) per 64 consecutive columns of one row of an OCTAVE, for all its kDogLevels key levels at once:
// the six Gaussian planes the five centre DoG values need are read once per pixel (10 reads when every level had its own
// wave), the common early exit -- |D| <= 0.8 * threshold, which most pixels take -- is decided from those registers, and only a
// level that passes it runs the full test (key_eval: neighbours, edge ratio, sub-pixel solve).  A flag byte per pixel and
// level + the rows' counts.  Most waves end after the six loads, so the launch lives on the number of independent waves in
// flight -- measured on the per-level form: four rows per 256-thread workgroup +50 % (a wave that runs the whole test holds
// the slots of its three finished neighbours), a wave walking 8 rows +50 %.
// Counted = what InitHist_Kernel (ProgramCU.cu:665-688) counts: rows 1 .. h-2, columns 1 .. w-2 with a non-zero key.
__global__ __launch_bounds__(64) void sift_key_flag_kernel(const SiftExtractor::LevelDesc* __restrict__ levels,
                                                           const int* __restrict__ orow2oct, int* __restrict__ rowcnt,
                                                           float dog_threshold0, float dog_threshold, float edge_threshold,
                                                           FrameStrides st) {
  const int oct = orow2oct[2 * blockIdx.y], row = orow2oct[2 * blockIdx.y + 1];
  rowcnt += (size_t)blockIdx.z * st.rows;
  const SiftExtractor::LevelDesc* __restrict__ lv = levels + oct * SiftExtractor::kDogLevels;
  const SiftExtractor::LevelDesc L0 = level_of_frame(lv[0], st, blockIdx.z);
  const int w = L0.w, h = L0.h;
  if ((int)blockIdx.x * 64 >= w) return;
  const int col = blockIdx.x * 64 + threadIdx.x;
  const bool interior = col < w && row > 0 && col > 0 && row < h - 1 && col < w - 1;
  const int index = row * w + col;
  // G[1] .. G[6] at the pixel: level j's centre value is G[j + 2] - G[j + 1]
  float c[SiftExtractor::kDogLevels + 1];
#pragma unroll
  for (int j = 0; j < SiftExtractor::kDogLevels; ++j) c[j] = interior ? (lv[j].g[1] + (size_t)blockIdx.z * st.planes)[index] : 0.f;
  c[SiftExtractor::kDogLevels] =
      interior ? (lv[SiftExtractor::kDogLevels - 1].g[2] + (size_t)blockIdx.z * st.planes)[index] : 0.f;
#pragma unroll
  for (int j = 0; j < SiftExtractor::kDogLevels; ++j) {
    const SiftExtractor::LevelDesc L = level_of_frame(lv[j], st, blockIdx.z);
    int8_t flag = 0;
    if (interior && fabsf(c[j + 1] - c[j]) > dog_threshold0) {
      const KeyEval e = key_eval(L.g, w, index, dog_threshold0, dog_threshold, edge_threshold);
      flag = e.result > 0.f ? 1 : (e.result < 0.f ? -1 : 0);
    }
    if (col < w) L.flags[(size_t)row * w + col] = flag;
    const uint64_t m = __ballot(flag != 0);
    if (threadIdx.x == 0 && m) atomicAdd(&rowcnt[L.row0 + row], (int)__popcll(m));
  }
}

// per level: exclusive scan of its rows' counts, the level's total
__global__ __launch_bounds__(64) void sift_row_scan_kernel(const SiftExtractor::LevelDesc* __restrict__ levels,
                                                           const int* __restrict__ rowcnt, int* __restrict__ rowoff,
                                                           int* __restrict__ lvltot, FrameStrides st) {
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
  if (threadIdx.x == 0) lvltot[blockIdx.x] = base;
}

// one wave per row: the row's extrema in column order -> the level's candidate list (x, y, sign, dx, dy, ds)
__global__ __launch_bounds__(64) void sift_key_emit_kernel(const SiftExtractor::LevelDesc* __restrict__ levels,
                                                           const int* __restrict__ row2lvl, const int* __restrict__ rowcnt,
                                                           const int* __restrict__ rowoff, const int* __restrict__ lvltot,
                                                           float* __restrict__ cand, int cand_cap, float dog_threshold0,
                                                           float dog_threshold, float edge_threshold, FrameStrides st) {
  const int grow = blockIdx.x;
  rowcnt += (size_t)blockIdx.y * st.rows;
  rowoff += (size_t)blockIdx.y * st.rows;
  lvltot += (size_t)blockIdx.y * st.lvltot;
  cand += (size_t)blockIdx.y * st.cand;
  if (rowcnt[grow] == 0) return;
  const int lvl = row2lvl[grow];
  const SiftExtractor::LevelDesc L = level_of_frame(levels[lvl], st, blockIdx.y);
  const int row = grow - L.row0;
  int base = rowoff[grow];
  for (int l = 0; l < lvl; ++l) base += lvltot[l];
  for (int c0 = 0; c0 < L.w; c0 += 64) {
    const int col = c0 + (int)threadIdx.x;
    const bool on = col < L.w && L.flags[(size_t)row * L.w + col] != 0;
    const uint64_t m = __ballot(on);
    if (on) {
      const int rank = base + (int)__popcll(m & (((uint64_t)1 << threadIdx.x) - 1));
      if (rank < cand_cap) {
        const KeyEval e = key_eval(L.g, L.w, row * L.w + col, dog_threshold0, dog_threshold, edge_threshold);
        float* o = cand + (size_t)rank * 6;
        o[0] = (float)col; o[1] = (float)row; o[2] = e.result; o[3] = e.dx; o[4] = e.dy; o[5] = e.ds;
      }
    }
    base += (int)__popcll(m);
  }
}

// ---- gradient of a Gaussian plane at an interior pixel (ComputeDOG_Kernel, ProgramCU.cu:466-473) -------------------------------
__device__ __forceinline__ float2 grad_at(const float* __restrict__ G, int w, int px, int py) {
  const int index = py * w + px;
  const float vxn = G[index + 1], vxp = G[index - 1], vyp = G[index - w], vyn = G[index + w];
  const float dx = vxn - vxp, dy = vyn - vyp;
  const float grd = 0.5f * sqrtf(dx * dx + dy * dy);
  const float rot = (grd == 0.0f ? 0.0f : atan2f(dy, dx));
  return make_float2(grd, rot);
}

struct LevelJobs {  // the kept levels of a frame: consecutive segments of the work list
  int n;
  int base;           // the frame's first item in the batch-wide feature list (outputs) / candidate offsets are absolute
  int begin[65];      // first work item of the segment (begin[n] = total); kMaxOctaves * kDogLevels = 60 segments at most
  int src_off[64];    // candidate / feature offset of the segment's first item
  const float* g[64]; // the Gaussian plane gradients are taken from (G[j + 1] of the level's octave)
  int w[64], h[64];
  float sigma[64];
};

// ComputeOrientation_Kernel (ProgramCU.cu:774-935), num_orientation = 2, sub-pixel on, no existing keypoints.
// The reference runs one THREAD per keypoint through a few hundred samples (gradient + atan2 + exp each): a few dozen
// long waves on a 256-CU chip.  Here a WAVE owns a keypoint: the samples of its window go round-robin (raster order) over
// the 64 lanes, each lane adds into its own column of a [36 bins][64 lanes] LDS histogram (no atomics: deterministic), 36
// lanes then sum their bin's 64 partials in lane order, the 6 smoothing passes are circular 3-tap filters across lanes
// (the reference's in-place loop reads only old values: `one_third * ((pre + v) + next)`, same association), and the
// two-peak selection is the reference's sequential scan on wave-uniform scalars.  Only the ORDER of the weight sums
// differs from the reference (per-lane partial sums): ~1e-7 relative, far inside the libm tolerance of this stage.
__global__ __launch_bounds__(64) void sift_orientation_kernel(const LevelJobs* __restrict__ jobs_of_frame,
                                                              const float* __restrict__ cand, float4* __restrict__ feat,
                                                              float sigma_step, float gaussian_factor, float sample_factor) {
  __shared__ float hist[36][64];
  const float ten_degree_per_radius = 5.7295779513082320876798154814105;
  const LevelJobs& jobs = jobs_of_frame[blockIdx.y];
  const int idx = blockIdx.x;
  if (idx >= jobs.begin[jobs.n]) return;   // the grid is sized for the batch's largest frame
  const int lane = threadIdx.x;
  int s = 0;
  while (s + 1 < jobs.n && idx >= jobs.begin[s + 1]) ++s;
  const int k = idx - jobs.begin[s];
  const float* c = cand + (size_t)(jobs.src_off[s] + k) * 6;
  const int width = jobs.w[s], height = jobs.h[s];
  const float* __restrict__ G = jobs.g[s];
  float4 key;
  key.x = c[0] + 0.5f;
  key.y = c[1] + 0.5f;
  key.z = jobs.sigma[s];
  key.x += c[3];
  key.y += c[4];
  key.z *= powf(sigma_step, c[5]);
  const float gsigma = key.z * gaussian_factor;
  const float win = fabsf(key.z) * sample_factor;
  const float dist_threshold = (float)((double)(win * win) + 0.5);
  const float factor = -0.5f / (gsigma * gsigma);
  const float xmin = fmaxf(1.5f, floorf(key.x - win) + 0.5f);
  const float ymin = fmaxf(1.5f, floorf(key.y - win) + 0.5f);
  const float xmax = fminf(width - 1.5f, floorf(key.x + win) + 0.5f);
  const float ymax = fminf(height - 1.5f, floorf(key.y + win) + 0.5f);
  const int nx = xmax >= xmin ? (int)(xmax - xmin) + 1 : 0;   // iterations of `for (x = xmin; x <= xmax; x += 1.0f)`
  const int ny = ymax >= ymin ? (int)(ymax - ymin) + 1 : 0;
#pragma unroll
  for (int b = 0; b < 36; ++b) hist[b][lane] = 0.0f;
  const int total = nx * ny;
  for (int t = lane; t < total; t += 64) {
    const int iy = t / nx, ix = t - iy * nx;
    const float x = xmin + (float)ix, y = ymin + (float)iy;
    const float dx = x - key.x;
    const float dy = y - key.y;
    const float sq_dist = dx * dx + dy * dy;
    if (sq_dist >= dist_threshold) continue;
    const float2 got = grad_at(G, width, (int)floorf(x), (int)floorf(y));
    const float weight = got.x * expf(sq_dist * factor);
    const float fidx = floorf(got.y * ten_degree_per_radius);
    int oidx = (int)fidx;
    if (oidx < 0) oidx += 36;
    hist[oidx][lane] += weight;
  }
  __syncthreads();
  float v = 0.0f;
  if (lane < 36)
    for (int l = 0; l < 64; ++l) v += hist[lane][l];
  const int lp = lane < 36 ? (lane + 35) % 36 : lane, ln = lane < 36 ? (lane + 1) % 36 : lane;
  const float one_third = 1.0 / 3.0;
  for (int i = 0; i < 6; ++i) {
    const float pre = __shfl(v, lp), next = __shfl(v, ln);
    v = one_third * (pre + v + next);
  }
  float max_vote = lane < 36 ? v : -1.0f;
  for (int d = 32; d >= 1; d >>= 1) max_vote = fmaxf(max_vote, __shfl_xor(max_vote, d));
  const float vote_threshold = max_vote * 0.8f;
  const float pre = __shfl(v, lp), next = __shfl(v, ln);
  const bool peak = lane < 36 && v > vote_threshold && v > pre && v > next;
  const float di = 0.5f * ((next - pre) / (v + v - next - pre));
  const float rot = lane + di + 0.5f;
  const uint64_t peaks = __ballot(peak);
  float max_rot[2] = {0.f, 0.f}, max_vot[2] = {0.f, 0.f};
  int ocount = 0;
#pragma unroll
  for (int i = 0; i < 36; ++i) {   // the reference's scan, on wave-uniform values
    if (!((peaks >> i) & 1)) continue;
    const float weight = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
    const float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rot), i));
    if (weight > max_vot[1]) {
      if (weight > max_vot[0]) {
        max_vot[1] = max_vot[0]; max_rot[1] = max_rot[0];
        max_vot[0] = weight; max_rot[0] = r;
      } else {
        max_vot[1] = weight; max_rot[1] = r;
      }
      ocount++;
    }
  }
  if (lane != 0) return;
  float fr1 = max_rot[0] / 36.0f;
  if (fr1 < 0) fr1 += 1.0f;
  const unsigned short us1 = ocount == 0 ? 65535 : ((unsigned short)floorf(fr1 * 65535.0f));
  unsigned short us2 = 65535;
  if (ocount > 1) {
    float fr2 = max_rot[1] / 36.0f;
    if (fr2 < 0) fr2 += 1.0f;
    us2 = (unsigned short)floorf(fr2 * 65535.0f);
  }
  const unsigned int uspack = ((unsigned int)us2 << 16) | us1;
  key.w = __uint_as_float(uspack);
  feat[jobs.base + idx] = key;
}

// ComputeDescriptor_Kernel<false> (ProgramCU.cu:967-1046).  The reference gives each of a feature's 16 cells one thread;
// here a cell gets a WAVE: the samples of the cell's bounding box go round-robin over the lanes, every lane keeps its own
// 8 + 1 bins in registers (the reference's compare-and-add over k, so no dynamic indexing), a fixed butterfly sums the
// lanes.  Per-sample arithmetic is the reference's; only the order of the sums differs (see the orientation kernel).
__global__ __launch_bounds__(256) void sift_descriptor_kernel(const LevelJobs* __restrict__ jobs_of_frame,
                                                              const float4* __restrict__ feat, float4* __restrict__ d_des,
                                                              float window_factor) {
  const float rpi = 4.0 / 3.14159265358979323846;
  const LevelJobs& jobs = jobs_of_frame[blockIdx.y];
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);      // feature * 16 + cell: four cells (waves) per workgroup
  const int lane = threadIdx.x & 63;
  const int fidx = idx >> 4;
  if (fidx >= jobs.begin[jobs.n]) return;
  int s = 0;
  while (s + 1 < jobs.n && fidx >= jobs.begin[s + 1]) ++s;
  const int width = jobs.w[s], height = jobs.h[s];
  const float* __restrict__ G = jobs.g[s];
  const float4 key = feat[jobs.base + fidx];
  const int bidx = idx & 0xf, ix = bidx & 0x3, iy = bidx >> 2;
  const float spt = fabsf(key.z * window_factor);
  float sn, cs;
  sincosf(key.w, &sn, &cs);
  const float anglef = key.w > 3.14159265358979323846 ? (float)((double)key.w - (2.0 * 3.14159265358979323846)) : key.w;
  const float cspt = cs * spt, sspt = sn * spt;
  const float crspt = cs / spt, srspt = sn / spt;
  float2 offsetpt, pt;
  offsetpt.x = ix - 1.5f;
  offsetpt.y = iy - 1.5f;
  pt.x = cspt * offsetpt.x - sspt * offsetpt.y + key.x;
  pt.y = cspt * offsetpt.y + sspt * offsetpt.x + key.y;
  const float bsz = fabsf(cspt) + fabsf(sspt);
  const float xmin = fmaxf(1.5f, floorf(pt.x - bsz) + 0.5f);
  const float ymin = fmaxf(1.5f, floorf(pt.y - bsz) + 0.5f);
  const float xmax = fminf(width - 1.5f, floorf(pt.x + bsz) + 0.5f);
  const float ymax = fminf(height - 1.5f, floorf(pt.y + bsz) + 0.5f);
  const int nx = xmax >= xmin ? (int)(xmax - xmin) + 1 : 0;
  const int ny = ymax >= ymin ? (int)(ymax - ymin) + 1 : 0;
  float des[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) des[i] = 0.0f;
  const int total = nx * ny;
  for (int t = lane; t < total; t += 64) {
    const int jy = t / nx, jx = t - jy * nx;
    const float x = xmin + (float)jx, y = ymin + (float)jy;
    const float dx = x - pt.x;
    const float dy = y - pt.y;
    const float nxf = crspt * dx + srspt * dy;
    const float nyf = crspt * dy - srspt * dx;
    const float nxn = fabsf(nxf);
    const float nyn = fabsf(nyf);
    if (nxn < 1.0f && nyn < 1.0f) {
      const float2 cc = grad_at(G, width, (int)floorf(x), (int)floorf(y));
      const float dnx = nxf + offsetpt.x;
      const float dny = nyf + offsetpt.y;
      const float ww = expf(-0.125f * (dnx * dnx + dny * dny));
      const float wx = (float)(1.0 - (double)nxn);
      const float wy = (float)(1.0 - (double)nyn);
      const float weight = ww * wx * wy * cc.x;
      float theta = (anglef - cc.y) * rpi;
      if (theta < 0) theta += 8.0f;
      const float fo = floorf(theta);
      const int fi = (int)fo;
      const float weight1 = fo + 1.0f - theta;
      const float weight2 = theta - fo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k == fi) {
          des[k] += (weight1 * weight);
          des[k + 1] += (weight2 * weight);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i)
    for (int d = 32; d >= 1; d >>= 1) des[i] += __shfl_xor(des[i], d);
  if (lane != 0) return;
  des[0] += des[8];
  const int didx = (jobs.base * 16 + idx) << 1;
  d_des[didx] = make_float4(des[0], des[1], des[2], des[3]);
  d_des[didx + 1] = make_float4(des[4], des[5], des[6], des[7]);
}

#define SIFT_HIP(expr)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RGBDFE_ERR_HIP; } \
  } while (0)

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

}  // namespace

SiftExtractor::~SiftExtractor() { release(); }

void SiftExtractor::release() {
  if (d_gray) (void)hipFree(d_gray);
  if (d_input) (void)hipFree(d_input);
  if (d_up) (void)hipFree(d_up);
  if (d_planes) (void)hipFree(d_planes);
  if (d_flags) (void)hipFree(d_flags);
  if (d_rowcnt) (void)hipFree(d_rowcnt);
  if (d_levels) (void)hipFree(d_levels);
  if (d_cand) (void)hipFree(d_cand);
  if (d_feat) (void)hipFree(d_feat);
  if (d_desc) (void)hipFree(d_desc);
  if (d_jobs) (void)hipFree(d_jobs);
  if (d_orow2oct) (void)hipFree(d_orow2oct);
  d_orow2oct = nullptr;
  if (h_counts) (void)hipHostFree(h_counts);
  if (h_stage) (void)hipHostFree(h_stage);
  if (h_gray) (void)hipHostFree(h_gray);
  if (h_jobs) (void)hipHostFree(h_jobs);
  if (h_desc) (void)hipHostFree(h_desc);
  h_desc = nullptr; h_desc_cap = 0;
  d_gray = nullptr; d_input = d_up = d_planes = nullptr; d_flags = nullptr; d_rowcnt = d_rowoff = d_lvltot = nullptr;
  d_levels = nullptr; d_cand = nullptr; d_feat = nullptr; d_desc = nullptr; h_counts = nullptr; h_stage = nullptr;
  d_jobs = nullptr; h_jobs = nullptr;
  h_gray = nullptr; gray_cap = 0; stage_floats = 0; cand_cap = feat_cap = desc_cap = 0; W = H = 0; frames_cap = 0;
}

void SiftExtractor::init_params() {  // SiftParam::ParseSiftParam (SiftGPU.cpp:433-473) with "-d 5 -e 10.0"
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

float SiftExtractor::initial_smooth_sigma(int om) const {
  const float sa = sigma0 * powf(2.0f, float(-1) / float(kDogLevels));
  const float sb = 0.5f / powf(2.0f, float(om));
  return sa > sb + 0.001 ? sqrtf(sa * sa - sb * sb) : 0.0f;
}

float SiftExtractor::level_sigma(int lev) const { return sigma0 * powf(2.0f, float(lev) / float(kDogLevels)); }

// PyramidCU::InitPyramid / ResizePyramid / FitPyramid (PyramidCU.cpp:86-306): the geometry a frame of this size gets;
// every buffer holds nf frames side by side
int SiftExtractor::prepare(int rows, int cols, int nf, std::string& err) {
  init_params();
  if (rows == H && cols == W && d_planes && nf <= frames_cap) return RGBDFE_OK;
  const int tw = cols & 0xfffffffc;  // GLTexInput::TruncateWidthCU (GLTexImage.h:125)
  if (tw < 16 || rows < 16) { err = "image too small for SIFT extraction"; return RGBDFE_ERR_INVALID_ARG; }
  int om = -1;  // "-fo -1"
  int wp = tw << 1, hp = rows << 1;
  while (wp > 3200 || hp > 3200) { om++; wp >>= 1; hp >>= 1; }  // GlobalUtil::_texMaxDim (GlobalUtil.cpp:86)
  if (om > 0) { err = "images beyond 3200 x 3200 pixels are not supported"; return RGBDFE_ERR_CAPACITY; }
  int on = (int)floor(log(double(std::min(wp, hp))) / log(2.0)) - 3;
  if (on < 1) on = 1;
  if (on > kMaxOctaves) on = kMaxOctaves;
  if (rows == H && cols == W && nf < frames_cap) nf = frames_cap;
  release();
  W = cols; H = rows; w4 = tw; octave_min = om; octave_num = on;
  size_t total = 0;
  int w = wp, h = hp;
  total_rows = 0;
  for (int i = 0; i < on; ++i) {
    oct[i].w = ((w + 3) / 4) * 4;
    oct[i].h = h;
    oct[i].plane = (size_t)oct[i].w * h;
    total += oct[i].plane * kLevels;
    total_rows += h * kDogLevels;
    w >>= 1; h >>= 1;
  }
  planes_floats = total;
  input_floats = (size_t)tw * rows;
  const size_t F = (size_t)nf;
  SIFT_HIP(hipMalloc((void**)&d_gray, F * rows * cols));
  SIFT_HIP(hipMalloc((void**)&d_input, F * input_floats * 4));
  SIFT_HIP(hipMalloc((void**)&d_up, F * oct[0].plane * 4));
  SIFT_HIP(hipMalloc((void**)&d_planes, F * total * 4));
  size_t off = 0, foff = 0;
  for (int i = 0; i < on; ++i)
    for (int l = 0; l < kLevels; ++l) { oct[i].g[l] = d_planes + off; off += oct[i].plane; }   // frame 0's planes
  for (int i = 0; i < on; ++i) foff += oct[i].plane * kDogLevels;
  flags_bytes = foff;
  SIFT_HIP(hipMalloc((void**)&d_flags, F * flags_bytes));
  // rowcnt [nf][total_rows] | rowoff [nf][total_rows] | row2lvl [total_rows] | lvltot [nf][64]
  SIFT_HIP(hipMalloc((void**)&d_rowcnt, sizeof(int) * ((size_t)total_rows * (2 * F + 1) + 64 * F)));
  d_rowoff = d_rowcnt + (size_t)total_rows * F;
  d_lvltot = d_rowcnt + (size_t)total_rows * (2 * F + 1);
  h_levels.assign((size_t)on * kDogLevels, LevelDesc{});
  std::vector<int> row2lvl((size_t)total_rows);
  int row0 = 0;
  foff = 0;
  for (int i = 0; i < on; ++i)
    for (int j = 0; j < kDogLevels; ++j) {
      LevelDesc& L = h_levels[(size_t)i * kDogLevels + j];
      const int l = j + 2;  // key level: DoG l - 1, l, l + 1 = Gaussian l - 2 .. l + 1
      for (int k = 0; k < 4; ++k) L.g[k] = oct[i].g[l - 2 + k];
      L.flags = d_flags + foff;
      L.w = oct[i].w; L.h = oct[i].h; L.row0 = row0;
      for (int r = 0; r < oct[i].h; ++r) row2lvl[(size_t)row0 + r] = i * kDogLevels + j;
      row0 += oct[i].h;
      foff += oct[i].plane;
    }
  SIFT_HIP(hipMalloc((void**)&d_levels, sizeof(LevelDesc) * h_levels.size()));
  SIFT_HIP(hipMemcpy(d_levels, h_levels.data(), sizeof(LevelDesc) * h_levels.size(), hipMemcpyHostToDevice));
  SIFT_HIP(hipMemcpy(d_rowcnt + (size_t)total_rows * 2 * F, row2lvl.data(), sizeof(int) * (size_t)total_rows, hipMemcpyHostToDevice));
  {  // (octave, row) of every row of the stacked octaves: the keypoint scan's launch walks it
    std::vector<int> o2((size_t)total_rows / kDogLevels * 2);
    size_t k = 0;
    for (int i = 0; i < on; ++i)
      for (int r = 0; r < oct[i].h; ++r) { o2[k++] = i; o2[k++] = r; }
    SIFT_HIP(hipMalloc((void**)&d_orow2oct, sizeof(int) * o2.size()));
    SIFT_HIP(hipMemcpy(d_orow2oct, o2.data(), sizeof(int) * o2.size(), hipMemcpyHostToDevice));
  }
  cand_cap = std::max<size_t>((size_t)1 << 16, oct[0].plane / 16);
  SIFT_HIP(hipMalloc((void**)&d_cand, F * cand_cap * 6 * 4));
  feat_cap = cand_cap * 2;                      // per frame; the batch-wide lists are packed: F * feat_cap entries at most
  SIFT_HIP(hipMalloc((void**)&d_feat, F * feat_cap * 16));
  SIFT_HIP(hipHostMalloc((void**)&h_counts, sizeof(int) * 64 * F, hipHostMallocDefault));
  stage_floats = F * cand_cap * 8;
  SIFT_HIP(hipHostMalloc((void**)&h_stage, stage_floats * 4, hipHostMallocDefault));
  gray_cap = (size_t)rows * cols;
  SIFT_HIP(hipHostMalloc((void**)&h_gray, F * gray_cap, hipHostMallocDefault));
  SIFT_HIP(hipMalloc((void**)&d_jobs, sizeof(LevelJobs) * F));
  SIFT_HIP(hipHostMalloc((void**)&h_jobs, sizeof(LevelJobs) * F, hipHostMallocDefault));
  frames_cap = nf;
  return RGBDFE_OK;
}

// images in (GLTexInput::SetImageData, CUDA branch, GLTexImage.cpp:971-1009) + BuildPyramid (PyramidCU.cpp:946-998) for nf frames
int SiftExtractor::enqueue_pyramid(const uint8_t* const* gray, int nf, hipStream_t s, std::string& err) {
  const int rows = H, cols = W;
  const unsigned NF = (unsigned)nf;
  for (int f = 0; f < nf; ++f) memcpy(h_gray + (size_t)f * gray_cap, gray[f], gray_cap);
  SIFT_HIP(hipMemcpyAsync(d_gray, h_gray, (size_t)nf * gray_cap, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(sift_convert_kernel, dim3((w4 * rows + 255) / 256, NF), dim3(256), 0, s, d_gray, cols, w4, rows, d_input);
  auto filter = [&](const float* src, size_t src_stride, float* dst, int w, int h, float sg) {
    launch_filter_any(FilterArgs{src, dst, w, h, nf, src_stride, planes_floats}, make_taps(sg), s);
  };
  for (int i = 0; i < octave_num; ++i) {
    const Octave& o = oct[i];
    if (i == 0) {
      const float sg = initial_smooth_sigma(octave_min);
      if (octave_min < 0) {  // SampleImageU + FilterImage in place through the buffer plane
        hipLaunchKernelGGL(sift_upsample2_kernel, dim3((w4 + 127) / 128, rows << 1, NF), dim3(128), 0, s, d_input, w4, rows, d_up,
                           o.plane);
        filter(d_up, o.plane, o.g[0], o.w, o.h, sg);
      } else {
        filter(d_input, input_floats, o.g[0], o.w, o.h, sg);
      }
    } else {  // SampleImageD from level_ds of the octave below (index level_ds - level_min = 5); sigma_skip1 = 0
      const Octave& p = oct[i - 1];
      hipLaunchKernelGGL(sift_downsample2_kernel, dim3((o.w + 127) / 128, o.h, NF), dim3(128), 0, s, p.g[kDogLevels], p.w, o.w, o.h,
                         o.g[0], planes_floats);
    }
    for (int l = 1; l < kLevels; ++l) filter(o.g[l - 1], planes_floats, o.g[l], o.w, o.h, sigma[l - 1]);
  }
  SIFT_HIP(hipGetLastError());
  return RGBDFE_OK;
}

int SiftExtractor::run_batch(const uint8_t* const* gray, int nf, int rows, int cols, int max_features, std::vector<SiftKey>* keys,
                             const float** desc, hipStream_t s, std::string& err) {
  const int rc = begin_batch(gray, nf, rows, cols, s, err);
  return rc != RGBDFE_OK ? rc : finish_batch(max_features, keys, desc, s, err);
}

// The first, shape-static half of a batch -- images in, pyramids, extremum flags, ordered candidate lists, the per-level
// counts on their way to the host -- is only ENQUEUED here; finish_batch waits for it.  The batch entry point runs two
// extractors alternately, each on its own stream, so that this half of chunk k + 1 executes while the host works through the
// second half of chunk k (its three waits, the feature-count limits, the list reshaping).
int SiftExtractor::begin_batch(const uint8_t* const* gray, int nf, int rows, int cols, hipStream_t s, std::string& err) {
  if (nf < 1 || nf > kMaxBatch) { err = "SIFT batch size out of range"; return RGBDFE_ERR_INVALID_ARG; }
  pending_nf = 0;
  int rc = prepare(rows, cols, nf, err);
  if (rc != RGBDFE_OK) return rc;
  const int nlv = octave_num * kDogLevels;
  const unsigned NF = (unsigned)nf;
  FrameStrides st{};
  st.planes = planes_floats; st.flags = flags_bytes; st.cand = cand_cap * 6; st.rows = total_rows; st.lvltot = 64;
  rc = enqueue_pyramid(gray, nf, s, err);
  if (rc != RGBDFE_OK) return rc;
  // ---- DetectKeypointsEX + the list part of GenerateFeatureList: flags, row counts, scan, ordered emit ----------------------
  const float tdog = dog_threshold, tdog1 = 0.8f * tdog;
  const float tedge = (edge_threshold + 1) * (edge_threshold + 1) / edge_threshold;
  int* d_row2lvl = d_rowcnt + (size_t)total_rows * 2 * frames_cap;
  st.rows = total_rows;
  SIFT_HIP(hipMemsetAsync(d_rowcnt, 0, sizeof(int) * (size_t)total_rows * nf, s));
  hipLaunchKernelGGL(sift_key_flag_kernel, dim3((oct[0].w + 63) / 64, total_rows / kDogLevels, NF), dim3(64),
                     0, s, d_levels, d_orow2oct, d_rowcnt, tdog1, tdog, tedge, st);
  hipLaunchKernelGGL(sift_row_scan_kernel, dim3(nlv, NF), dim3(64), 0, s, d_levels, d_rowcnt, d_rowoff, d_lvltot, st);
  hipLaunchKernelGGL(sift_key_emit_kernel, dim3(total_rows, NF), dim3(64), 0, s, d_levels, d_row2lvl, d_rowcnt, d_rowoff, d_lvltot,
                     d_cand, (int)cand_cap, tdog1, tdog, tedge, st);
  SIFT_HIP(hipGetLastError());
  SIFT_HIP(hipMemcpyAsync(h_counts, d_lvltot, sizeof(int) * 64 * (size_t)nf, hipMemcpyDeviceToHost, s));
  pending_nf = nf;
  return RGBDFE_OK;
}

int SiftExtractor::finish_batch(int max_features, std::vector<SiftKey>* keys, const float** desc, hipStream_t s, std::string& err) {
  const int nf = pending_nf;
  if (nf < 1) { err = "finish_batch without begin_batch"; return RGBDFE_ERR_INVALID_ARG; }
  pending_nf = 0;
  for (int f = 0; f < nf; ++f) { keys[f].clear(); desc[f] = nullptr; }
  const int nlv = octave_num * kDogLevels;
  const unsigned NF = (unsigned)nf;
  SIFT_HIP(hipStreamSynchronize(s));
  // ---- per frame: which levels run -- GenerateFeatureList's "-tc2" order (coarse octaves first, PyramidCU.cpp:797-850) and
  //      SiftPyramid::LimitFeatureCount(0) (SiftPyramid.cpp:170-210, _TruncateMethod = 1).  A skipped level contributes
  //      nothing (the reference leaves the previous frame's list in it: DESIGN.md 4.11) ---------------------------------
  struct FrameState {
    std::vector<int> cnt, off, level_num;
    int feature_num = 0, total = 0, base = 0, erased = 0;
    std::vector<float> list, keybuf;
  };
  std::vector<FrameState> fs((size_t)nf);
  auto limit = [&](FrameState& F) {
    if (max_features <= 0) return 0;
    int i = 0, erased = 0;
    while (i < nlv && F.feature_num - F.level_num[(size_t)i] > max_features) {
      erased += F.level_num[(size_t)i];
      F.feature_num -= F.level_num[(size_t)i];
      F.level_num[(size_t)i++] = 0;
    }
    return erased;
  };
  LevelJobs* hj = static_cast<LevelJobs*>(h_jobs);
  int grand = 0, max_total = 0;
  for (int f = 0; f < nf; ++f) {
    FrameState& F = fs[(size_t)f];
    F.cnt.assign(h_counts + (size_t)f * 64, h_counts + (size_t)f * 64 + nlv);
    F.off.assign((size_t)nlv + 1, 0);
    for (int i = 0; i < nlv; ++i) F.off[(size_t)i + 1] = F.off[(size_t)i] + F.cnt[(size_t)i];
    if ((size_t)F.off[(size_t)nlv] > cand_cap) { err = "more SIFT keypoint candidates than the candidate buffer holds"; return RGBDFE_ERR_CAPACITY; }
    F.level_num.assign((size_t)nlv, 0);
    for (int i = octave_num - 1; i >= 0; --i)
      for (int j = kDogLevels - 1; j >= 0; --j) {
        if (max_features > 0 && F.feature_num > max_features) continue;
        F.level_num[(size_t)i * kDogLevels + j] = F.cnt[(size_t)i * kDogLevels + j];
        F.feature_num += F.cnt[(size_t)i * kDogLevels + j];
      }
    limit(F);
    // ---- GetFeatureOrientations (PyramidCU.cpp:1145-1172): the frame's segment table --------------------------------------
    LevelJobs& jobs = hj[f];
    memset(&jobs, 0, sizeof(jobs));
    int total = 0;
    for (int idx = 0; idx < nlv; ++idx) {
      if (F.level_num[(size_t)idx] <= 0) continue;
      const int i = idx / kDogLevels, j = idx % kDogLevels;
      const int n = jobs.n++;
      jobs.begin[n] = total;
      jobs.src_off[n] = (int)((size_t)f * cand_cap) + F.off[(size_t)idx];
      jobs.g[n] = oct[i].g[j + 1] + (size_t)f * planes_floats;
      jobs.w[n] = oct[i].w; jobs.h[n] = oct[i].h;
      jobs.sigma[n] = level_sigma(j);  // GetLevelSigma(j + level_min + 1)
      total += F.level_num[(size_t)idx];
    }
    jobs.begin[jobs.n] = total;
    jobs.base = grand;
    F.total = total; F.base = grand;
    grand += total;
    max_total = std::max(max_total, total);
  }
  lvl_count = fs[0].cnt;
  lvl_off = fs[0].off;
  if (grand == 0) return RGBDFE_OK;
  if ((size_t)grand * 4 > stage_floats) { err = "SIFT staging buffer too small"; return RGBDFE_ERR_CAPACITY; }
  const float sigma_step = powf(2.0f, 1.0f / kDogLevels);
  SIFT_HIP(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LevelJobs) * (size_t)nf, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(sift_orientation_kernel, dim3(max_total, NF), dim3(64), 0, s, static_cast<const LevelJobs*>(d_jobs), d_cand, d_feat,
                     sigma_step, 1.5f, 1.5f * 2.0f);
  SIFT_HIP(hipGetLastError());
  SIFT_HIP(hipMemcpyAsync(h_stage, d_feat, (size_t)grand * 16, hipMemcpyDeviceToHost, s));
  SIFT_HIP(hipStreamSynchronize(s));
  // ---- ReshapeFeatureListCPU (PyramidCU.cpp:501-585, NO_DUPLICATE_DOWNLOAD) + LimitFeatureCount(1), per frame ------------------
  const double twopi = 2.0 * 3.14159265358979323846;
  const double factor = 2.0 * 3.14159265358979323846 / 65535.0;
  const float os = octave_min >= 0 ? float(1 << octave_min) : 1.0f / (1 << (-octave_min));
  int grand2 = 0, max_total2 = 0;
  for (int f = 0; f < nf; ++f) {
    FrameState& F = fs[(size_t)f];
    const LevelJobs& jobs = hj[f];
    F.list.reserve((size_t)F.total * 8);     // final feature list in level coordinates (x, y, scale, orientation)
    F.keybuf.reserve((size_t)F.total * 8);   // image coordinates
    F.feature_num = 0;
    int seg = 0;
    for (int idx = 0; idx < nlv; ++idx) {
      if (F.level_num[(size_t)idx] <= 0) continue;
      const float* src = h_stage + ((size_t)F.base + jobs.begin[seg]) * 4;
      const int cnt = F.level_num[(size_t)idx];
      int fcount = 0;
      const float oss = os * (1 << (idx / kDogLevels));
      for (int k = 0; k < cnt; ++k, src += 4) {
        unsigned short orientations[2];
        memcpy(orientations, &src[3], 4);
        auto push = [&](unsigned short o) {
          const float fo = float(factor * o);
          F.list.push_back(src[0]); F.list.push_back(src[1]); F.list.push_back(src[2]); F.list.push_back(fo);
          F.keybuf.push_back(oss * (src[0] - 0.5f) + 0.5f);
          F.keybuf.push_back(oss * (src[1] - 0.5f) + 0.5f);
          F.keybuf.push_back(oss * src[2]);
          F.keybuf.push_back((float)fmod(twopi - fo, twopi));
          fcount++;
        };
        if (orientations[0] != 65535) {
          push(orientations[0]);
          if (orientations[1] != 65535 && orientations[1] != orientations[0]) push(orientations[1]);
        }
      }
      F.level_num[(size_t)idx] = fcount;
      F.feature_num += fcount;
      ++seg;
    }
    F.erased = limit(F);
  }
  // ---- GetFeatureDescriptors (PyramidCU.cpp:393-432) ---------------------------------------------------------------------------
  for (int f = 0; f < nf; ++f) {
    FrameState& F = fs[(size_t)f];
    LevelJobs& dj = hj[f];
    memset(&dj, 0, sizeof(dj));
    int total = 0;
    for (int idx = 0; idx < nlv; ++idx) {
      if (F.level_num[(size_t)idx] <= 0) continue;
      const int i = idx / kDogLevels, j = idx % kDogLevels;
      const int n = dj.n++;
      dj.begin[n] = total;
      dj.g[n] = oct[i].g[j + 1] + (size_t)f * planes_floats;
      dj.w[n] = oct[i].w; dj.h[n] = oct[i].h;
      total += F.level_num[(size_t)idx];
    }
    dj.begin[dj.n] = total;
    dj.base = grand2;
    F.total = total; F.base = grand2;
    grand2 += total;
    max_total2 = std::max(max_total2, total);
  }
  if (grand2 == 0) return RGBDFE_OK;
  if ((size_t)grand2 > feat_cap * (size_t)frames_cap) { err = "more SIFT features than the feature buffer holds"; return RGBDFE_ERR_CAPACITY; }
  if ((size_t)grand2 * 4 > stage_floats) { err = "SIFT staging buffer too small"; return RGBDFE_ERR_CAPACITY; }
  if ((size_t)grand2 * 128 > desc_cap) {
    if (d_desc) (void)hipFree(d_desc);
    d_desc = nullptr; desc_cap = 0;
    SIFT_HIP(hipMalloc((void**)&d_desc, (size_t)grand2 * 128 * 4 * 2));
    desc_cap = (size_t)grand2 * 128 * 2;
  }
  for (int f = 0; f < nf; ++f) {
    const FrameState& F = fs[(size_t)f];
    if (F.total > 0) memcpy(h_stage + (size_t)F.base * 4, F.list.data() + (size_t)F.erased * 4, (size_t)F.total * 16);
  }
  SIFT_HIP(hipMemcpyAsync(d_feat, h_stage, (size_t)grand2 * 16, hipMemcpyHostToDevice, s));
  SIFT_HIP(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LevelJobs) * (size_t)nf, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(sift_descriptor_kernel, dim3(max_total2 * 4, NF), dim3(256), 0, s, static_cast<const LevelJobs*>(d_jobs), d_feat,
                     (float4*)d_desc, 3.0f);
  SIFT_HIP(hipGetLastError());
  if ((size_t)grand2 * 128 > h_desc_cap) {
    if (h_desc) (void)hipHostFree(h_desc);
    h_desc = nullptr; h_desc_cap = 0;
    SIFT_HIP(hipHostMalloc((void**)&h_desc, (size_t)grand2 * 128 * 4 * 2, hipHostMallocDefault));
    h_desc_cap = (size_t)grand2 * 128 * 2;
  }
  SIFT_HIP(hipMemcpyAsync(h_desc, d_desc, (size_t)grand2 * 128 * 4, hipMemcpyDeviceToHost, s));
  SIFT_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < nf; ++f) {
    const FrameState& F = fs[(size_t)f];
    desc[f] = h_desc + (size_t)F.base * 128;
    keys[f].resize((size_t)F.total);
    if (F.total > 0) memcpy(keys[f].data(), F.keybuf.data() + (size_t)F.erased * 4, (size_t)F.total * 16);
  }
  return RGBDFE_OK;
}

// SiftGPUWrapper::detect with a caller-provided keypoint list (sift_gpu_wrapper.cpp:132-142): SiftGPU::SetKeypointList(num,
// keys) with its default keys_have_orientation = 1 (SiftGPU.h:150) = SIFT_SKIP_DETECTION | SIFT_SKIP_ORIENTATION
// (SiftPyramid.cpp:244-261): the pyramid is built, every keypoint is assigned to the (octave, level) whose scale band holds its
// scale (PyramidCU::GenerateFeatureListTex, :434-498: level coordinates, orientation mirrored), descriptors are computed
// at the given positions / scales / orientations and put back into the callers' order (GetFeatureDescriptors, :393-432).
// keys_in: n x (x, y, scale, orientation in radians); desc: n x 128 in a pinned buffer of this object.
int SiftExtractor::describe(const uint8_t* gray, int rows, int cols, const SiftKey* keys_in, int n, const float** desc, hipStream_t s,
                            std::string& err) {
  *desc = nullptr;
  int rc = prepare(rows, cols, 1, err);
  if (rc != RGBDFE_OK) return rc;
  if (n <= 0) return RGBDFE_OK;
  rc = enqueue_pyramid(&gray, 1, s, err);
  if (rc != RGBDFE_OK) return rc;
  const int nlv = octave_num * kDogLevels;
  const double twopi = 2.0 * 3.14159265358979323846;
  const float sigma_half_step = powf(2.0f, 0.5f / kDogLevels);
  float octave_sigma = octave_min >= 0 ? float(1 << octave_min) : 1.0f / (1 << (-octave_min));
  const float offset = 0.5f;   // GlobalUtil::_LoweOrigin = 0
  std::vector<float> list;     // level coordinates, level by level
  std::vector<int> index;      // _keypoint_index: the input position of every list entry
  LevelJobs* dj = static_cast<LevelJobs*>(h_jobs);
  memset(dj, 0, sizeof(LevelJobs));
  int total = 0;
  for (int i = 0; i < octave_num; ++i, octave_sigma *= 2.0f)
    for (int j = 0; j < kDogLevels; ++j) {
      const float level_sg = level_sigma(j) * octave_sigma;   // GetLevelSigma(j + level_min + 1)
      const float sigma_min = level_sg / sigma_half_step, sigma_max = level_sg * sigma_half_step;
      int fcount = 0;
      for (int k = 0; k < n; ++k) {
        const float sigmak = keys_in[k].s;
        if ((sigmak >= sigma_min && sigmak < sigma_max) || (sigmak < sigma_min && i == 0 && j == 0) ||
            (sigmak > sigma_max && i == octave_num - 1 && j == kDogLevels - 1)) {
          list.push_back((keys_in[k].x - offset) / octave_sigma + 0.5f);
          list.push_back((keys_in[k].y - offset) / octave_sigma + 0.5f);
          list.push_back(keys_in[k].s / octave_sigma);
          list.push_back((float)fmod(twopi - keys_in[k].o, twopi));
          index.push_back(k);
          ++fcount;
        }
      }
      if (fcount == 0) continue;
      const int m = dj->n++;
      dj->begin[m] = total;
      dj->g[m] = oct[i].g[j + 1];
      dj->w[m] = oct[i].w; dj->h[m] = oct[i].h;
      total += fcount;
    }
  (void)nlv;
  dj->begin[dj->n] = total;
  dj->base = 0;
  // (a scale exactly on a band's edge can satisfy two bands or none in float arithmetic: the reference then lists the keypoint
  // twice -- and overruns its buffers -- or not at all; here a keypoint keeps the LAST band that took it, one without a band a
  // zero descriptor)
  if ((size_t)total > feat_cap || (size_t)total * 4 > stage_floats) { err = "more SIFT keypoints than the feature buffer holds"; return RGBDFE_ERR_CAPACITY; }
  const size_t rows_out = (size_t)std::max(total, n);
  if (rows_out * 128 > desc_cap) {
    if (d_desc) (void)hipFree(d_desc);
    d_desc = nullptr; desc_cap = 0;
    SIFT_HIP(hipMalloc((void**)&d_desc, rows_out * 128 * 4 * 2));
    desc_cap = rows_out * 128 * 2;
  }
  if ((rows_out + (size_t)n) * 128 > h_desc_cap) {
    if (h_desc) (void)hipHostFree(h_desc);
    h_desc = nullptr; h_desc_cap = 0;
    SIFT_HIP(hipHostMalloc((void**)&h_desc, (rows_out + (size_t)n) * 128 * 4 * 2, hipHostMallocDefault));
    h_desc_cap = (rows_out + (size_t)n) * 128 * 2;
  }
  float* ordered = h_desc + rows_out * 128;   // the second half of the pinned buffer: the callers' order
  memset(ordered, 0, (size_t)n * 128 * 4);
  if (total > 0) {
    memcpy(h_stage, list.data(), (size_t)total * 16);
    SIFT_HIP(hipMemcpyAsync(d_feat, h_stage, (size_t)total * 16, hipMemcpyHostToDevice, s));
    SIFT_HIP(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LevelJobs), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(sift_descriptor_kernel, dim3(total * 4, 1), dim3(256), 0, s, static_cast<const LevelJobs*>(d_jobs), d_feat,
                       (float4*)d_desc, 3.0f);
    SIFT_HIP(hipGetLastError());
    SIFT_HIP(hipMemcpyAsync(h_desc, d_desc, (size_t)total * 128 * 4, hipMemcpyDeviceToHost, s));
    SIFT_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < total; ++i) memcpy(ordered + (size_t)index[(size_t)i] * 128, h_desc + (size_t)i * 128, 128 * 4);
  }
  *desc = ordered;
  return RGBDFE_OK;
}

int SiftExtractor::debug_plane(int octave, int level, std::vector<float>& out, int* w, int* h, hipStream_t s) {
  if (octave < 0 || octave >= octave_num || level < 0 || level >= kLevels || !d_planes) return RGBDFE_ERR_INVALID_ARG;
  out.resize(oct[octave].plane);
  *w = oct[octave].w; *h = oct[octave].h;
  if (hipMemcpyAsync(out.data(), oct[octave].g[level], oct[octave].plane * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return RGBDFE_ERR_HIP;
  return RGBDFE_OK;
}

int SiftExtractor::debug_candidates(int octave, int dog_level, std::vector<float>& out) {
  if (octave < 0 || octave >= octave_num || dog_level < 0 || dog_level >= kDogLevels || lvl_count.empty())
    return RGBDFE_ERR_INVALID_ARG;
  const int idx = octave * kDogLevels + dog_level;
  out.resize((size_t)lvl_count[(size_t)idx] * 6);
  if (!out.empty() &&
      hipMemcpy(out.data(), d_cand + (size_t)lvl_off[(size_t)idx] * 6, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return RGBDFE_ERR_HIP;
  return RGBDFE_OK;
}

}  // namespace rgbdfe
