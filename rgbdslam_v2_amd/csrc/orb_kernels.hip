// orb_kernels.hip -- ORB detect / describe kernels for gfx950 (rows a1, a6 of SURVEY.md section 8).
//
// Device side of cv::ORB as the reference uses it:
//   detect : ORB::create(10000, 1.2f, 8, 15, 0, 2, HARRIS_SCORE, 31, thr)->detect(sub_image, kps, sub_mask)
//            per grid cell (src/feature_adjuster.cpp:94, 286-317)
//   compute: ORB::create()->compute(gray, kps, desc)                    (src/features.cpp:117-119, node.cpp:202)
// OpenCV is not part of the reference tree: the arithmetic restated here is the published OpenCV 3.3
// algorithm (features2d/src/orb.cpp, fast.cpp, fast_score.cpp; imgproc/src/resize.cpp, smooth.cpp) and is
// tested for exact equality against oracle/orb_oracle.c ("parity unpinned" against a real OpenCV).
//
// All images of one step (9 grid cells x 8 pyramid levels, plus the full-frame pyramid) are described by
// ImgDesc records and processed by ONE launch per stage: a frame is ~1.7 M pixels, far too little to fill
// 256 CUs unless every (cell, level) image rides in the same grid.  Pixel kernels are streaming stencils
// (HBM/L2 bound): one byte per lane, rows contiguous across lanes.  Keypoint lists are compacted in raster
// order with __ballot / mbcnt prefixes (row counts -> block scan -> emit), so the order of keypoints is
// the one cv::FAST produces; per-keypoint measurements (Harris response, intensity-centroid angle, rBRIEF)
// use one wave per keypoint with shuffle reductions.
#include <stdlib.h>
#include <string.h>

#include "orb_internal.h"

namespace rgbdfe {

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// ------------------------------------------------------------------------------------------------
// cv::resize(..., INTER_LINEAR) for 8-bit images (fixed point, INTER_RESIZE_COEF_BITS = 11), one
// pyramid level for every job in the launch; the mask variant applies threshold(254, TOZERO).
// ------------------------------------------------------------------------------------------------
// (every 2-D stage walks a host-built list of (image, tile) units: the images of a step differ in size by a factor of 13,
// a grid sized for the largest one would be mostly empty workgroups)
// A workgroup owns 64 x 16 destination pixels; a thread a column of four rows (the horizontal coefficients are formed once).
constexpr int kResizeTH = 16;
__global__ __launch_bounds__(256) void orb_resize_kernel(uint8_t* __restrict__ pool, const ResizeJob* __restrict__ jobs,
                                                         const TileUnit* __restrict__ units) {
  const TileUnit u = units[blockIdx.x];
  const ResizeJob j = jobs[u.img];
  const int dx = u.bx * 64 + (threadIdx.x & 63);
  const int dy0 = u.by * kResizeTH + (threadIdx.x >> 6) * 4;
  if (dx >= j.dw || dy0 >= j.dh) return;
  const uint8_t* __restrict__ src = pool + j.src_off;
  float fx = (float)((dx + 0.5) * j.scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  const bool edge = (sx + 1 >= j.sw);
  if (sx >= j.sw - 1) { fx = 0; sx = j.sw - 1; }
  const int a0 = max(min(__float2int_rn((1.f - fx) * 2048), 32767), -32768);
  const int a1 = max(min(__float2int_rn(fx * 2048), 32767), -32768);
  const int sx1 = edge ? sx : sx + 1;          // at the right edge the second tap is not read: a0 = 2048, a1 = 0 there
  const int c1 = edge ? 0 : a1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int dy = dy0 + t;
    if (dy >= j.dh) break;
    float fy = (float)((dy + 0.5) * j.scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = max(min(__float2int_rn((1.f - fy) * 2048), 32767), -32768);
    const int b1 = max(min(__float2int_rn(fy * 2048), 32767), -32768);
    const int y0 = min(max(sy, 0), j.sh - 1), y1 = min(max(sy + 1, 0), j.sh - 1);
    const uint8_t* r0 = src + (size_t)y0 * j.sstride;
    const uint8_t* r1 = src + (size_t)y1 * j.sstride;
    const int h0 = r0[sx] * (edge ? 2048 : a0) + r0[sx1] * c1;
    const int h1 = r1[sx] * (edge ? 2048 : a0) + r1[sx1] * c1;
    int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    if (j.is_mask && v <= 254) v = 0;
    pool[j.dst_off + (size_t)dy * j.dw + dx] = (uint8_t)v;
  }
}

// ------------------------------------------------------------------------------------------------
// The seven resize steps of every image chain of a step in ONE launch (round 4, VERDICT r3 #5a).  A level is the bilinear
// resize of the level below it, so seven launches form a chain of dependent round trips through memory (a quarter of the
// kernel time of a super-frame).  Here a workgroup follows ONE tile of one chain through all seven levels: level l of the
// tile is computed from level l - 1 held in LDS (the tile's part of the uploaded image is copied there first, whole dwords,
// every load in flight at once), stored to the pool where the tile owns
// it, and kept in LDS -- with the rim the deeper levels of the tile read, which neighbouring workgroups compute again
// (a pixel of level l is the same integer expression of the same four pixels of level l - 1 wherever it is evaluated).
// The regions are planned on the host (OrbWorkspace::plan_pyramid) with resize_tap_x / resize_tap_y, the functions the
// coefficient tables below are filled with: per region one table entry per column and per row instead of the double
// arithmetic per pixel.
// ------------------------------------------------------------------------------------------------
// kPyrWaves waves share a tile's pixels level by level.  Measured in one call (640x480, 7 frames per launch, us per frame:
// profiles/r05_logs/pyr_waves_ab*.log): 1 wave 24.7, 2 waves 17.2, 4 waves 9.4-9.8, 8 waves 13.8, 16 waves 19.0 -- fewer waves leave
// the chain of seven dependent levels too long, more waves spend their time in the level loops' bookkeeping and barriers.
#ifndef RGBDFE_PYR_WAVES
#define RGBDFE_PYR_WAVES 4
#endif
constexpr int kPyrWaves = RGBDFE_PYR_WAVES, kPyrThreads = 64 * kPyrWaves;
__global__ __launch_bounds__(kPyrThreads) void orb_pyramid_kernel(uint8_t* __restrict__ pool, const ResizeJob* __restrict__ jobs,
                                                          const PyrTile* __restrict__ tiles, const PyrPlan plan) {
  // (every LDS access below indexes pyr_lds itself with an integer offset: pointers picked from an array of two buffer
  // pointers made the compiler fall back to flat loads and stores, each with its own wait)
  extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];
  const int buf_off[2] = {0, plan.buf_bytes[0]};   // level l lives at buf_off[(l + 1) & 1]
  const int xtab_off = plan.buf_bytes[0] + plan.buf_bytes[1], ytab_off = xtab_off + 8 * plan.max_rw;
  const PyrTile* __restrict__ tl = tiles + blockIdx.x;
  const int chain = tl->chain;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  // the tile's part of the uploaded image -> LDS, whole dwords (the region starts at a multiple of four columns of the
  // image and its LDS rows are a multiple of four bytes long; the last dword of a row may reach past the region: the pool
  // has slack behind its last image)
  int px0 = tl->nx0[0], py0 = tl->ny0[0], prw = (tl->nx1[0] - tl->nx0[0] + 3) & ~3;   // the level below, as it lies in LDS
  {
    const ResizeJob j = jobs[plan.level_job_begin[1] + chain];
    const uint8_t* __restrict__ src = pool + j.src_off + (size_t)py0 * j.sstride + px0;
    const int dwords = prw >> 2, rows = tl->ny1[0] - py0;
    for (int r = ty; r < rows; r += kPyrWaves)
      for (int d = tx; d < dwords; d += 64)
        *reinterpret_cast<uint32_t*>(&pyr_lds[buf_off[1] + 4 * (r * dwords + d)]) =
            *reinterpret_cast<const u32_unaligned*>(src + (size_t)r * j.sstride + 4 * d);
  }
  for (int l = 1; l < 8; ++l) {
    const int x0 = tl->nx0[l], x1 = tl->nx1[l], y0 = tl->ny0[l], y1 = tl->ny1[l];
    const int rw = x1 - x0, rh = y1 - y0;
    if (rw <= 0 || rh <= 0) break;   // (block-uniform; a tile without pixels at level l has none below it either)
    const ResizeJob j = jobs[plan.level_job_begin[l] + chain];
    // taps and weights of the region's columns and rows; source coordinates relative to the level below as it lies in LDS
    for (int i = tid; i < rw + rh; i += kPyrThreads) {
      if (i < rw) {
        const ResizeTapX t = resize_tap_x(x0 + i, j.scale_x, j.sw);
        *reinterpret_cast<ushort4*>(&pyr_lds[xtab_off + 8 * i]) =
            make_ushort4((unsigned short)(t.s0 - px0), (unsigned short)(t.s1 - px0), (unsigned short)t.w0, (unsigned short)t.w1);
      } else {
        const ResizeTapY t = resize_tap_y(y0 + (i - rw), j.scale_y, j.sh);
        *reinterpret_cast<ushort4*>(&pyr_lds[ytab_off + 8 * (i - rw)]) =
            make_ushort4((unsigned short)((t.r0 - py0) * prw), (unsigned short)((t.r1 - py0) * prw), (unsigned short)t.b0,
                         (unsigned short)t.b1);
      }
    }
    __syncthreads();   // the level below and the tables are complete
    const int cur = buf_off[(l + 1) & 1], prev = buf_off[l & 1];
    const int ox0 = tl->ox0[l], ox1 = tl->ox1[l], oy0 = tl->oy0[l], oy1 = tl->oy1[l];
    uint8_t* __restrict__ dst = pool + j.dst_off;
    const bool is_mask = j.is_mask != 0;
    for (int x = tx; x < rw; x += 64) {
      const ushort4 cx = *reinterpret_cast<const ushort4*>(&pyr_lds[xtab_off + 8 * x]);
      const int w0 = (short)cx.z, w1 = (short)cx.w;
      const bool own_x = x0 + x >= ox0 && x0 + x < ox1;
      const int c0 = prev + cx.x, c1 = prev + cx.y;
      // four rows per step: all their taps are read before the first result is formed
      for (int yb = ty; yb < rh; yb += 4 * kPyrWaves) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int y = min(yb + kPyrWaves * u, rh - 1);   // (rows past the region repeat its last row and are not stored)
          const ushort4 cy = *reinterpret_cast<const ushort4*>(&pyr_lds[ytab_off + 8 * y]);
          const int h0 = pyr_lds[c0 + cy.x] * w0 + pyr_lds[c1 + cy.x] * w1;
          const int h1 = pyr_lds[c0 + cy.y] * w0 + pyr_lds[c1 + cy.y] * w1;
          int t = ((((int)(short)cy.z * (h0 >> 4)) >> 16) + (((int)(short)cy.w * (h1 >> 4)) >> 16) + 2) >> 2;
          t = min(max(t, 0), 255);
          if (is_mask && t <= 254) t = 0;
          v[u] = t;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int y = yb + kPyrWaves * u;
          if (y < rh) {
            pyr_lds[cur + y * rw + x] = (uint8_t)v[u];
            if (own_x && y0 + y >= oy0 && y0 + y < oy1) dst[(size_t)(y0 + y) * j.dw + (x0 + x)] = (uint8_t)v[u];
          }
        }
      }
    }
    px0 = x0; py0 = y0; prw = rw;
    __syncthreads();   // the level is complete in LDS; the tables may be overwritten
  }
}

// ------------------------------------------------------------------------------------------------
// cv::FAST TYPE_9_16 corner test + cornerScore<16> + 3x3 non-maximum suppression for every pixel of every image of the
// launch, one kernel.
//
// cv::FAST keeps a pixel when 9 contiguous ring pixels are all brighter than v + t or all darker than v - t, and scores it
// with cornerScore<16> = (the largest t' for which that still holds) = max over the 16 arcs of 9 of min |difference| over
// the arc, taken over both polarities, minus 1 (fast_score.cpp: a0 starts at the threshold and only grows, b0 starts at
// -a0).  So with  P = max_arcs min_{k in arc} d[k]  and  N = max_arcs min_{k in arc} -d[k]  (d = centre - ring):
//   corner  <=>  max(P, N) > t,      score = max(P, N) - 1.
// An arc minimum is min3 over three min3's (9 = 3 x 3, v_min3_i32), the maximum over the 16 arcs eight max3's: ~100 integer
// operations per pixel, no branches, the same integers as the reference's loops.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// ptr -> the centre pixel inside an LDS tile of row stride `stride`
__device__ __forceinline__ int fast_arc_score(const uint8_t* __restrict__ ptr, int stride) {
  const int v = ptr[0];
  int d[16];
  d[0] = v - ptr[3 * stride];       d[1] = v - ptr[1 + 3 * stride];   d[2] = v - ptr[2 + 2 * stride];
  d[3] = v - ptr[3 + stride];       d[4] = v - ptr[3];                d[5] = v - ptr[3 - stride];
  d[6] = v - ptr[2 - 2 * stride];   d[7] = v - ptr[1 - 3 * stride];   d[8] = v - ptr[-3 * stride];
  d[9] = v - ptr[-1 - 3 * stride];  d[10] = v - ptr[-2 - 2 * stride]; d[11] = v - ptr[-3 - stride];
  d[12] = v - ptr[-3];              d[13] = v - ptr[-3 + stride];     d[14] = v - ptr[-2 + 2 * stride];
  d[15] = v - ptr[-1 + 3 * stride];
  int lo3[16], hi3[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    hi3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int P = -256, N = 256;
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const int p0 = min3i(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    const int p1 = min3i(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]);
    P = max3i(P, p0, p1);
    const int n0 = max3i(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
    const int n1 = max3i(hi3[k + 1], hi3[(k + 4) & 15], hi3[(k + 7) & 15]);
    N = min3i(N, n0, n1);
  }
  return max(P, -N);
}

// A workgroup owns 64 x 16 pixels of one (cell, level) image: the source tile with a halo of 4 goes through LDS (unaligned
// dword loads), the scores of the 66 x 18 pixels around the tile are computed into LDS, and the 3x3 test + mask + border
// filters run from there -- no score plane leaves the workgroup, only the survivors' scores (sparse byte stores into the plane
// buffer, read by the emit stage).  Survivors leave as one bit per pixel (64 pixels per word) plus per-row counts (atomics:
// an image row can span several tiles; the row scan that consumes the counts zeroes them again).
constexpr int kFastTW = 64, kFastTH = 16, kFastSrcStride = 76, kFastScStride = 68;
#ifndef RGBDFE_FAST_PRESCREEN
#define RGBDFE_FAST_PRESCREEN 1   // 0: the arc score for every pixel (rounds 1-3)
#endif
__global__ __launch_bounds__(256) void orb_fast_nms_kernel(const uint8_t* __restrict__ pool, const ImgDesc* __restrict__ imgs,
                                                           const OrbCtl ctl, uint8_t* __restrict__ score_pool, int edge,
                                                           int* __restrict__ row_cnt, uint64_t* __restrict__ keep_mask,
                                                           int* __restrict__ grand_total,
                                                           const TileUnit* __restrict__ units) {
  __shared__ __attribute__((aligned(4))) uint8_t src[(kFastTH + 8) * kFastSrcStride];
  __shared__ uint8_t sc[(kFastTH + 2) * kFastScStride];
  __shared__ uint16_t cand[(kFastTH + 2) * 66];   // the tile's pixels that pass the four-pixel screen
  __shared__ int n_cand;
  if (blockIdx.x == 0 && threadIdx.x == 0) *grand_total = 0;   // the row scan (next launch) adds the per-image totals
  const TileUnit u = units[blockIdx.x];
  const ImgDesc im = imgs[u.img];
  if (!ctl.active[im.cell]) return;
  const int tid = threadIdx.x;
  const int x0 = u.bx * kFastTW, y0 = u.by * kFastTH;
  const uint8_t* __restrict__ img = pool + im.off;
  // source rows y0 - 4 .. y0 + 19, columns x0 - 4 .. x0 + 67 (18 dwords per row).  Rows are clamped into the image and a
  // dword left of column 0 is not loaded: such bytes only feed pixels whose score is 0 by definition (3-pixel border);
  // a dword may run over the right end of a row (into the next row / the pool's slack) for the same reason.
  for (int i = tid; i < (kFastTH + 8) * 18; i += 256) {
    const int r = (i * 3641) >> 16, j = i - r * 18;   // i / 18 for i < 432
    const int gy = min(max(y0 - 4 + r, 0), im.h - 1), gx = x0 - 4 + 4 * j;
    uint32_t v = 0;
    if (gx >= 0 && gx < im.w) v = *reinterpret_cast<const u32_unaligned*>(img + (size_t)gy * im.stride + gx);
    *reinterpret_cast<uint32_t*>(src + r * kFastSrcStride + 4 * j) = v;
  }
  if (tid == 0) n_cand = 0;
  __syncthreads();
  int thr = ctl.thr[im.cell];
  thr = min(max(thr, 0), 255);
  uint8_t* __restrict__ score_img = score_pool + im.score_off;
#if RGBDFE_FAST_PRESCREEN
  // Round 4: the ~100-operation arc score only for the pixels that can be corners at all.  An arc of 9 of the 16 ring pixels
  // contains one pixel of every antipodal pair, so "all nine brighter than v + t" needs max(d[k], d[k + 8]) > t for every k
  // (d = centre - ring), "all nine darker" min(d[k], d[k + 8]) < -t: testing the pairs (0, 8) and (4, 12) -- four ring pixels --
  // is a NECESSARY condition for a score above the threshold, exact whatever it lets through.  Pixels that pass are queued
  // in LDS (ballot + one atomic per wave) and scored densely afterwards; everything else scores 0 as before.
  for (int i0 = 0; i0 < (kFastTH + 2) * 66; i0 += 256) {
    const int i = i0 + tid;
    bool pass = false;
    if (i < (kFastTH + 2) * 66) {
      const int ty = (i * 993) >> 16, tx = i - ty * 66;   // i / 66 for i < 1188
      const int x = x0 - 1 + tx, y = y0 - 1 + ty;
      sc[ty * kFastScStride + tx] = 0;
      if (x >= 3 && x < im.w - 3 && y >= 3 && y < im.h - 3) {
        const uint8_t* ptr = src + (ty + 3) * kFastSrcStride + (tx + 3);
        const int v = ptr[0];
        const int d0 = v - ptr[3 * kFastSrcStride], d8 = v - ptr[-3 * kFastSrcStride], d4 = v - ptr[3], d12 = v - ptr[-3];
        pass = (max(d0, d8) > thr && max(d4, d12) > thr) || (min(d0, d8) < -thr && min(d4, d12) < -thr);
      }
    }
    const uint64_t m = __ballot(pass);
    if (m) {
      const int lane = tid & 63;
      int base = 0;
      if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&n_cand, (int)__popcll(m));
      base = __shfl(base, (int)__builtin_ctzll(m));
      if (pass) cand[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)i;
    }
  }
  __syncthreads();
  for (int q = tid; q < n_cand; q += 256) {
    const int i = cand[q];
    const int ty = (i * 993) >> 16, tx = i - ty * 66;
    const int mm = fast_arc_score(src + (ty + 3) * kFastSrcStride + (tx + 3), kFastSrcStride);
    if (mm > thr) sc[ty * kFastScStride + tx] = (uint8_t)(mm - 1);
  }
  __syncthreads();
#else
  for (int i = tid; i < (kFastTH + 2) * 66; i += 256) {
    const int ty = (i * 993) >> 16, tx = i - ty * 66;   // i / 66 for i < 1188
    const int x = x0 - 1 + tx, y = y0 - 1 + ty;
    int s = 0;
    if (x >= 3 && x < im.w - 3 && y >= 3 && y < im.h - 3) {
      const int m = fast_arc_score(src + (ty + 3) * kFastSrcStride + (tx + 3), kFastSrcStride);
      s = m > thr ? m - 1 : 0;
    }
    sc[ty * kFastScStride + tx] = (uint8_t)s;
  }
  __syncthreads();
#endif
  // wave w: rows 4w .. 4w + 3 of the tile, lane = column
  const int lane = tid & 63, w = tid >> 6;
  const int words = (im.w + 63) >> 6;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int ty = w * 4 + rr, y = y0 + ty, x = x0 + lane;
    if (y >= im.h) break;
    const uint8_t* p = sc + (ty + 1) * kFastScStride + (lane + 1);
    const int s = p[0];
    bool keep = false;
    if (s && x < im.w) {
      keep = s > p[1] && s > p[-1] && s > p[-kFastScStride - 1] && s > p[-kFastScStride] && s > p[-kFastScStride + 1] &&
             s > p[kFastScStride - 1] && s > p[kFastScStride] && s > p[kFastScStride + 1];
      if (keep && im.has_mask && pool[im.mask_off + (size_t)y * im.mask_stride + x] == 0) keep = false;   // runByPixelsMask
      keep = keep && x >= edge && x < im.w - edge && y >= edge && y < im.h - edge;                          // runByImageBorder
    }
    // only the survivors' scores leave the workgroup (the emit stage reads nothing else of the score plane): a few thousand
    // bytes per frame instead of one byte per pixel of every (cell, level) image (round 4, VERDICT r3 #5b)
    if (keep) score_img[(size_t)y * im.w + x] = (uint8_t)s;
    const uint64_t m = __ballot(keep);
    if (lane == 0) {
      keep_mask[im.keep_off + (size_t)y * words + u.bx] = m;
      if (m) atomicAdd(&row_cnt[im.row_off + y], __popcll(m));
    }
  }
}

// one block per image: exclusive scan of the row counts (-> row_off), total per image; the counts are zeroed for the next
// pass (orb_fast_nms_kernel accumulates them with atomics)
__global__ __launch_bounds__(256) void orb_row_scan_kernel(const ImgDesc* __restrict__ imgs, const OrbCtl ctl,
                                                           int* __restrict__ row_cnt, int* __restrict__ row_off,
                                                           int* __restrict__ img_total, int* __restrict__ grand_total) {
  __shared__ int wave_total[4];
  const ImgDesc im = imgs[blockIdx.x];
  if (!ctl.active[im.cell]) { if (threadIdx.x == 0) img_total[blockIdx.x] = 0; return; }
  const int per = (im.h + 255) / 256;
  const int r0 = threadIdx.x * per;
  int sum = 0;
  for (int r = r0; r < min(r0 + per, im.h); ++r) sum += row_cnt[im.row_off + r];
  // exclusive prefix of the 256 partial sums: shuffles inside a wave, four wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  int acc = incl - sum;
  int total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = wave_total[w];
    if (w < wave) acc += t;
    total += t;
  }
  if (threadIdx.x == 0) {
    img_total[blockIdx.x] = total;
    if (total) atomicAdd(grand_total, total);   // (zeroed by the FAST launch) the measure waves read one word
  }
  for (int r = r0; r < min(r0 + per, im.h); ++r) {
    const int t = row_cnt[im.row_off + r];
    row_off[im.row_off + r] = acc;
    row_cnt[im.row_off + r] = 0;
    acc += t;
  }
}

// one wave per 64 rows of an image, LANE = ROW: the lane writes the keypoints of its row at img_base + row_offset + rank
// (raster order).  The row scan's offsets say which rows have keypoints at all (most have none) and where they go; a lane
// with keypoints loads its row's keep words (four at a time, all lanes' loads in flight together) and walks their set bits.
// (Rounds 1-4: one wave per row, ~55 000 waves per super-frame, most of them ending after the chain rows[] -> imgs[] -> mask
// words of an empty row.  A wave per 64 rows that walked its non-empty rows one after the other took as long: the chain of
// dependent loads per row had only moved inside the wave.)
__device__ __forceinline__ int wave_sum(int v);
__global__ __launch_bounds__(64) void orb_emit_kernel(const ImgDesc* __restrict__ imgs, const OrbCtl ctl,
                                                      const uint8_t* __restrict__ score_pool,
                                                      const uint64_t* __restrict__ keep_mask,
                                                      const int* __restrict__ row_off, const int* __restrict__ img_total,
                                                      RawKp* __restrict__ out, const TileUnit* __restrict__ rows) {
  const int lane_id = threadIdx.x;
  const TileUnit u = rows[blockIdx.x];
  const int img = u.img;
  const ImgDesc im = imgs[img];
  if (!ctl.active[im.cell]) return;
  const int y = u.by + lane_id;
  // keypoints of the lane's row = the next row's offset (the image's total behind the last row) minus its own
  int off_mine = 0, cnt_mine = 0;
  if (y < im.h) {
    off_mine = row_off[im.row_off + y];
    cnt_mine = (y + 1 < im.h ? row_off[im.row_off + y + 1] : img_total[img]) - off_mine;
  }
  if (__ballot(cnt_mine > 0) == 0ull) return;
  const int words = (im.w + 63) >> 6;
  // where this image's keypoints start = the keypoints of the images before it (at most 512 counts: a wave sums them,
  // which is cheaper than a scan launch in front of this kernel)
  int img_base = 0;
  for (int j0 = 0; j0 < img; j0 += 64) {
    const int j = j0 + lane_id;
    img_base += j < img ? img_total[j] : 0;
  }
  img_base = wave_sum(img_base);
  const bool mine = cnt_mine > 0;
  const uint64_t* __restrict__ km = keep_mask + im.keep_off + (size_t)(mine ? y : 0) * words;
  const uint8_t* __restrict__ sc = score_pool + im.score_off + (size_t)(mine ? y : 0) * im.w;
  RawKp* __restrict__ dst = out + img_base + off_mine;
  for (int wd0 = 0; wd0 < words; wd0 += 4) {
    uint64_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = (mine && wd0 + q < words) ? km[wd0 + q] : 0ull;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint64_t mm = m[q];
      while (mm != 0ull) {
        const int x = (wd0 + q) * 64 + (int)__builtin_ctzll(mm);
        mm &= mm - 1ull;
        RawKp k;
        k.x = (uint16_t)x; k.y = (uint16_t)y; k.img = (uint16_t)img; k.score = (uint16_t)sc[x];
        k.harris = 0.f; k.angle = 0.f;
        *dst++ = k;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// one wave per keypoint: HarrisResponses (7x7 block of Sobel products, k = 0.04) and ICAngles
// (intensity centroid over the radius-15 disc, cv::fastAtan2) -- orb.cpp
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// the sum over the wave as a wave-uniform value, on the DPP path (row shifts by 1, 2, 4, 8: lane 15 of every row of 16 holds
// its row; row_bcast 15 / 31: lane 63 holds everything): six VALU instructions and one readlane instead of six
// ds_bpermute round trips
__device__ __forceinline__ int wave_sum_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// first..first+grid keypoints.  The 31 x 31 patch around the keypoint (it lies >= 31 pixels inside the image: runByImageBorder)
// goes through LDS -- four unaligned dword loads per lane instead of 23 dependent byte gathers -- and both measurements read it
// from there.
// host_totals / host_kps (may be nullptr): the pass's read-back buffer in page-locked HOST memory -- [per-image counts + their
// sum | keypoints] -- written by this kernel itself: every wave stores its finished 16-byte record there as well, the first
// workgroup copies the counts.  Posted writes over PCIe that ride along with the kernel; the detection pass needs no copy
// (a 50-100 us blit kernel on the pass's stream at 14 frames of 640x480) behind it.
__global__ __launch_bounds__(256) void orb_measure_kernel(const uint8_t* __restrict__ pool,
                                                          const ImgDesc* __restrict__ imgs,
                                                          RawKp* __restrict__ kps, const int* __restrict__ img_total,
                                                          int n_imgs, int first, int* __restrict__ host_totals,
                                                          RawKp* __restrict__ host_kps) {
  __shared__ __attribute__((aligned(4))) uint8_t patch_all[4][31 * 32];
  const int k = first + blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (host_totals != nullptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i <= n_imgs; i += 256) host_totals[i] = img_total[i];
  // how many keypoints there are (the host has not seen the counts yet): the scan left the sum behind the per-image counts
  const int n_total = img_total[n_imgs];
  if (k >= n_total) return;
  RawKp kp = kps[k];
  const ImgDesc im = imgs[kp.img];
  const int stride = im.stride;
  uint8_t* __restrict__ patch = patch_all[threadIdx.x >> 6];
  const uint8_t* __restrict__ corner = pool + im.off + (size_t)(kp.y - 15) * stride + (kp.x - 15);
#pragma unroll
  for (int i0 = 0; i0 < 256; i0 += 64) {
    const int i = i0 + lane;
    if (i < 31 * 8) {
      const int r = i >> 3, j = i & 7;
      *reinterpret_cast<uint32_t*>(patch + r * 32 + 4 * j) = *reinterpret_cast<const u32_unaligned*>(corner + (size_t)r * stride + 4 * j);
    }
  }
  __builtin_amdgcn_wave_barrier();   // one wave: its LDS writes precede its LDS reads in program order
  const uint8_t* __restrict__ center = patch + 15 * 32 + 15;
  // Harris: 49 positions, one per lane
  int a = 0, b = 0, c = 0;
  if (lane < 49) {
    const uint8_t* ptr = center + (lane / 7 - 3) * 32 + (lane % 7 - 3);
    const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-32 + 1] - ptr[-32 - 1]) + (ptr[32 + 1] - ptr[32 - 1]);
    const int Iy = (ptr[32] - ptr[-32]) * 2 + (ptr[32 - 1] - ptr[-32 - 1]) + (ptr[32 + 1] - ptr[-32 + 1]);
    a = Ix * Ix; b = Iy * Iy; c = Ix * Iy;
  }
  a = wave_sum_dpp(a); b = wave_sum_dpp(b); c = wave_sum_dpp(c);
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  const float harris = ((float)a * b - (float)c * c - 0.04f * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
  // IC angle: integer moments are order-independent -> any lane assignment is exact.  Two patch rows per step: lanes 0-31
  // row 2i, lanes 32-63 row 2i + 1.
  int m01 = 0, m10 = 0;
  const int u = (lane & 31) - 15, half = lane >> 5;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int v = 2 * it + half - 15;
    const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
    if (v <= 15 && au <= c_umax[av & 15]) {
      const int val = center[u + v * 32];
      m10 += u * val;
      m01 += v * val;
    }
  }
  m01 = wave_sum_dpp(m01); m10 = wave_sum_dpp(m10);
  if (lane == 0) {
    kp.harris = harris;
    kp.angle = fast_atan2_deg((float)m01, (float)m10);
    kps[k] = kp;
    if (host_kps != nullptr) host_kps[k] = kp;
  }
}

// ------------------------------------------------------------------------------------------------
// GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101, 8-bit fixed point (kernel x256, (v + 2^15) >> 16)
// ------------------------------------------------------------------------------------------------
__constant__ int c_gauss[7] = {18, 34, 49, 55, 49, 34, 18};

// Separable through LDS: the reference sums c[j] * (sum_i c[i] * p[y+j][x+i]) -- the inner sums h are shared by the 7
// output rows that use them (integer arithmetic: the same numbers whatever the order of evaluation).  A block owns 64 x 16
// output pixels: 22 x 72 source bytes (reflect-101 coordinates; dword loads wherever four bytes lie inside the row) ->
// 22 x 64 row sums (<= 255 * 257: 16 bits; a thread forms four neighbouring sums from three LDS dwords) -> 16 x 64 outputs
// (a thread forms four rows of a column from ten row sums).
constexpr int kBlurTH = 16, kBlurStride = 76;
__global__ __launch_bounds__(256) void orb_blur_kernel(const uint8_t* __restrict__ pool, const ImgDesc* __restrict__ imgs,
                                                       uint8_t* __restrict__ blur_pool, const TileUnit* __restrict__ units) {
  __shared__ __attribute__((aligned(8))) uint8_t patch[(kBlurTH + 6) * kBlurStride];   // columns x0 - 4 .. x0 + 67
  __shared__ __attribute__((aligned(8))) uint16_t hsum[(kBlurTH + 6) * 64];
  const TileUnit u = units[blockIdx.x];
  const ImgDesc im = imgs[u.img];
  const int x0 = u.bx * 64, y0 = u.by * kBlurTH;
  const uint8_t* __restrict__ src = pool + im.off;
  const int tid = threadIdx.x;
  for (int i = tid; i < (kBlurTH + 6) * 18; i += 256) {
    const int r = (i * 3641) >> 16, j = i - r * 18;   // i / 18 for i < 396
    const int gy = reflect101(min(y0 + r - 3, im.h + 2), im.h);
    const uint8_t* __restrict__ row = src + (size_t)gy * im.stride;
    const int gx = x0 - 4 + 4 * j;
    uint32_t v;
    if (gx >= 0 && gx + 3 < im.w) {
      v = *reinterpret_cast<const u32_unaligned*>(row + gx);
    } else {
      v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) v |= (uint32_t)row[reflect101(min(max(gx + k, -3), im.w + 2), im.w)] << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(patch + r * kBlurStride + 4 * j) = v;
  }
  __syncthreads();
  for (int i = tid; i < (kBlurTH + 6) * 16; i += 256) {
    const int r = i >> 4, g = i & 15;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(patch + r * kBlurStride + 4 * g);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    int b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 255; b[4 + k] = (w1 >> (8 * k)) & 255; b[8 + k] = (w2 >> (8 * k)) & 255; }
    uint32_t h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      int acc = 0;
#pragma unroll
      for (int k = 0; k < 7; ++k) acc += c_gauss[k] * b[t + k + 1];   // output column 4g + t reads patch columns 4g + t + 1 ..
      h[t] = (uint32_t)acc;
    }
    uint2 o;
    o.x = h[0] | (h[1] << 16);
    o.y = h[2] | (h[3] << 16);
    *reinterpret_cast<uint2*>(hsum + r * 64 + 4 * g) = o;
  }
  __syncthreads();
  {
    const int tx = tid & 63, ty0 = (tid >> 6) * 4;
    const int x = x0 + tx;
    int hv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) hv[j] = hsum[(ty0 + j) * 64 + tx];
    if (x < im.w) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int y = y0 + ty0 + t;
        if (y >= im.h) break;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) sum += c_gauss[j] * hv[t + j];
        int v = (sum + (1 << 15)) >> 16;
        v = min(max(v, 0), 255);
        blur_pool[im.score_off + (size_t)y * im.w + x] = (uint8_t)v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// rBRIEF (computeOrbDescriptors, WTA_K = 2): one wave per keypoint, lane = (byte, half): every lane
// evaluates 4 of the 256 tests; bits are assembled with shuffles.
// ------------------------------------------------------------------------------------------------
__constant__ int8_t c_pattern[1024];

void orb_upload_pattern(const int8_t* host_pattern) {   // (on the setup stream: see on_setup_stream, orb_host.hip)
  (void)orb_setup_stream_run([&](hipStream_t s) {
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pattern), host_pattern, 1024, 0, hipMemcpyHostToDevice, s);
  });
}

__global__ __launch_bounds__(256) void orb_brief_kernel(const uint8_t* __restrict__ pool, const uint8_t* __restrict__ blur_pool,
                                                        const ImgDesc* __restrict__ imgs, const DescKp* __restrict__ kps,
                                                        int n, uint8_t* __restrict__ desc) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const int lane = threadIdx.x & 63;
  const DescKp kp = kps[k];
  const ImgDesc im = imgs[kp.level];
  const uint8_t* __restrict__ raw = pool + im.off;
  const uint8_t* __restrict__ blur = blur_pool + im.score_off;
  const float a = kp.cos_a, b = kp.sin_a;
  const int byte = lane >> 1, half = lane & 1;
  int bits = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int test = byte * 8 + half * 4 + t;
    int v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float px = (float)c_pattern[(test * 2 + e) * 2], py = (float)c_pattern[(test * 2 + e) * 2 + 1];
      const float x = px * a - py * b;
      const float y = px * b + py * a;
      const int ix = kp.cx + __float2int_rn(x), iy = kp.cy + __float2int_rn(y);
      if (ix >= 0 && ix < im.w && iy >= 0 && iy < im.h)
        v[e] = blur[(size_t)iy * im.w + ix];
      else  // the unblurred reflect-101 border copyMakeBorder wrote before the in-place blur
        v[e] = raw[(size_t)reflect101(iy, im.h) * im.stride + reflect101(ix, im.w)];
    }
    bits |= (v[0] < v[1]) << (half * 4 + t);
  }
  bits |= __shfl_xor(bits, 1);
  if (half == 0) desc[(size_t)k * 32 + byte] = (uint8_t)bits;
}

// ------------------------------------------------------------------------------------------------
// rBRIEF with the Gaussian computed where it is read (round 5; RGBDFE_ORB_BRIEF=patch, NOT the default): a keypoint's 256
// tests read 512 pixels of the BLURRED level within 18 pixels of it, and blurring the whole pyramid (orb_blur_kernel: a read
// and a write of 3.1 x W x H bytes per frame) to read ~1000 x 512 of its pixels is a second full pass over the pyramid.
// Here a wave copies the 45 x 45 raw pixels around its keypoint (the samples' reach + the 7 x 7 kernel's halo, reflect-101
// coordinates) into LDS and every lane forms the blurred value of its 8 samples from there:
// (sum_j c[j] * sum_i c[i] * p[y + j][x + i] + 2^15) >> 16, the integer the separable pass computes (the row sums fit 16
// bits, nothing is rounded in between).  Samples outside the level read the raw reflected pixel, as before.
// Same bytes (tests/test_gpu_orb.py passes in both modes) -- and measured SLOWER at the bench's density: 49 byte reads and
// multiply-adds per sample are 13.1 us per 640x480 frame of ~920 keypoints against 7.1 (blur) + 1.7 (brief) through the pool
// (profiles/r05_logs/brief_patch_ab.log); it pays below ~500 keypoints per frame, which no caller of the batch pipeline has.
// Kept as the switch for sparse callers and as the record of the attempt (VERDICT r4 #4a asked for the blur pass to go).
// ------------------------------------------------------------------------------------------------
constexpr int kBriefReach = 19;                       // |rotated pattern point| <= 18.39 (the table's largest radius), rounded
constexpr int kBriefR = kBriefReach + 3, kBriefW = 2 * kBriefR + 1, kBriefStride = 48;
static_assert(kBriefStride >= kBriefW && kBriefStride % 4 == 0, "rows of whole dwords");
__global__ __launch_bounds__(256) void orb_brief_patch_kernel(const uint8_t* __restrict__ pool, const ImgDesc* __restrict__ imgs,
                                                              const DescKp* __restrict__ kps, int n, uint8_t* __restrict__ desc) {
  __shared__ __attribute__((aligned(4))) uint8_t region_all[4][kBriefW * kBriefStride];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + wave;
  if (k >= n) return;   // (no workgroup barrier below: a wave's region is its own)
  uint8_t* __restrict__ region = region_all[wave];
  const DescKp kp = kps[k];
  const ImgDesc im = imgs[kp.level];
  const uint8_t* __restrict__ raw = pool + im.off;
  const int gx0 = kp.cx - kBriefR, gy0 = kp.cy - kBriefR;
  const bool rows_inside = gx0 >= 0 && gx0 + kBriefStride - 1 < im.w;   // whole dwords of every row lie inside the level
  for (int i = lane; i < kBriefW * (kBriefStride / 4); i += 64) {
    const int r = i / (kBriefStride / 4), j = i - r * (kBriefStride / 4);
    const uint8_t* __restrict__ row = raw + (size_t)reflect101(gy0 + r, im.h) * im.stride;
    uint32_t v;
    if (rows_inside) {
      v = *reinterpret_cast<const u32_unaligned*>(row + gx0 + 4 * j);
    } else {
      v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) v |= (uint32_t)row[reflect101(gx0 + 4 * j + b, im.w)] << (8 * b);
    }
    *reinterpret_cast<uint32_t*>(region + r * kBriefStride + 4 * j) = v;
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const float a = kp.cos_a, b = kp.sin_a;
  const int byte = lane >> 1, half = lane & 1;
  int bits = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int test = byte * 8 + half * 4 + t;
    int v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float px = (float)c_pattern[(test * 2 + e) * 2], py = (float)c_pattern[(test * 2 + e) * 2 + 1];
      const float x = px * a - py * b;
      const float y = px * b + py * a;
      const int dx = __float2int_rn(x), dy = __float2int_rn(y);
      const int ix = kp.cx + dx, iy = kp.cy + dy;
      const uint8_t* __restrict__ p = region + (dy + kBriefR) * kBriefStride + (dx + kBriefR);
      if (ix >= 0 && ix < im.w && iy >= 0 && iy < im.h) {
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          int h = 0;
#pragma unroll
          for (int i = 0; i < 7; ++i) h += c_gauss[i] * (int)p[(j - 3) * kBriefStride + (i - 3)];
          acc += c_gauss[j] * h;
        }
        v[e] = min(max((acc + (1 << 15)) >> 16, 0), 255);
      } else {  // the unblurred reflect-101 border copyMakeBorder wrote before the in-place blur
        v[e] = p[0];
      }
    }
    bits |= (v[0] < v[1]) << (half * 4 + t);
  }
  bits |= __shfl_xor(bits, 1);
  if (half == 0) desc[(size_t)k * 32 + byte] = (uint8_t)bits;
}

static bool orb_brief_from_pool() {
  static const bool patch = getenv("RGBDFE_ORB_BRIEF") && strcmp(getenv("RGBDFE_ORB_BRIEF"), "patch") == 0;
  return !patch;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_orb_resize(uint8_t* pool, const ResizeJob* jobs, const TileUnit* units, int n_units, hipStream_t s) {
  if (n_units == 0) return;
  hipLaunchKernelGGL(orb_resize_kernel, dim3(n_units), dim3(256), 0, s, pool, jobs, units);
}
void launch_orb_pyramid(uint8_t* pool, const ResizeJob* jobs, const PyrTile* tiles, int n_tiles, const PyrPlan& plan,
                        hipStream_t s) {
  if (n_tiles == 0) return;
  const size_t lds = (size_t)plan.buf_bytes[0] + plan.buf_bytes[1] + sizeof(ushort4) * (size_t)(plan.max_rw + plan.max_rh);
  hipLaunchKernelGGL(orb_pyramid_kernel, dim3(n_tiles), dim3(kPyrThreads), lds, s, pool, jobs, tiles, plan);
}
// FAST-9/16 scores + 3x3 NMS + mask + border filters (keep bits, per-row counts), then the per-image scan of the row counts
void launch_orb_fast_nms(const uint8_t* pool, const ImgDesc* imgs, int n_imgs, const TileUnit* units, int n_units,
                         const OrbCtl& ctl, uint8_t* score_pool, int edge, int* row_cnt, int* row_off, int* img_total,
                         uint64_t* keep_mask, hipStream_t s) {
  hipLaunchKernelGGL(orb_fast_nms_kernel, dim3(n_units), dim3(256), 0, s, pool, imgs, ctl, score_pool, edge, row_cnt,
                     keep_mask, img_total + n_imgs, units);
  hipLaunchKernelGGL(orb_row_scan_kernel, dim3(n_imgs), dim3(256), 0, s, imgs, ctl, row_cnt, row_off, img_total,
                     img_total + n_imgs);
}
// Keypoints of every active image in raster order, then Harris response + orientation for the first `measure_bound` of
// them -- all without the host knowing the count (it reads img_total back together with the keypoints; a frame with more
// keypoints than the bound gets the rest measured by launch_orb_measure_rest).
void launch_orb_emit(const uint8_t* pool, const ImgDesc* imgs, int n_imgs, const TileUnit* rows, int n_rows,
                     const OrbCtl& ctl, const uint8_t* score_pool, const uint64_t* keep_mask, const int* row_off,
                     const int* img_total, RawKp* out, int measure_bound, hipStream_t s, int* host_totals, RawKp* host_kps) {
  hipLaunchKernelGGL(orb_emit_kernel, dim3(n_rows), dim3(64), 0, s, imgs, ctl, score_pool, keep_mask, row_off, img_total,
                     out, rows);
  if (measure_bound > 0)
    hipLaunchKernelGGL(orb_measure_kernel, dim3((measure_bound + 3) / 4), dim3(256), 0, s, pool, imgs, out, img_total,
                       n_imgs, 0, host_totals, host_kps);
}
void launch_orb_measure_rest(const uint8_t* pool, const ImgDesc* imgs, RawKp* out, const int* img_total, int n_imgs,
                             int first, int count, hipStream_t s) {
  if (count > 0)
    hipLaunchKernelGGL(orb_measure_kernel, dim3((count + 3) / 4), dim3(256), 0, s, pool, imgs, out, img_total, n_imgs,
                       first, (int*)nullptr, (RawKp*)nullptr);
}
// (no blurred pyramid with RGBDFE_ORB_BRIEF=patch: that descriptor kernel blurs what it reads)
void launch_orb_blur(const uint8_t* pool, const ImgDesc* imgs, const TileUnit* units, int n_units, uint8_t* blur_pool,
                     hipStream_t s) {
  if (!orb_brief_from_pool()) return;
  hipLaunchKernelGGL(orb_blur_kernel, dim3(n_units), dim3(256), 0, s, pool, imgs, blur_pool, units);
}
// the blur kernel whatever the mode (tests/test_emu_orb_kernels.py, A/B runs)
void launch_orb_blur_always(const uint8_t* pool, const ImgDesc* imgs, const TileUnit* units, int n_units, uint8_t* blur_pool,
                            hipStream_t s) {
  hipLaunchKernelGGL(orb_blur_kernel, dim3(n_units), dim3(256), 0, s, pool, imgs, blur_pool, units);
}
void launch_orb_brief(const uint8_t* pool, const uint8_t* blur_pool, const ImgDesc* imgs, const DescKp* kps, int n,
                      uint8_t* desc, hipStream_t s) {
  if (n == 0) return;
  if (orb_brief_from_pool())
    hipLaunchKernelGGL(orb_brief_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pool, blur_pool, imgs, kps, n, desc);
  else
    hipLaunchKernelGGL(orb_brief_patch_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pool, imgs, kps, n, desc);
}

}  // namespace rgbdfe
