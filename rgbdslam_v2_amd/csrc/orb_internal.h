// orb_internal.h -- shared by orb_kernels.hip and orb_host.hip
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <functional>

#include "rgbdfe_internal.h"

namespace rgbdfe {

// one 8-bit image of a step (a pyramid level of a grid cell or of the whole frame) inside the byte pool
struct ImgDesc {
  uint32_t off;        // first pixel in the image pool
  int32_t w, h, stride;
  uint32_t score_off;  // FAST score map (cell images) / blurred copy (frame images), tightly packed w*h
  uint32_t mask_off;   // mask pixels in the image pool
  int32_t mask_stride;
  int32_t has_mask;
  int32_t cell;        // grid cell (index into thresholds / active flags)
  int32_t level;
  int32_t row_off;     // first entry of this image in the row-count array
  uint32_t keep_off;   // first 64-bit word of this image in the keep-mask array (ceil(w / 64) words per row)
};

struct ResizeJob {
  uint32_t src_off, dst_off;
  int32_t sw, sh, sstride, dw, dh, is_mask;
  double scale_x, scale_y;  // 1 / ((double)dw / sw), computed on the host exactly as cv::resize does
};

// a FAST keypoint as the device emits it (16 bytes)
struct RawKp {
  uint16_t x, y, img, score;
  float harris, angle;
};

// a keypoint handed to the descriptor kernel
struct DescKp {
  int32_t cx, cy, level;
  float cos_a, sin_a;
};

// per-cell FAST thresholds and active flags of a detection pass: travel as a KERNEL ARGUMENT (no upload, one dependent
// device operation fewer per pass)
constexpr int kOrbCtlMax = 256;   // (frame, cell) detectors of a launch: 28 frames of a 3 x 3 grid
struct OrbCtl {
  int32_t thr[kOrbCtlMax];
  int32_t active[kOrbCtlMax];
};

// one workgroup of a 2-D stage: tile (bx, by) of 64 x 4 (resize) or 64 x 16 (FAST, blur) pixels of image / resize job `img`;
// for the row stages (one wave per image row) by is the row
struct TileUnit {
  uint16_t img, bx, by, pad;
};

// One workgroup of the fused pyramid kernel: a tile of ONE chain of images (a cell's gray levels, a cell's mask levels or a
// frame's gray levels) followed through levels 1..7.  n*: the region of level l the workgroup COMPUTES (into LDS: what it owns
// plus what its regions of the deeper levels read), o*: the part of it the workgroup STORES (the tiles of a chain partition
// every level).  Index 0 is unused (level 0 is the uploaded image).
struct PyrTile {
  uint16_t chain, pad;
  uint16_t nx0[8], nx1[8], ny0[8], ny1[8];
  uint16_t ox0[8], ox1[8], oy0[8], oy1[8];
};
struct PyrPlan {
  int32_t level_job_begin[8];   // jobs[level_job_begin[l] + chain]: the resize of the chain's level l - 1 into level l
  int32_t buf_bytes[2];         // LDS: regions of the odd / even levels
  int32_t max_rw, max_rh;       // LDS: coefficient tables (8 bytes per column / row of a region)
};

void launch_orb_resize(uint8_t* pool, const ResizeJob* jobs, const TileUnit* units, int n_units, hipStream_t s);
void launch_orb_pyramid(uint8_t* pool, const ResizeJob* jobs, const PyrTile* tiles, int n_tiles, const PyrPlan& plan,
                        hipStream_t s);
// cv::resize's horizontal / vertical taps of destination column / row d (host and device: the fused kernel's regions are
// planned on the host with the arithmetic the kernels use)
struct ResizeTapX { int s0, s1, w0, w1; };
struct ResizeTapY { int r0, r1, b0, b1; };
__host__ __device__ inline int resize_coef(float f) {
  const int v = (int)rintf(f * 2048.f);
  return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
}
__host__ __device__ inline ResizeTapX resize_tap_x(int dx, double scale_x, int sw) {
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  const bool edge = (sx + 1 >= sw);   // at the right edge the second tap is not read: weights 2048 / 0 there
  if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
  ResizeTapX t;
  t.s0 = sx; t.s1 = edge ? sx : sx + 1;
  t.w0 = edge ? 2048 : resize_coef(1.f - fx);
  t.w1 = edge ? 0 : resize_coef(fx);
  return t;
}
__host__ __device__ inline ResizeTapY resize_tap_y(int dy, double scale_y, int sh) {
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  ResizeTapY t;
  t.b0 = resize_coef(1.f - fy);
  t.b1 = resize_coef(fy);
  t.r0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
  t.r1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
  return t;
}
void launch_orb_fast_nms(const uint8_t* pool, const ImgDesc* imgs, int n_imgs, const TileUnit* units, int n_units,
                         const OrbCtl& ctl, uint8_t* score_pool, int edge, int* row_cnt, int* row_off, int* img_total,
                         uint64_t* keep_mask, hipStream_t s);
// img_total[n_imgs] (the per-image counts) doubles as the source of every prefix the later kernels need: no scan launch
void launch_orb_emit(const uint8_t* pool, const ImgDesc* imgs, int n_imgs, const TileUnit* rows, int n_rows,
                     const OrbCtl& ctl, const uint8_t* score_pool, const uint64_t* keep_mask, const int* row_off,
                     const int* img_total, RawKp* out, int measure_bound, hipStream_t s, int* host_totals = nullptr,
                     RawKp* host_kps = nullptr);   // (page-locked host memory: the measure kernel writes the read-back itself)
void launch_orb_measure_rest(const uint8_t* pool, const ImgDesc* imgs, RawKp* out, const int* img_total, int n_imgs,
                             int first, int count, hipStream_t s);
void launch_orb_blur(const uint8_t* pool, const ImgDesc* imgs, const TileUnit* units, int n_units, uint8_t* blur_pool,
                     hipStream_t s);
void launch_orb_blur_always(const uint8_t* pool, const ImgDesc* imgs, const TileUnit* units, int n_units, uint8_t* blur_pool,
                            hipStream_t s);
void launch_orb_brief(const uint8_t* pool, const uint8_t* blur_pool, const ImgDesc* imgs, const DescKp* kps, int n,
                      uint8_t* desc, hipStream_t s);
void orb_upload_pattern(const int8_t* host_pattern);
hipError_t orb_setup_stream_run(const std::function<hipError_t(hipStream_t)>& op);   // orb_host.hip: op on the setup stream, then wait

}  // namespace rgbdfe
