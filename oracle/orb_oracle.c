/*
 * oracle/orb_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's per-frame feature path:
 *   createDetector / adjustedGridWrapper                      src/features.cpp:35-113
 *   DetectorAdjuster::detect, tooFew/tooMany/good             src/feature_adjuster.cpp:85-153
 *   VideoDynamicAdaptedFeatureDetector::detect                src/feature_adjuster.cpp:185-224
 *   VideoGridAdaptedFeatureDetector::detect, keepStrongest,
 *   aggregateKeypointsPerGridCell                             src/feature_adjuster.cpp:247-317
 *   Node::Node (detect -> removeDepthless -> retainBest ->
 *   compute -> projectTo3D)                                   src/node.cpp:139-210
 * and of the OpenCV 3.3 code those call (cv::ORB, cv::FAST, cv::resize, cv::GaussianBlur,
 * KeyPointsFilter).  OpenCV is NOT in the reference tree and not installed here:
 *
 *      >>>  PARITY UNPINNED  <<<
 *
 * everything below is restated from the published OpenCV 3.3 algorithm (modules/features2d/src/
 * orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp; modules/imgproc/src/resize.cpp, smooth.cpp,
 * filter.cpp) and anchored on the reference's call sites.  The HIP kernels are tested for exact
 * equality against THIS restatement; equality with a real OpenCV build could not be checked.
 * What IS pinned: the detector grid and its threshold adaptation (the first five entries above) -- the
 * reference's own feature_adjuster.cpp and features.cpp:35-60, compiled from where they lie into
 * oracle/_ref/libref_adjuster.so around this file's orb_detect, return the keypoints of orb_grid_detect
 * frame after frame (tests/test_oracle_orb.py).  Unpinned remains what happens inside cv::ORB itself.
 *
 * Deliberate deviation (same class as D2 in rgbd_oracle.c): wherever the reference's result
 * depends on std::nth_element's unspecified order (KeyPointsFilter::retainBest, keepStrongest),
 * ties are resolved by the original (raster) order and the surviving elements keep that order.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

/* cvRound: round half to even (lrint under the default rounding mode) */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (v < (float)i); }
static inline short sat_short_from_float(float v) {
  int iv = cv_round_f(v);
  return (short)(iv > 32767 ? 32767 : iv < -32768 ? -32768 : iv);
}
static inline int reflect101(int p, int len) {
  /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (imgproc/resize.cpp, fixed     */
/* point: INTER_RESIZE_COEF_BITS = 11)                                                         */
/* ------------------------------------------------------------------------------------------ */
void orb_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                          int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      if (dx < xmax) xmax = dx;
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_short_from_float((1.f - fx) * 2048);
    ialpha[dx * 2 + 1] = sat_short_from_float(fx * 2048);
  }
  int* rows[2];
  rows[0] = (int*)malloc(sizeof(int) * (size_t)dw);
  rows[1] = (int*)malloc(sizeof(int) * (size_t)dw);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    const short b0 = sat_short_from_float((1.f - fy) * 2048);
    const short b1 = sat_short_from_float(fy * 2048);
    for (int k = 0; k < 2; ++k) {
      int y = sy + k;
      y = y < 0 ? 0 : (y >= sh ? sh - 1 : y); /* clip(sy - ksize2 + 1 + k, 0, ssize.height) */
      const uint8_t* S = src + (size_t)y * sstride;
      int* D = rows[k];
      for (int dx = 0; dx < dw; ++dx) {
        const int sx = xofs[dx];
        if (dx < xmax)
          D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        else
          D[dx] = S[sx] * 2048;
      }
    }
    uint8_t* d = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; ++x) {
      /* VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>> */
      const int v = (((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2;
      d[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
  free(xofs); free(ialpha); free(rows[0]); free(rows[1]);
}

/* ------------------------------------------------------------------------------------------ */
/* ORB pyramid geometry (orb.cpp detectAndCompute): scale_l = (float)pow(scaleFactor, l),      */
/* size_l = cvRound(cols / scale_l); level l > 0 is resized from level l-1.                    */
/* ------------------------------------------------------------------------------------------ */
void orb_level_geometry(int cols, int rows, int nlevels, float* scale, int* lw, int* lh) {
  const double scaleFactor = (double)1.2f; /* ORB::create(..., float scaleFactor = 1.2f, ...) */
  for (int l = 0; l < nlevels; ++l) {
    scale[l] = (float)pow(scaleFactor, (double)l);
    lw[l] = cv_round_f((float)cols / scale[l]);
    lh[l] = cv_round_f((float)rows / scale[l]);
  }
}

/* Builds levels 1..nlevels-1 of image and mask pyramids.  level 0 is the input itself.
 * out_img[l] / out_mask[l] must hold lw[l]*lh[l] bytes (tightly packed) for l >= 1. */
void orb_build_pyramid(const uint8_t* img, const uint8_t* mask, int cols, int rows, int stride,
                       int mstride, int nlevels, const int* lw, const int* lh, uint8_t** out_img,
                       uint8_t** out_mask) {
  const uint8_t* prev = img;
  const uint8_t* prevm = mask;
  int pw = cols, ph = rows, ps = stride, pms = mstride;
  for (int l = 1; l < nlevels; ++l) {
    orb_resize_linear_u8(prev, pw, ph, ps, out_img[l], lw[l], lh[l], lw[l]);
    if (mask) {
      orb_resize_linear_u8(prevm, pw, ph, pms, out_mask[l], lw[l], lh[l], lw[l]);
      /* threshold(currMask, currMask, 254, 0, THRESH_TOZERO) */
      for (int i = 0; i < lw[l] * lh[l]; ++i)
        if (out_mask[l][i] <= 254) out_mask[l][i] = 0;
      prevm = out_mask[l];
      pms = lw[l];
    }
    prev = out_img[l];
    pw = lw[l]; ph = lh[l]; ps = lw[l];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* cv::FAST (TYPE_9_16) corner test + cornerScore<16> (features2d/src/fast.cpp, fast_score.cpp) */
/* ------------------------------------------------------------------------------------------ */
static const int kCircle[16][2] = {{0, 3},  {1, 3},  {2, 2},  {3, 1},  {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int orb_fast_score_at(const uint8_t* img, int stride, int x, int y, int threshold) {
  const uint8_t* ptr = img + (size_t)y * stride + x;
  const int v = ptr[0];
  int d[25];
  for (int k = 0; k < 25; ++k) {
    const int kk = k & 15;
    d[k] = v - ptr[kCircle[kk][0] + kCircle[kk][1] * stride];
  }
  /* corner test: >= 9 contiguous circle pixels all darker than v - t or all brighter than v + t
   * (fast.cpp FAST_t<16>: count > K with K = 8 over N = 25 wrapped samples) */
  int is_corner = 0;
  {
    int cnt_d = 0, cnt_b = 0;
    for (int k = 0; k < 25; ++k) {
      if (d[k] > threshold) { if (++cnt_d > 8) is_corner = 1; } else cnt_d = 0;  /* x < v - t */
      if (d[k] < -threshold) { if (++cnt_b > 8) is_corner = 1; } else cnt_b = 0; /* x > v + t */
    }
  }
  if (!is_corner) return 0;
  /* cornerScore<16> */
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    a = a < d[k + 3] ? a : d[k + 3];
    if (a <= a0) continue;
    for (int j = 4; j <= 8; ++j) a = a < d[k + j] ? a : d[k + j];
    int m = a < d[k] ? a : d[k];
    a0 = a0 > m ? a0 : m;
    m = a < d[k + 9] ? a : d[k + 9];
    a0 = a0 > m ? a0 : m;
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    for (int j = 3; j <= 5; ++j) b = b > d[k + j] ? b : d[k + j];
    if (b >= b0) continue;
    for (int j = 6; j <= 8; ++j) b = b > d[k + j] ? b : d[k + j];
    int m = b > d[k] ? b : d[k];
    b0 = b0 < m ? b0 : m;
    m = b > d[k + 9] ? b : d[k + 9];
    b0 = b0 < m ? b0 : m;
  }
  return -b0 - 1;
}

/* score map: 0 for non-corners and for the 3-pixel frame FAST never visits */
void orb_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score) {
  threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold; /* fast.cpp: min(max(t,0),255) */
  memset(score, 0, (size_t)w * h);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      int s = orb_fast_score_at(img, stride, x, y, threshold);
      score[(size_t)y * w + x] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
    }
}

/* 3x3 non-maximum suppression (strict >) in raster order; then the ORB level filters:
 * KeyPointsFilter::runByPixelsMask and runByImageBorder(edgeThreshold).  Returns the count. */
int orb_fast_keypoints(const uint8_t* score, const uint8_t* mask, int w, int h, int edge,
                       orb_keypoint* out, int cap) {
  int n = 0;
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x) {
      const int s = score[(size_t)y * w + x];
      if (!s) continue;
      const uint8_t* p = score + (size_t)y * w + x;
      if (!(s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] &&
            s > p[w] && s > p[w + 1]))
        continue;
      if (mask && mask[(size_t)(int)(y + 0.5f) * w + (int)(x + 0.5f)] == 0) continue;
      /* Rect(Point(b, b), Point(cols - b, rows - b)).contains(pt) */
      if (!(x >= edge && x < w - edge && y >= edge && y < h - edge)) continue;
      if (n < cap) {
        out[n].x = (float)x; out[n].y = (float)y; out[n].size = 7.f; out[n].angle = -1.f;
        out[n].response = (float)s; out[n].octave = 0;
      }
      ++n;
    }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* HarrisResponses (orb.cpp): blockSize 7, k = 0.04                                            */
/* ------------------------------------------------------------------------------------------ */
float orb_harris_at(const uint8_t* img, int stride, int x0, int y0) {
  const int blockSize = 7, r = blockSize / 2;
  const float harris_k = 0.04f;
  const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < blockSize; ++i)
    for (int j = 0; j < blockSize; ++j) {
      const uint8_t* ptr = img + (size_t)(y0 - r + i) * stride + (x0 - r + j);
      const int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-stride + 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[stride - 1]);
      const int Iy = (ptr[stride] - ptr[-stride]) * 2 + (ptr[stride - 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[-stride + 1]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
  return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ------------------------------------------------------------------------------------------ */
/* ICAngles (orb.cpp) with cv::fastAtan2 (core/src/mathfuncs_core.cpp)                          */
/* ------------------------------------------------------------------------------------------ */
float orb_fast_atan2(float y, float x) {
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

void orb_umax(int* umax /* [halfPatch + 2] */) {
  const int half = 15;
  int v, v0;
  const int vmax = cv_floor_f(half * sqrtf(2.f) / 2 + 1);
  const int vmin = (int)ceilf(half * sqrtf(2.f) / 2);
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)half * half - v * v));
  for (v = half, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

float orb_ic_angle_at(const uint8_t* img, int stride, int x, int y, const int* umax) {
  const int half_k = 15;
  const uint8_t* center = img + (size_t)y * stride + x;
  int m_01 = 0, m_10 = 0;
  for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
  for (int v = 1; v <= half_k; ++v) {
    int v_sum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orb_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------------------------ */
/* KeyPointsFilter::retainBest (keypoint.cpp): keep everything with response >= the n-th        */
/* largest; deterministic: survivors keep their order.                                          */
/* ------------------------------------------------------------------------------------------ */
static int cmp_float_desc(const void* a, const void* b) {
  const float x = *(const float*)a, y = *(const float*)b;
  return x > y ? -1 : x < y ? 1 : 0;
}
int orb_retain_best(orb_keypoint* kp, int n, int n_points) {
  if (n_points < 0 || n <= n_points) return n;
  if (n_points == 0) return 0;
  float* r = (float*)malloc(sizeof(float) * (size_t)n);
  for (int i = 0; i < n; ++i) r[i] = kp[i].response;
  qsort(r, (size_t)n, sizeof(float), cmp_float_desc);
  const float ambiguous = r[n_points - 1];
  free(r);
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (kp[i].response >= ambiguous) kp[m++] = kp[i];
  return m;
}

/* keepStrongest (feature_adjuster.cpp:247-255): exactly N by |response|; ties by order. */
typedef struct { float key; int idx; } keyidx;
static int cmp_keyidx(const void* a, const void* b) {
  const keyidx* x = (const keyidx*)a; const keyidx* y = (const keyidx*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;
}
int orb_keep_strongest(orb_keypoint* kp, int n, int N) {
  if (n <= N) return n;
  keyidx* k = (keyidx*)malloc(sizeof(keyidx) * (size_t)n);
  for (int i = 0; i < n; ++i) { k[i].key = fabsf(kp[i].response); k[i].idx = i; }
  qsort(k, (size_t)n, sizeof(keyidx), cmp_keyidx);
  char* keep = (char*)calloc((size_t)n, 1);
  for (int i = 0; i < N; ++i) keep[k[i].idx] = 1;
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (keep[i]) kp[m++] = kp[i];
  free(k); free(keep);
  return m;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::ORB::create(10000, 1.2f, 8, 15, 0, 2, HARRIS_SCORE, 31, fastThreshold)->detect(img,      */
/* keypoints, mask) on one (sub-)image  -- feature_adjuster.cpp:94 + orb.cpp computeKeyPoints   */
/* ------------------------------------------------------------------------------------------ */
int orb_detect(const uint8_t* img, const uint8_t* mask, int cols, int rows, int stride, int mstride,
               int fast_threshold, orb_keypoint* out, int cap) {
  enum { NL = 8 };
  const int nfeatures = 10000, edgeThreshold = 15, patchSize = 31;
  float scale[NL];
  int lw[NL], lh[NL];
  orb_level_geometry(cols, rows, NL, scale, lw, lh);
  uint8_t* limg[NL]; uint8_t* lmask[NL];
  limg[0] = NULL; lmask[0] = NULL;
  for (int l = 1; l < NL; ++l) {
    limg[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l] + 1);
    lmask[l] = mask ? (uint8_t*)malloc((size_t)lw[l] * lh[l] + 1) : NULL;
  }
  orb_build_pyramid(img, mask, cols, rows, stride, mstride, NL, lw, lh, limg, lmask);

  int nfeaturesPerLevel[NL];
  {
    const float factor = (float)(1.0 / (double)1.2f);
    float ndesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)NL));
    int sum = 0;
    for (int l = 0; l < NL - 1; ++l) {
      nfeaturesPerLevel[l] = cv_round_f(ndesired);
      sum += nfeaturesPerLevel[l];
      ndesired *= factor;
    }
    nfeaturesPerLevel[NL - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
  }
  int umax[17];
  orb_umax(umax);

  int n_out = 0;
  for (int l = 0; l < NL; ++l) {
    const uint8_t* I = l ? limg[l] : img;
    const int S = l ? lw[l] : stride;
    const uint8_t* Mk = mask ? (l ? lmask[l] : mask) : NULL;
    const int w = lw[l], h = lh[l];
    if (w < 7 || h < 7) continue;
    uint8_t* score = (uint8_t*)malloc((size_t)w * h);
    /* the score map works on a tightly packed copy when the level is a strided view */
    uint8_t* packed = NULL;
    uint8_t* mpacked = NULL;
    if (S != w) {
      packed = (uint8_t*)malloc((size_t)w * h);
      for (int y = 0; y < h; ++y) memcpy(packed + (size_t)y * w, I + (size_t)y * S, (size_t)w);
    }
    if (Mk && l == 0 && mstride != w) {
      mpacked = (uint8_t*)malloc((size_t)w * h);
      for (int y = 0; y < h; ++y) memcpy(mpacked + (size_t)y * w, Mk + (size_t)y * mstride, (size_t)w);
    }
    const uint8_t* Ip = packed ? packed : I;
    const uint8_t* Mp = mpacked ? mpacked : Mk;
    orb_fast_score_map(Ip, w, h, w, fast_threshold, score);
    const int capl = w * h / 4 + 16;
    orb_keypoint* kp = (orb_keypoint*)malloc(sizeof(orb_keypoint) * (size_t)capl);
    int n = orb_fast_keypoints(score, Mp, w, h, edgeThreshold, kp, capl);
    if (n > capl) n = capl;
    n = orb_retain_best(kp, n, 2 * nfeaturesPerLevel[l]); /* HARRIS_SCORE: 2 * featuresNum */
    for (int i = 0; i < n; ++i) {
      kp[i].octave = l;
      kp[i].size = patchSize * scale[l];
      kp[i].response = orb_harris_at(Ip, w, cv_round_f(kp[i].x), cv_round_f(kp[i].y));
    }
    n = orb_retain_best(kp, n, nfeaturesPerLevel[l]);
    for (int i = 0; i < n; ++i) {
      kp[i].angle = orb_ic_angle_at(Ip, w, cv_round_f(kp[i].x), cv_round_f(kp[i].y), umax);
      kp[i].x *= scale[l];
      kp[i].y *= scale[l];
      if (n_out < cap) out[n_out] = kp[i];
      ++n_out;
    }
    free(kp); free(score); free(packed); free(mpacked);
  }
  for (int l = 1; l < NL; ++l) { free(limg[l]); free(lmask[l]); }
  return n_out < cap ? n_out : cap;
}

/* ------------------------------------------------------------------------------------------ */
/* The reference's detector object: 3x3 grid of threshold-adaptive ORB detectors                */
/* (features.cpp:42-60, feature_adjuster.cpp:185-317).  state->thresh[cell] persists across     */
/* frames (the "Video" in the class names).                                                     */
/* ------------------------------------------------------------------------------------------ */
void orb_grid_state_init(orb_grid_state* st, int max_keypoints, int grid_res, int max_iters) {
  memset(st, 0, sizeof(*st));
  st->grid = grid_res;
  st->max_iters = max_iters;                       /* adjuster_max_iterations (5) */
  const int mn = max_keypoints;                    /* features.cpp:47 */
  const int mx = (int)(mn * 1.5);                  /* :48 */
  const int cells = grid_res * grid_res;
  st->cell_min = (int)roundf(mn / (float)cells);   /* :52 */
  st->cell_max = (int)roundf(mx / (float)cells);   /* :53 */
  st->max_total = mx;                              /* VideoGridAdaptedFeatureDetector(detector, max, ...) */
  st->edge = 31;                                   /* feature_adjuster.h: edgeThreshold = 31 */
  for (int i = 0; i < cells && i < ORB_MAX_CELLS; ++i) st->thresh[i] = 20.0; /* DetectorAdjuster("ORB", 20) */
}

int orb_grid_detect(orb_grid_state* st, const uint8_t* img, const uint8_t* mask, int cols, int rows,
                    orb_keypoint* out, int cap) {
  const int G = st->grid;
  const int maxPerCell = st->max_total / (G * G); /* feature_adjuster.cpp:292 */
  int n_out = 0;
  orb_keypoint* cellkp = (orb_keypoint*)malloc(sizeof(orb_keypoint) * (size_t)cap);
  for (int i = 0; i < G; ++i) {
    const int rowstart = (i * rows) / G - st->edge > 0 ? (i * rows) / G - st->edge : 0;
    const int rowend = rows < ((i + 1) * rows) / G + st->edge ? rows : ((i + 1) * rows) / G + st->edge;
    for (int j = 0; j < G; ++j) {
      const int colstart = (j * cols) / G - st->edge > 0 ? (j * cols) / G - st->edge : 0;
      const int colend = cols < ((j + 1) * cols) / G + st->edge ? cols : ((j + 1) * cols) / G + st->edge;
      const int cw = colend - colstart, ch = rowend - rowstart;
      const uint8_t* sub = img + (size_t)rowstart * cols + colstart;
      const uint8_t* submask = mask ? mask + (size_t)rowstart * cols + colstart : NULL;
      double* thr = &st->thresh[j + i * G];
      /* VideoDynamicAdaptedFeatureDetector::detect (feature_adjuster.cpp:185-224) */
      int iter_count = st->max_iters;
      int checked_for_non_zero_mask = 0;
      int n = 0;
      do {
        n = orb_detect(sub, submask, cw, ch, cols, cols, (int)*thr, cellkp, cap);
        if (n < st->cell_min) {
          *thr *= 0.7; /* tooFew: decrease_factor */
          if (*thr < 2) *thr = 2;
          if (n == 0 && !checked_for_non_zero_mask) {
            checked_for_non_zero_mask = 1;
            int nz = 0;
            if (submask)
              for (int y = 0; y < ch && !nz; ++y)
                for (int x = 0; x < cw; ++x)
                  if (submask[(size_t)y * cols + x]) { nz = 1; break; }
            if (!nz) break; /* hasNonZero(mask) == false */
          }
        } else if (n > st->cell_max) {
          *thr *= 1.3; /* tooMany */
          if (*thr > 10000) *thr = 10000;
          break;
        } else
          break;
        iter_count--;
      } while (iter_count > 0 && (*thr > 2 && *thr < 10000)); /* good() */
      n = orb_keep_strongest(cellkp, n, maxPerCell);
      /* aggregateKeypointsPerGridCell (:259-282) */
      for (int k = 0; k < n; ++k) {
        cellkp[k].x += colstart;
        cellkp[k].y += rowstart;
        if (n_out < cap) out[n_out] = cellkp[k];
        ++n_out;
      }
    }
  }
  free(cellkp);
  return n_out < cap ? n_out : cap;
}

/* ------------------------------------------------------------------------------------------ */
/* GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8U: separable filter  */
/* with the kernel in 8-bit fixed point (smooth.cpp createGaussianFilter -> filter.cpp          */
/* createSeparableLinearFilter: bits = 8 for rows and columns, FixedPtCastEx<int,uchar>(16))    */
/* ------------------------------------------------------------------------------------------ */
void orb_gauss7_kernel_fixed(int k[7]) {
  const double sigma = 2.0;
  float kf[7];
  double sum = 0;
  const double scale2X = -0.5 / (sigma * sigma);
  for (int i = 0; i < 7; ++i) {
    const double x = i - 3;
    const double t = exp(scale2X * x * x);
    kf[i] = (float)t;
    sum += kf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; ++i) {
    kf[i] = (float)(kf[i] * sum);
    k[i] = cv_round_d((double)kf[i] * 256.0); /* kernel.convertTo(CV_32S, 1 << 8) */
  }
}

void orb_gaussian_blur7(const uint8_t* src, int w, int h, int stride, uint8_t* dst) {
  int k[7];
  orb_gauss7_kernel_fixed(k);
  int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 7; ++i) s += k[i] * src[(size_t)y * stride + reflect101(x + i - 3, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 7; ++i) s += k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
      const int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* cv::ORB::create()->compute(gray, keypoints, descriptors)  (features.cpp:117-119,             */
/* orb.cpp detectAndCompute with useProvidedKeypoints + computeOrbDescriptors, WTA_K = 2)       */
/* ------------------------------------------------------------------------------------------ */
#include "orb_pattern.inc" /* static const int8_t orb_bit_pattern_31[256 * 4 * ... ] */

int orb_compute(const uint8_t* img, int cols, int rows, orb_keypoint* kp, int n, uint8_t* desc) {
  const int edgeThreshold = 31;
  /* KeyPointsFilter::runByImageBorder(keypoints, image.size(), edgeThreshold) */
  int m = 0;
  int max_level = 0;
  for (int i = 0; i < n; ++i) {
    const float x = kp[i].x, y = kp[i].y;
    if (x >= edgeThreshold && x < cols - edgeThreshold && y >= edgeThreshold && y < rows - edgeThreshold)
      kp[m++] = kp[i];
  }
  n = m;
  for (int i = 0; i < n; ++i) {
    const int lv = kp[i].octave > 0 ? kp[i].octave : 0;
    if (lv > max_level) max_level = lv;
  }
  const int nlevels = max_level + 1;
  /* not sorted by level -> stable regroup by level */
  {
    orb_keypoint* t = (orb_keypoint*)malloc(sizeof(orb_keypoint) * (size_t)(n ? n : 1));
    int c = 0;
    for (int l = 0; l < nlevels; ++l)
      for (int i = 0; i < n; ++i)
        if (kp[i].octave == l) t[c++] = kp[i];
    memcpy(kp, t, sizeof(orb_keypoint) * (size_t)c);
    n = c;
    free(t);
  }
  float scale[32];
  int lw[32], lh[32];
  orb_level_geometry(cols, rows, nlevels, scale, lw, lh);
  uint8_t* limg[32]; uint8_t* lblur[32];
  limg[0] = NULL;
  for (int l = 1; l < nlevels; ++l) limg[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l] + 1);
  orb_build_pyramid(img, NULL, cols, rows, cols, 0, nlevels, lw, lh, limg, NULL);
  for (int l = 0; l < nlevels; ++l) {
    lblur[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l] + 1);
    orb_gaussian_blur7(l ? limg[l] : img, lw[l], lh[l], lw[l], lblur[l]);
  }
  for (int j = 0; j < n; ++j) {
    const int l = kp[j].octave;
    const float sc = 1.f / scale[l];
    float angle = kp[j].angle;
    angle *= (float)(M_PI / 180.f);
    /* cos/sin: the double functions rounded to float (the reference's cosf/sinf are
     * not bit-reproducible across libm / device math libraries) */
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const int cx = cv_round_f(kp[j].x * sc), cy = cv_round_f(kp[j].y * sc);
    const uint8_t* raw = l ? limg[l] : img;
    const uint8_t* blur = lblur[l];
    const int w = lw[l], h = lh[l];
    for (int i = 0; i < 32; ++i) {
      int val = 0;
      for (int bit = 0; bit < 8; ++bit) {
        int t[2];
        for (int e = 0; e < 2; ++e) {
          const int idx = (i * 16 + bit * 2 + e) * 2;
          const float px = (float)orb_bit_pattern_31[idx], py = (float)orb_bit_pattern_31[idx + 1];
          const float x = px * a - py * b;
          const float y = px * b + py * a;
          const int ix = cx + cv_round_f(x), iy = cy + cv_round_f(y);
          /* inside the level: blurred pixel; outside: the UNBLURRED reflect-101 border that
           * copyMakeBorder wrote before the in-place GaussianBlur of the level's ROI */
          if (ix >= 0 && ix < w && iy >= 0 && iy < h)
            t[e] = blur[(size_t)iy * w + ix];
          else
            t[e] = raw[(size_t)reflect101(iy, h) * w + reflect101(ix, w)];
        }
        val |= (t[0] < t[1]) << bit;
      }
      desc[(size_t)j * 32 + i] = (uint8_t)val;
    }
  }
  for (int l = 0; l < nlevels; ++l) { if (l) free(limg[l]); free(lblur[l]); }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* Node::Node feature path (node.cpp:139-210) for a gray image + float depth + mono8 mask       */
/* ------------------------------------------------------------------------------------------ */
static int remove_depthless(orb_keypoint* kp, int n, const float* depth, int rows, int cols) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float px = kp[i].x, py = kp[i].y;
    if (px >= (float)cols || px < 0 || py >= (float)rows || py < 0 || isnan(px) || isnan(py)) continue;
    int r = (int)roundf(py), c = (int)roundf(px);
    if (r >= rows) r = rows - 1;
    if (c >= cols) c = cols - 1;
    if (isnan(depth[(size_t)r * cols + c])) continue;
    kp[m++] = kp[i];
  }
  return m;
}

/* getMinDepthInNeighborhood (misc.cpp:774-793); the same restatement as rgbd_oracle.c's orc_min_depth_in_neighborhood
 * (which is the one pinned on the reference function) -- repeated here because this file is also linked on its own. */
static float orc_min_depth_in_neighborhood(const float* depth, int rows, int cols, float cx, float cy, float diameter) {
  const int radius = (int)((diameter - 1) / 2);
  int top = (int)(cy - (float)radius); top = top < 0 ? 0 : top;
  int left = (int)(cx - (float)radius); left = left < 0 ? 0 : left;
  int bot = (int)(cy + (float)radius); bot = bot > rows ? rows : bot;
  int right = (int)(cx + (float)radius); right = right > cols ? cols : right;
  float minv = 3.402823466e+38f;
  int found = 0;
  for (int r = top; r < bot; ++r)
    for (int c = left; c < right; ++c) {
      const float v = depth[(size_t)r * (size_t)cols + (size_t)c];
      if (v < minv) { minv = v; found = 1; }
    }
  if (!found || minv == 0.0f) return NAN;
  return minv;
}
static int remove_depthless_min_depth(orb_keypoint* kp, int n, const float* depth, int rows, int cols) {
  int m = 0;  /* node.cpp:82: Z = getMinDepthInNeighborhood(depth, p2d, size) */
  for (int i = 0; i < n; ++i) {
    const float px = kp[i].x, py = kp[i].y;
    if (px >= (float)cols || px < 0 || py >= (float)rows || py < 0 || isnan(px) || isnan(py)) continue;
    if (isnan(orc_min_depth_in_neighborhood(depth, rows, cols, px, py, kp[i].size))) continue;
    kp[m++] = kp[i];
  }
  return m;
}

static int g_use_feature_min_depth = 0;
void orb_set_use_feature_min_depth(int on) { g_use_feature_min_depth = on; }

int orb_node_features(orb_grid_state* st, const uint8_t* gray, const uint8_t* mask, const float* depth,
                      int cols, int rows, int max_keypoints, orb_keypoint* kp, int cap, uint8_t* desc) {
  int n = orb_grid_detect(st, gray, mask, cols, rows, kp, cap);   /* node.cpp:160 */
  n = g_use_feature_min_depth ? remove_depthless_min_depth(kp, n, depth, rows, cols)
                              : remove_depthless(kp, n, depth, rows, cols);                  /* :186 */
  if (n > max_keypoints) {                                          /* :188-191 */
    /* retainBest keeps ties of the n-th response, resize() then cuts: the survivors are the
     * max_keypoints strongest, ties by order */
    keyidx* k = (keyidx*)malloc(sizeof(keyidx) * (size_t)n);
    for (int i = 0; i < n; ++i) { k[i].key = kp[i].response; k[i].idx = i; }
    qsort(k, (size_t)n, sizeof(keyidx), cmp_keyidx);
    char* keep = (char*)calloc((size_t)n, 1);
    for (int i = 0; i < max_keypoints; ++i) keep[k[i].idx] = 1;
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (keep[i]) kp[m++] = kp[i];
    n = m;
    free(k); free(keep);
  }
  n = orb_compute(gray, cols, rows, kp, n, desc);                  /* :202 */
  return n; /* removeDepthless (:206) removes nothing new; projectTo3D is rgbd_oracle.c */
}
const int8_t* orb_pattern(void) { return orb_bit_pattern_31; }
