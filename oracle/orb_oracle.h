/* oracle/orb_oracle.h -- TEST INFRASTRUCTURE ONLY (see orb_oracle.c: PARITY UNPINNED). */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORB_MAX_CELLS 64

typedef struct {
  float x, y;      /* cv::KeyPoint::pt */
  float size;
  float angle;     /* degrees, fastAtan2 */
  float response;  /* FAST score, then Harris response */
  int32_t octave;
} orb_keypoint;

typedef struct {
  int32_t grid, max_iters, cell_min, cell_max, max_total, edge;
  double thresh[ORB_MAX_CELLS]; /* DetectorAdjuster::thresh_ of every grid cell, persists across frames */
} orb_grid_state;

void orb_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw,
                          int dh, int dstride);
void orb_level_geometry(int cols, int rows, int nlevels, float* scale, int* lw, int* lh);
void orb_build_pyramid(const uint8_t* img, const uint8_t* mask, int cols, int rows, int stride,
                       int mstride, int nlevels, const int* lw, const int* lh, uint8_t** out_img,
                       uint8_t** out_mask);
int orb_fast_score_at(const uint8_t* img, int stride, int x, int y, int threshold);
void orb_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score);
int orb_fast_keypoints(const uint8_t* score, const uint8_t* mask, int w, int h, int edge,
                       orb_keypoint* out, int cap);
float orb_harris_at(const uint8_t* img, int stride, int x0, int y0);
float orb_fast_atan2(float y, float x);
void orb_umax(int* umax);
float orb_ic_angle_at(const uint8_t* img, int stride, int x, int y, const int* umax);
int orb_retain_best(orb_keypoint* kp, int n, int n_points);
int orb_keep_strongest(orb_keypoint* kp, int n, int N);
int orb_detect(const uint8_t* img, const uint8_t* mask, int cols, int rows, int stride, int mstride,
               int fast_threshold, orb_keypoint* out, int cap);
void orb_grid_state_init(orb_grid_state* st, int max_keypoints, int grid_res, int max_iters);
int orb_grid_detect(orb_grid_state* st, const uint8_t* img, const uint8_t* mask, int cols, int rows,
                    orb_keypoint* out, int cap);
void orb_gauss7_kernel_fixed(int k[7]);
void orb_gaussian_blur7(const uint8_t* src, int w, int h, int stride, uint8_t* dst);
int orb_compute(const uint8_t* img, int cols, int rows, orb_keypoint* kp, int n, uint8_t* desc);
int orb_node_features(orb_grid_state* st, const uint8_t* gray, const uint8_t* mask, const float* depth,
                      int cols, int rows, int max_keypoints, orb_keypoint* kp, int cap, uint8_t* desc);
void orb_set_use_feature_min_depth(int on);  /* parameter "use_feature_min_depth" for orb_node_features (node.cpp:82) */
const int8_t* orb_pattern(void);

#ifdef __cplusplus
}
#endif
#endif
