/*
 * oracle/rgbd_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the rgbdslam_v2 visual front-end pair path:
 * ORB brute-force Hamming matching + match truncation + RANSAC rigid-transform
 * estimation, plus the per-frame depth filter / back-projection.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported baseline.  The product path
 * (rgbdslam_v2_amd/csrc, librgbdfe.so) never links, loads or calls anything in
 * oracle/.
 *
 * Parity status (see DESIGN.md section "Oracle"):
 *   - orc_hamming_nn / orc_hamming_nn_batch: PINNED against the reference's own
 *     bruteForceSearchORB (src/features.cpp:163-182) compiled into
 *     oracle/_ref/libref_bforb.so and against tests/golden/hamming_*.npz.
 *   - everything that leans on PCL / Eigen / OpenCV arithmetic that is not in the
 *     reference tree (TransformationFromCorrespondences, JacobiSVD, LLT):
 *     "parity unpinned" -- restated from the published algorithm, anchored on
 *     the reference's call sites.
 */
#ifndef RGBD_ORACLE_H
#define RGBD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_MATCHES 320

typedef struct {
  int32_t  max_matches;          /* parameter_server.cpp:86  (300)  */
  int32_t  min_matches;          /* parameter_server.cpp:85  (20)   */
  int32_t  ransac_iterations;    /* parameter_server.cpp:101 (200)  */
  float    max_dist_for_inliers; /* parameter_server.cpp:100 (3.0)  */
  double   depth_cov;            /* frozen depth_covariance() value, misc2.h:30-35 */
  uint32_t seed;                 /* replaces srand(clock()), node.cpp:1102 */
} orc_params;

typedef struct {
  int32_t id1, id2;              /* edge ids; -1,-1 <=> no edge (node.cpp:1419-1422) */
  int32_t n_all, n_inl;
  float   rmse;
  float   T[16];                 /* Matrix4f, column-major, maps new->old frame */
  double  info_scale;            /* informationMatrix = I6 * info_scale (node.cpp:1335) */
  int32_t valid_iterations, real_iterations;
  int32_t all_q[ORC_MAX_MATCHES], all_t[ORC_MAX_MATCHES], all_hd[ORC_MAX_MATCHES];
  int32_t inl_idx[ORC_MAX_MATCHES]; /* positions in all_* of the inlier matches */
} orc_result;

/* features.cpp:163-182 */
int orc_hamming_nn(const uint64_t* v, const uint64_t* search_array, uint32_t size,
                   int* result_index);
void orc_hamming_nn_batch(const uint8_t* qdesc, uint32_t nq, const uint8_t* tdesc,
                          uint32_t nt, int32_t* out_hd, int32_t* out_idx);
/* node.cpp:561-576 + 520-531,674 ; returns number of matches (sorted by hd, then queryIdx) */
int orc_feature_matching_orb(const uint8_t* qdesc, uint32_t nq, const uint8_t* tdesc,
                             uint32_t nt, int max_matches,
                             int32_t* mq, int32_t* mt, int32_t* mhd);
/* deterministic replacement of rand(): node.cpp:1024-1047 */
uint32_t orc_rand31(uint32_t seed, uint32_t uid, uint32_t iter, uint32_t k);
uint32_t orc_pair_uid(int32_t query_id, int32_t train_id);
int orc_sample4(uint32_t seed, uint32_t uid, uint32_t iter, uint32_t n, uint32_t ids[4]);
/* transformation_estimation_euclidean.cpp:7-61 + PCL TransformationFromCorrespondences */
void orc_fit_transform(const float* qxyz1, const float* txyz1, const int32_t* mq,
                       const int32_t* mt, const int32_t* sel, int nsel, float T[16]);
void orc_svd3(const float C[9], float U[9], float S[3], float V[9]);
/* misc.cpp:697-770 */
void orc_raster_cov(double* cx, double* cy);
double orc_error_function2(const float x1[4], const float x2[4], const double T[16],
                           double depth_cov);
/* node.cpp:968-1020 */
int orc_compute_inliers_and_error(const float* qxyz1, const float* txyz1,
                                  const int32_t* mq, const int32_t* mt, int n,
                                  const float T[16], double sq_max_dist, double depth_cov,
                                  int32_t* inl, double* mean_error);
/* node.cpp:1074-1277 */
int orc_ransac(const float* qxyz1, const float* txyz1, const int32_t* mq,
               const int32_t* mt, int n, const orc_params* prm, uint32_t uid,
               float T[16], float* rmse, int32_t* inl, int* n_inl,
               int* valid_iterations, int* real_iterations);
/* node.cpp:1305-1429 */
void orc_match_node_pair(const uint8_t* qdesc, const float* qxyz1, uint32_t nq, int32_t qid,
                         const uint8_t* tdesc, const float* txyz1, uint32_t nt, int32_t tid,
                         const orc_params* prm, orc_result* out);
/* pair-parallel driver for the CPU baseline (graph_manager.cpp:541-548) */
void orc_match_pairs_mt(const uint8_t* const* desc, const float* const* xyz1,
                        const uint32_t* counts, const int32_t* node_ids,
                        const int32_t* pair_q, const int32_t* pair_t, int n_pairs,
                        const orc_params* prm, orc_result* out, int n_threads);
/* node.cpp:67-97 and 900-965, misc2.h:49-65 ; returns number of kept keypoints */
int orc_project_to_3d(const float* kp_xy, int n_kp, const float* depth, int rows, int cols,
                      double fx, double fy, double cx, double cy, double depth_scaling,
                      int max_keypoints, int32_t* kept_idx, float* xyz1);
/* "use_feature_min_depth" variants (misc.cpp:774-793, node.cpp:82, :940): kp_size = cv::KeyPoint::size per keypoint */
float orc_min_depth_in_neighborhood(const float* depth, int rows, int cols, float cx, float cy, float diameter);
int orc_remove_depthless_min_depth(const float* kp_xy, const float* kp_size, int n_kp, const float* depth, int rows,
                                   int cols, int32_t* kept_idx);
int orc_project_to_3d_min_depth(const float* kp_xy, const float* kp_size, int n_kp, const float* depth, int rows,
                                int cols, double fx, double fy, double cx, double cy, double depth_scaling,
                                int max_keypoints, int32_t* kept_idx, float* xyz1);
int orc_num_cores(void);
/* SiftGPUWrapper::match (sift_gpu_wrapper.cpp:169-227) over the CUDA SiftMatchGPU kernels */
/* SURVEY 8(f) rows 3 + 2: depthToCV8UC1, createXYZRGBPointCloud, observationLikelihood (misc.cpp) */
void orc_depth_to_mono8_f32(const float* depth, size_t n, uint8_t* mono8);
void orc_depth_u16_to_mono8_f32(const uint16_t* depth_mm, size_t n, uint8_t* mono8, float* depth_m);
void orc_create_point_cloud(const float* depth, int rows, int cols, const uint8_t* rgb, int channels,
                            int encoding_bgr, double fx, double fy, double cx, double cy,
                            double depth_scaling, double min_depth, int skip_step, float* cloud);
void orc_observation_likelihood(const float* new_cloud, const float* old_cloud, int ch, int cw,
                                const float* T, double fx, double fy, double cx, double cy,
                                int cloud_skip, int skip_step, double depth_cov, uint32_t counts[4]);
void orc_emm_erf_boundaries(double* q_lo, double* q_hi);
int orc_observation_criterion_met(unsigned int inliers, unsigned int outliers, unsigned int all,
                                  double obs_thresh, double* quality);
/* a20: projectTo3DSiftGPU (node.cpp:695-769) and squareroot_descriptor_space (node.cpp:1557-1571) */
int orc_project_to_3d_cloud(const float* kp_xy, int n_kp, const float* cloud, int rows, int cols,
                            double maximum_depth, int max_keypoints, int32_t* kept_idx, float* xyz1);
int orc_project_to_3d_sift(const float* kp_xy, int n_kp, const float* depth, int rows, int cols,
                           double fx, double fy, double cx, double cy, double depth_scaling,
                           int max_keypoints, int32_t* kept_idx, float* xyz1);
void orc_gather_rows_f32(const float* in, const int32_t* kept_idx, int n, int dim, float* out);
void orc_root_sift(float* desc, int n_rows, int dim);
int orc_sift_match(const float* d1, int n1, const float* d2, int n2, int32_t* mq, int32_t* mt,
                   float* dist_out);
int orc_g2o_refine(const float* qxyz1, const float* txyz1, const float* qkp, const float* tkp, const int32_t* mq,
                   const int32_t* mt, const int32_t* sel, int nsel, float T[16], int iterations, double depth_cov);
int orc_g2o_block(const float* qxyz1, const float* txyz1, const float* qkp, const float* tkp, const int32_t* mq,
                  const int32_t* mt, int n, const orc_params* prm, int g2o_iterations, float T[16], float* rmse_io,
                  int32_t* matches, int* n_matches_io, int* valid_iterations_io);
void orc_match_node_pair_g2o(const uint8_t* qdesc, const float* qxyz1, const float* qkp, uint32_t nq, int32_t qid,
                             const uint8_t* tdesc, const float* txyz1, const float* tkp, uint32_t nt, int32_t tid,
                             const orc_params* prm, int g2o_iterations, orc_result* out);
int orc_flann_match(const float* qdesc, int nq, const float* tdesc, int nt, int dim, double nn_distance_ratio,
                    int32_t* mq, int32_t* mt, float* md);
void orc_match_float_node_pair(const float* qdesc, const float* qxyz1, int nq, int32_t qid, const float* tdesc,
                               const float* txyz1, int nt, int32_t tid, int dim, double nn_distance_ratio,
                               const orc_params* prm, orc_result* out, float* all_dist);
void orc_match_sift_node_pair(const float* qdesc, const float* qxyz1, int nq, int32_t qid,
                              const float* tdesc, const float* txyz1, int nt, int32_t tid,
                              const orc_params* prm, orc_result* out, float* all_dist);


/* Sensitivity harness: alternative roundings of the third-party arithmetic (Eigen 3.2 JacobiSVD / LLT, PCL 1.7
 * TransformationFromCorrespondences) that cannot be pinned here.  0 = the restatement the kernels follow. */
#define ORC_VAR_LLT_RECIPROCAL     0x001u  /* LLT column scaling  A21 *= 1/x  instead of  A21 /= x  (misc.cpp:763) */
#define ORC_VAR_SOLVE_ORDER        0x002u  /* triangular solves: term-by-term subtraction, reciprocal pivots */
#define ORC_VAR_SVD_SWEEP_ORDER    0x004u  /* Jacobi sweeps visit (2,1) (2,0) (1,0) instead of (1,0) (2,0) (2,1) */
#define ORC_VAR_SVD_PAIR_THRESHOLD 0x008u  /* rotation skipped relative to the pair's own diagonal (older Eigen) */
#define ORC_VAR_SVD_NO_PRESCALE    0x010u  /* no division of the matrix by its largest coefficient */
#define ORC_VAR_PCL_COV_ASSOC      0x020u  /* (1-a)*C + ((1-a)*a)*d2*d1^T  instead of  (1-a)*(C + a*d2*d1^T) */
#define ORC_VAR_ROT_ASSOC          0x040u  /* R = U*(S*V^T), dot products summed from the last term */
#define ORC_VAR_COV_ASSOC          0x080u  /* errorFunction2: R^T*(cov1*R) instead of (R^T*cov1)*R */
#define ORC_VAR_ALL                0x0FFu
void orc_set_variant(unsigned flags);
unsigned orc_get_variant(void);
/* per-pair trace of orc_match_pairs_mt: a hash of the inlier set the adopted transform was fitted from (0 = none) */
void orc_set_trace(uint64_t* per_pair);

#ifdef __cplusplus
}
#endif
#endif
