// sift_extract.h -- per-context workspace of the SIFT extraction path (sift_extract.hip): SiftGPUWrapper::detect
// (src/sift_gpu_wrapper.cpp:113-167) = SiftGPU's pyramid / DoG / keypoint / orientation / descriptor pipeline
// (external/SiftGPU/src/SiftGPU/ProgramCU.cu:113-1186, PyramidCU.cpp, SiftPyramid.cpp:49-168) with the options the
// wrapper's constructor sets (:29-88): -s 1 -tc2 <max_keypoints> -fo -1 -unn -d 5 -e 10.0 -ofix-not.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace rgbdfe {

struct SiftKey { float x, y, s, o; };  // SiftGPU::SiftKeypoint: level-0 pixel coordinates, scale, orientation (radians)

struct SiftExtractor {
  static constexpr int kDogLevels = 5;            // "-d 5"
  static constexpr int kLevels = kDogLevels + 3;  // Gaussian levels per octave (level_min = -1 .. level_max = 6)
  static constexpr int kMaxOctaves = 12;
  struct Octave { int w, h; size_t plane; float* g[kLevels]; };  // w = the padded width (multiple of 4) every kernel uses
  ~SiftExtractor();
  void release();
  static constexpr int kMaxBatch = 8;   // frames per launch chain of run_batch
  // nf <= kMaxBatch frames of one size through ONE launch chain: keys[f] (n x 4) + desc[f] (n x 128, unnormalised) in SiftGPU's
  // output order.  Frames are independent (the pipeline keeps no state between images): the batch only shares launches.
  // desc[f] points into a pinned buffer of this object (valid until the next call): the caller copies from there once
  int run_batch(const uint8_t* const* gray, int nf, int rows, int cols, int max_features, std::vector<SiftKey>* keys,
                const float** desc, hipStream_t s, std::string& err);
  int begin_batch(const uint8_t* const* gray, int nf, int rows, int cols, hipStream_t s, std::string& err);
  int finish_batch(int max_features, std::vector<SiftKey>* keys, const float** desc, hipStream_t s, std::string& err);
  // finish_batch in its three wait-work-enqueue steps (sift_extract.hip)
  int finish_orientations(int max_features, hipStream_t s, std::string& err);
  int finish_descriptors(hipStream_t s, std::string& err);
  int finish_outputs(std::vector<SiftKey>* keys, const float** desc, hipStream_t s, std::string& err);
  struct FrameState {   // a frame of the batch between those steps
    std::vector<int> cnt, off, level_num;
    int feature_num = 0, total = 0, base = 0, erased = 0;
    std::vector<float> list, keybuf;
  };
  std::vector<FrameState> fs;
  int fin_nf = 0, fin_stage = 0, fin_max_features = 0, fin_grand = 0, fin_grand2 = 0;
  double stage_us = 0;  // host time spent copying callers' images into the pinned stage (RGBDFE_SIFT_TIMING reports it)
  int pending_nf = 0;   // frames of the batch begin_batch enqueued and finish_batch has not collected yet
  int enqueue_begin(int nf, hipStream_t s, std::string& err);
  hipGraph_t begin_graph[kMaxBatch + 1] = {};          // begin_batch's launch chain per batch size, captured on first use
  hipGraphExec_t begin_exec[kMaxBatch + 1] = {};
  bool begin_capture_failed = false;
  int run(const uint8_t* gray, int rows, int cols, int max_features, std::vector<SiftKey>& keys, const float*& desc,
          hipStream_t s, std::string& err) {
    return run_batch(&gray, 1, rows, cols, max_features, &keys, &desc, s, err);
  }
  // descriptors for the caller's keypoints (x, y, scale, orientation): SiftGPUWrapper::detect's second mode
  int describe(const uint8_t* gray, int rows, int cols, const SiftKey* keys_in, int n, const float** desc, hipStream_t s,
               std::string& err);
  int enqueue_pyramid(const uint8_t* const* gray, int nf, hipStream_t s, std::string& err);
  // stage access for the parity tests: a Gaussian plane of the latest call's FIRST frame / the keypoint candidates of one
  // (octave, dog level) of that frame as (x, y, sign, dx, dy, ds) in list order, before the feature-count limits
  int debug_plane(int octave, int level, std::vector<float>& out, int* w, int* h, hipStream_t s);
  int debug_candidates(int octave, int dog_level, std::vector<float>& out, hipStream_t s);

  // parameters (SiftParam::ParseSiftParam, SiftGPU.cpp:433-473)
  float sigma0 = 0, sigmak = 0, dsigma0 = 0, sigma[kLevels - 1] = {}, dog_threshold = 0, edge_threshold = 0;
  bool params_ready = false;
  void init_params();
  float initial_smooth_sigma(int octave_min) const;  // SiftParam::GetInitialSmoothSigma (SiftGPU.cpp:425-431)
  float level_sigma(int lev) const;                  // SiftParam::GetLevelSigma (SiftGPU.cpp:1200-1203)

  // geometry of the current image size
  int W = 0, H = 0, w4 = 0, octave_min = 0, octave_num = 0;
  Octave oct[kMaxOctaves];
  // device
  uint8_t* d_gray = nullptr; float* d_input = nullptr; float* d_up = nullptr;  // bytes, /255 floats, resampled base
  float* d_planes = nullptr; size_t planes_floats = 0;
  int8_t* d_flags = nullptr; size_t flags_bytes = 0;   // extremum sign per pixel of every (octave, dog level)
  int* d_rowcnt = nullptr; int* d_rowoff = nullptr; int* d_lvltot = nullptr; int total_rows = 0;
  struct LevelDesc { const float* g[4]; int8_t* flags; int w, h, row0; float pad; };
  LevelDesc* d_levels = nullptr;
  std::vector<LevelDesc> h_levels;
  std::vector<int> h_row2lvl;                          // level (octave * kDogLevels + dog level) of every stacked row
#ifndef RGBDFE_KEY_TILE_H
#define RGBDFE_KEY_TILE_H 8    // rows per extremum tile, a multiple of 4 (measured: 4 -> 224, 8 -> 193, 12 -> 205, 16 -> 236 us per 8 VGA frames)
#endif
  static constexpr int kKeyTileH = RGBDFE_KEY_TILE_H;
  struct KeyTile { int oct, x0, y0; };                 // a 64 x kKeyTileH pixel tile of an octave
  std::vector<KeyTile> h_key_tiles;
  struct OctDesc { size_t plane_off, flag_off; int w, h, row0, pad; };   // an octave inside a frame's planes / flags / rows
  std::vector<OctDesc> h_octs;
  void* d_octs = nullptr;
  float* d_cand = nullptr; size_t cand_cap = 0;        // candidates: 6 floats each, per level at its offset
  float4* d_feat = nullptr; size_t feat_cap = 0;       // feature list (x, y, scale, packed / final orientation)
  float* d_desc = nullptr; size_t desc_cap = 0;
  float* h_desc = nullptr; size_t h_desc_cap = 0;      // pinned: the descriptors of the latest call (128 floats per feature)
  int* h_counts = nullptr;                             // pinned: per-level totals, 64 per frame
  float* h_stage = nullptr; size_t stage_floats = 0;   // pinned staging for lists (all frames of a batch)
  // The three read-backs of a batch -- the levels' candidate counts, the oriented features, the descriptors -- are stored into
  // the pinned buffers by the kernels that produce them (posted PCIe writes beside the kernel) instead of being copied behind
  // them (a blit kernel each on the batch's stream: 16 % of a chunk's kernel time).  RGBDFE_SIFT_HOSTWRITE=0: the copies.
  bool host_write = true;
  uint8_t* h_gray = nullptr; size_t gray_cap = 0;      // pinned staging of the caller's (pageable) images
  void* d_key_tiles = nullptr; int n_key_tiles = 0;    // the (octave, x0, y0) tiles of the extremum launch
  void* d_jobs = nullptr; void* h_jobs = nullptr;      // the per-frame segment tables of the orientation / descriptor launches
  std::vector<int> lvl_count, lvl_off;                 // candidates per (octave, dog level) of the latest call's first frame
  int frames_cap = 0;                                  // frames the buffers hold (every device buffer is [frames_cap][...])
  size_t input_floats = 0;                             // per-frame strides: d_input; d_up = oct[0].plane; d_planes = planes_floats
  int prepare(int rows, int cols, int nf, hipStream_t s, std::string& err);
  int plan_geometry(int rows, int cols, std::string& err);   // sizes only (defined in sift_pyramid_kernels.h, as bind_levels)
  void bind_levels();
};

}  // namespace rgbdfe
