// orb_host.h -- per-context ORB workspace (device buffers + the detector's persistent state)
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "orb_internal.h"

namespace rgbdfe {

struct KpOut {  // cv::KeyPoint fields in use
  float x, y, size, angle, response;
  int octave;
};

struct OrbWorkspace {
  struct Cell { int x0, y0, w, h; };
  // the read-back of one detection pass: per-image counts, their prefix, the scored corners
  struct PassView { const int* totals = nullptr; const int* base = nullptr; const RawKp* raw = nullptr; };
  // (see replay_counts)
  struct Deferred { bool valid = false; PassView pv; std::vector<int> thr_final; };
  ~OrbWorkspace();
  void release();
  void reset_detector(int max_keypoints, int grid_res, int max_iters);
  // geometry_only: the host-side tables only (images, resize jobs, workgroup lists), no device call -- the pyramid plan's
  // host check (rgbdfe_debug_pyramid_plan_check) runs on machines without a GPU
  int prepare(int cols, int rows, bool use_grid, std::string& err, int n_frames = 1, bool geometry_only = false);
  int frames = 1;  // frames per super-frame this workspace holds (1: the single-frame paths)
  // set = 0 / 1: into that image / pyramid set (ensure_alt allocates set 1; use_set makes a set the current one, the one
  // the detection and description kernels read); -1 = the current set.  rgbdfe_detect_describe_batch uploads frame k+1
  // into the other set, from a helper thread on another stream, while frame k is being detected.  Reads only geometry
  // that is constant between two prepare() calls: safe beside a detection running on the current set.
  int upload_and_build(const uint8_t* gray, const uint8_t* mask, hipStream_t s, std::string& err, int set = -1,
                       bool defer_blur = false);
  void build_pyramids(uint8_t* pool, hipStream_t s);
  void stage_images(const uint8_t* gray, const uint8_t* mask, int set);          // CPU half (any thread)
  void stage_image_at(const uint8_t* gray, const uint8_t* mask, int stage, int k);  // super-frame: frame k of staging buffer `stage`
  int enqueue_staged_super(int nf, hipStream_t s, std::string& err, int set, int stage);
  int enqueue_staged(bool has_mask, hipStream_t s, std::string& err, int set);    // device half (the HIP thread)
  int ensure_alt(std::string& err);
  void use_set(int set);
  // called once, by the next gpu_pass, after its work is enqueued and before the host waits for it
  std::function<int()> before_wait;
  bool blur_pending = false;        // upload_and_build left the blur to the first detection pass (single-call path)
  hipEvent_t ev_readback = nullptr;
  // A detection pass = gpu_pass (FAST + NMS + Harris + angle for every corner at the cells' thresholds, one round trip)
  // + select_pass (orb.cpp computeKeyPoints' per-level selections on the host).  select_pass may ask for HIGHER thresholds
  // than the gpu_pass ran with: the corners at threshold t are exactly the corners at any floor f <= t whose FAST score
  // is >= t (see select_pass), which lets grid_detect serve two adjuster iterations from one round trip.
  int gpu_pass(const std::vector<int>& active, const std::vector<int>& thr, hipStream_t s, std::string& err);
  void select_pass(const std::vector<int>& active, const std::vector<int>& thr, std::vector<std::vector<KpOut>>& out);
  int detect_pass(const std::vector<int>& active, const std::vector<int>& thr,
                  std::vector<std::vector<KpOut>>& out, hipStream_t s, std::string& err);
  int grid_detect(std::vector<KpOut>& kps, hipStream_t s, std::string& err);
  // super-frame workspace: the frames [0, nf) of the current image set in order (cell_mask_nonzero holds nf * grid^2 flags)
  int super_detect(int nf, std::vector<std::vector<KpOut>>& kps_per_frame, hipStream_t s, std::string& err,
                   const std::vector<int>* covered_floors = nullptr, Deferred* deferred = nullptr);
  void compute_prepare(std::vector<KpOut>& kps, int frame, std::vector<int>& order, std::vector<DescKp>& dk) const;
  // software pipeline of the batch entry point: the device pass of super-frame s + 1 runs while the host replays the
  // adjuster over super-frame s -- a pass's outputs (counts + keypoints, device and pinned host side) exist twice
  int super_pass_enqueue(int nf, int set, int slot, hipStream_t s, std::string& err);
  int super_replay(int nf, int set, int slot, std::vector<std::vector<KpOut>>& kps_per_frame, hipStream_t s, std::string& err,
                   Deferred* deferred = nullptr);
  void use_slot(int slot);
  static constexpr int kSets = 3;  // image sets / pass slots of the super-frame pipeline (kSets - 1 passes ahead of the replay)
  uint8_t* d_passout_slot[kSets] = {}; uint8_t* h_passout_slot[kSets] = {};
  int* h_base_slot[kSets] = {};
  hipEvent_t ev_pass[kSets] = {};
  int slot_bound[kSets] = {};
  std::vector<int> slot_floor[kSets];
  // super-frame staging (pinned): a ring deeper than the image sets, so that the helper thread's copies of the caller's
  // pageable images run several super-frames ahead of the uploads (with kSets + 1 buffers the copy of super-frame s + 2 could
  // only start once s - 1 had been replayed and was waited for right behind that: 0 - 60 us per frame, depending on where the
  // copy threads ran)
  static constexpr int kStages = kSets + 3;
  uint8_t* himg_stage[kStages] = {};
  // optional: runs fn(0) .. fn(n - 1) on several threads and returns when all are done (the batch entry point's worker
  // pool); the replay then runs the per-cell adjuster chains and the per-frame merges through it -- pure host code
  std::function<void(int, const std::function<void(int)>&)> parallel_for;
  PassView current_pass() const { PassView v; v.totals = h_totals; v.base = h_base; v.raw = pass_raw; return v; }
  void select_cell(const PassView& pv, int c, int t, std::vector<KpOut>& out) const;   // select_pass for one cell at threshold t
  // A super-frame whose one pass covers every frame is replayed from COUNTS: the adjuster only needs how many keypoints a
  // cell would return at a threshold (replay_counts); the selections themselves -- select_cell at each cell's final
  // threshold, keepStrongest, the aggregate -- are left to whoever prepares the frame's description (select_frame, any thread).
  int replay_counts(int nf, const std::vector<int>& floors, const PassView& pv, std::vector<int>& thr_final);
  void select_frame(const PassView& pv, int frame, const int* thr_final, std::vector<KpOut>& kps) const;
  int count_cell(const PassView& pv, int c, int t, bool* capped) const;
  int replay_chains(int nf, const std::vector<int>& floors, std::vector<std::vector<KpOut>>& kps_per_frame);
  long replay_fallbacks = 0;  // super-frames whose replay needed another device pass (diagnostics)
  double super_floor_factor = 0.49;  // floor of a super-frame pass = threshold x this (two x0.7 steps)
  long super_passes = 0;             // device passes run by super_detect (diagnostics)
  // enqueue_more (optional) is called after the descriptor work has been enqueued and before the one synchronisation,
  // so that the caller's own launches on the stream ride on the same round trip
  int compute(std::vector<KpOut>& kps, std::vector<uint8_t>& desc, hipStream_t s, std::string& err,
              const std::function<int()>& enqueue_more = nullptr, std::vector<int>* order_out = nullptr);
  // the same in two halves (enqueue everything on s / wait and fill desc)
  int compute_enqueue(std::vector<KpOut>& kps, std::vector<uint8_t>& desc, hipStream_t s, std::string& err,
                      const std::function<int()>& enqueue_more = nullptr, std::vector<int>* order_out = nullptr);
  int compute_finish(std::vector<uint8_t>& desc, hipStream_t s, std::string& err);
  int cmp_n = 0;
  uint8_t* cmp_stage = nullptr;
  std::vector<DescKp> cmp_dk_big;
  // (order_out: the positions, in the input list, of the keypoints that survive compute()'s border filter, in the
  // level-grouped order of the output)

  // detector state (the reference's detector_ object, openni_listener.h:195)
  int grid = 3, adjuster_iters = 5, cell_min = 0, cell_max = 0, max_total = 0;
  bool lookahead = true;  // grid_detect: one device pass per two adjuster iterations (RGBDFE_DETECT_LOOKAHEAD=0: off)
  double thresh[64];
  std::vector<char> cell_mask_nonzero;
  // geometry
  int W = 0, H = 0, n_cells = 0, max_w = 0, max_h = 0, n_rows_total = 0, kp_cap = 0;
  std::vector<Cell> cells;
  std::vector<ImgDesc> cell_imgs, frame_imgs;
  std::vector<ResizeJob> jobs;
  std::vector<int> level_job_begin;
  float scale[8];
  int flw[8], flh[8];
  size_t pool_bytes = 0;
  bool pattern_uploaded = false;
  // device
  uint8_t* d_pool = nullptr; uint8_t* d_score = nullptr; uint8_t* d_blur = nullptr;
  ImgDesc* d_cell_imgs = nullptr; ImgDesc* d_frame_imgs = nullptr; ResizeJob* d_jobs = nullptr;
  TileUnit* d_units = nullptr;  // workgroup lists: FAST tiles | blur tiles | cell-image rows | resize tiles per level
  int units_fast_off = 0, units_fast_n = 0, units_blur_off = 0, units_blur_n = 0, units_rows_off = 0, units_rows_n = 0;
  int units_resize_off[8] = {}, units_resize_n[8] = {};
  // the fused pyramid kernel's workgroups (orb_internal.h PyrTile) and its LDS plan; RGBDFE_ORB_PYRAMID=levels: one launch
  // per level instead (orb_resize_kernel)
  PyrTile* d_pyr_tiles = nullptr; int n_pyr_tiles = 0; PyrPlan pyr_plan{}; bool fused_pyramid = true;
  int plan_pyramid(std::vector<PyrTile>& tiles, std::string& err);
  uint64_t* d_keep = nullptr;  // NMS survivors, one bit per pixel of every (cell, level) image
  int* d_row_cnt = nullptr; int* d_row_off = nullptr; int* d_img_total = nullptr;  // d_img_total and d_kps live inside d_passout
  uint8_t* d_passout = nullptr; uint8_t* h_passout = nullptr; size_t passout_hdr = 0;  // [per-image counts | keypoints]
  RawKp* d_kps = nullptr; DescKp* d_desckp = nullptr; uint8_t* d_desc = nullptr;
  float* d_kpxy = nullptr; int32_t* d_kept = nullptr; float4* d_xyz = nullptr;
  int32_t* d_n = nullptr;
  int32_t* d_n_proj = nullptr; int32_t* h_n_proj = nullptr;  // projectTo3D's count (its own: it may run beside a detection pass)
  // pinned host staging for the small per-frame transfers (thresholds, counts, keypoints, descriptors, 3-D points):
  // pageable copies of a few hundred bytes cost 10-20 us each and there are a dozen per frame
  int last_n_total = 0;  // keypoints of the latest detection pass: sizes the next pass's speculative read-back
  int raw_cap = 0;  // scored corners a pass's pinned read-back buffer holds
  int pin_cap = 0;  // keypoints the staging buffers hold (larger transfers fall back to pageable vectors)
  int* h_totals = nullptr; int* h_base = nullptr;  // h_totals and h_raw live inside h_passout
  RawKp* h_raw = nullptr; DescKp* h_desckp = nullptr; uint8_t* h_desc = nullptr;
  float* h_xyz_in = nullptr; float* h_xyz_out = nullptr; int32_t* h_n = nullptr;
  const RawKp* pass_raw = nullptr;   // the latest gpu_pass: its corners (h_raw or pass_raw_big) ...
  std::vector<RawKp> pass_raw_big;
  std::vector<RawKp> pass_raw_big_slot[kSets];   // (per pass slot: a slot's corners are read until its frames are described)
  uint8_t* pool_set[kSets] = {}; uint8_t* blur_set[kSets] = {};
  uint8_t* himg_set[2] = {nullptr, nullptr};
  size_t blur_bytes = 0;
  uint8_t* h_img = nullptr;  // gray + mask staging (2 x W x H): the caller's pageable images go through it in chunks
  // RGBDFE_DETECT_TIMING=1: host wall clock per phase of rgbdfe_detect_describe, printed when the workspace is released
  struct Timing {
    bool on = false;
    long frames = 0, passes = 0;
    double us[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // 0 prepare + mask scan, 1 upload + pyramid enqueue, 2 pass enqueue, 3 pass wait, 4 pass host work,
    // 5 adjuster + cell merge, 6 removeDepthless + retainBest, 7 compute host prep + enqueue, 8 compute wait, 9 copy-out
  } timing;
};

double orb_now_us();

}  // namespace rgbdfe
