// orb_host.hip -- host side of the per-frame feature path (rows a1-a7 of SURVEY.md section 8):
// the reference's detector object (3x3 grid of threshold-adaptive ORB detectors, features.cpp:42-60,
// feature_adjuster.cpp:185-317) and the Node constructor's detect -> removeDepthless -> retainBest ->
// compute -> projectTo3D sequence (node.cpp:139-210).  Pixel work runs in orb_kernels.hip; what stays
// here is the data-dependent control loop (re-detection with a lower threshold, top-N selection),
// which the reference also runs on the host.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "orb_host.h"

#include <chrono>
#include <functional>
#include <mutex>
#include <cstdio>
#include "orb_pattern.inc"

namespace rgbdfe {

namespace {

constexpr int kLevels = 8;           // ORB nlevels
constexpr int kDetectEdge = 15;      // ORB::create(10000, 1.2, 8, 15, ...)   feature_adjuster.cpp:94
constexpr int kComputeEdge = 31;     // ORB::create() default edgeThreshold    features.cpp:118
constexpr int kDetectFeatures = 10000;

inline int cv_round_f(float v) { return (int)lrintf(v); }

void level_geometry(int cols, int rows, int nlevels, float* scale, int* lw, int* lh) {
  const double scaleFactor = (double)1.2f;
  for (int l = 0; l < nlevels; ++l) {
    scale[l] = (float)std::pow(scaleFactor, (double)l);
    lw[l] = cv_round_f((float)cols / scale[l]);
    lh[l] = cv_round_f((float)rows / scale[l]);
  }
}

#define ORB_HIP(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return RGBDFE_ERR_HIP; } \
  } while (0)

struct KP {  // cv::KeyPoint fields in use
  float x, y, size, angle, response;
  int octave;
  float score;  // FAST score (first retainBest)
};

// KeyPointsFilter::retainBest: keep everything >= the n-th largest response; survivors keep their order
template <typename F>
void retain_best(std::vector<KP>& v, int n_points, F key) {
  if (n_points < 0 || (int)v.size() <= n_points) return;
  if (n_points == 0) { v.clear(); return; }
  static thread_local std::vector<float> r;  // scratch: 144 selections per detection pass
  r.resize(v.size());
  for (size_t i = 0; i < v.size(); ++i) r[i] = key(v[i]);
  std::nth_element(r.begin(), r.begin() + (n_points - 1), r.end(), [](float a, float b) { return a > b; });
  const float ambiguous = r[n_points - 1];
  size_t m = 0;
  for (size_t i = 0; i < v.size(); ++i)
    if (key(v[i]) >= ambiguous) v[m++] = v[i];
  v.resize(m);
}

// exactly N strongest by key (descending), ties by original order; survivors keep their order
template <typename F>
void keep_strongest(std::vector<KP>& v, int N, F key) {
  if ((int)v.size() <= N) return;
  if (N <= 0) { v.clear(); return; }
  // the N-th element of the order (key descending, position ascending) is the cut: a selection, not a sort
  static thread_local std::vector<std::pair<float, int>> r;
  r.resize(v.size());
  for (size_t i = 0; i < v.size(); ++i) r[i] = std::make_pair(key(v[i]), (int)i);
  auto before = [](const std::pair<float, int>& a, const std::pair<float, int>& b) {
    return a.first > b.first || (a.first == b.first && a.second < b.second);
  };
  std::nth_element(r.begin(), r.begin() + (N - 1), r.end(), before);
  const std::pair<float, int> cut = r[(size_t)N - 1];
  size_t m = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    const std::pair<float, int> me = std::make_pair(key(v[i]), (int)i);
    if (!before(cut, me)) v[m++] = v[i];  // me is at or before the cut
  }
  v.resize(m);
}

}  // namespace

double orb_now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

OrbWorkspace::~OrbWorkspace() { release(); }

// Allocation-time zero-fills and table uploads go through a SETUP STREAM and wait for THAT stream.  Neither may use the NULL
// stream: a NULL-stream hipMemset would need a device-wide synchronisation to be ordered before the context's non-blocking
// streams, hipDeviceSynchronize() invalidates a hipGraph capture another thread of the process may have open
// (api_batches.hip), and a synchronous hipMemcpy / hipMemcpyToSymbol is itself refused with "operation would make the legacy
// stream depend on a capturing blocking stream" while any other thread is capturing -- which the SIFT batch path now does by
// default (tests/test_gpu_sift_threads.py found it).  One stream per device for the whole process (a caller that alternates
// devices -- the multi-device handle's thread, tests with contexts on several devices -- used to get a new thread-local
// stream on every switch and leak the old one).
static hipError_t on_setup_stream(const std::function<hipError_t(hipStream_t)>& op) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> g(mu);   // (allocation-time work: first use of a workspace, a grown buffer)
  if (!streams[dev]) {
    e = hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking);
    if (e != hipSuccess) { streams[dev] = nullptr; return e; }
  }
  e = op(streams[dev]);
  return e != hipSuccess ? e : hipStreamSynchronize(streams[dev]);
}
static hipError_t zero_fill_and_wait(void* p, size_t bytes) {
  return on_setup_stream([&](hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); });
}
static hipError_t upload_and_wait(void* dst, const void* src, size_t bytes) {
  return on_setup_stream([&](hipStream_t s) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s); });
}
hipError_t orb_setup_stream_run(const std::function<hipError_t(hipStream_t)>& op) { return on_setup_stream(op); }

void OrbWorkspace::release() {
  if (timing.on && timing.frames) {
    static const char* names[10] = {"prepare+mask scan", "upload+pyramid enqueue", "pass enqueue", "pass wait",
                                    "pass host work", "adjuster+cell merge", "removeDepthless+retainBest",
                                    "compute prep+enqueue", "compute wait", "copy-out"};
    double sum = 0;
    for (double u : timing.us) sum += u;
    fprintf(stderr, "[rgbdfe detect timing] %ld frames, %.2f passes per frame, %.1f us per frame\n", timing.frames,
            (double)timing.passes / timing.frames, sum / timing.frames);
    for (int i = 0; i < 10; ++i) fprintf(stderr, "  %-28s %8.1f us\n", names[i], timing.us[i] / timing.frames);
    timing.frames = 0;
  }
  if (ev_readback) { (void)hipEventDestroy(ev_readback); ev_readback = nullptr; }
  if (d_passout_slot[0]) use_slot(0);  // the member pointers the frees below go through
  for (int i = 1; i < kSets; ++i) {
    if (d_passout_slot[i]) { (void)hipFree(d_passout_slot[i]); d_passout_slot[i] = nullptr; }
    if (h_passout_slot[i]) { (void)hipHostFree(h_passout_slot[i]); h_passout_slot[i] = nullptr; }
    if (h_base_slot[i]) { (void)hipHostFree(h_base_slot[i]); h_base_slot[i] = nullptr; }
  }
  d_passout_slot[0] = nullptr; h_passout_slot[0] = nullptr; h_base_slot[0] = nullptr;
  for (int i = 0; i < kSets; ++i) if (ev_pass[i]) { (void)hipEventDestroy(ev_pass[i]); ev_pass[i] = nullptr; }
  for (int i = 0; i < kStages; ++i) if (himg_stage[i]) { (void)hipHostFree(himg_stage[i]); himg_stage[i] = nullptr; }
  auto fr = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  // d_pool / d_blur / h_img alias one of the sets
  for (int i = 0; i < kSets; ++i) { fr(pool_set[i]); fr(blur_set[i]); }
  for (int i = 0; i < 2; ++i)
    if (himg_set[i]) { (void)hipHostFree(himg_set[i]); himg_set[i] = nullptr; }
  d_pool = nullptr; d_blur = nullptr; h_img = nullptr;
  fr(d_score); fr(d_cell_imgs); fr(d_frame_imgs); fr(d_jobs); fr(d_units); fr(d_pyr_tiles);
  fr(d_row_cnt); fr(d_row_off); fr(d_keep); fr(d_passout); fr(d_desckp); fr(d_desc); fr(d_kpxy);
  d_img_total = nullptr; d_kps = nullptr;  // live inside d_passout
  fr(d_kept); fr(d_xyz); fr(d_n); fr(d_n_proj);
  auto frh = [](auto*& p) { if (p) { (void)hipHostFree(p); p = nullptr; } };
  frh(h_passout); frh(h_base); frh(h_desckp);
  h_totals = nullptr; h_raw = nullptr;  // live inside h_passout
  frh(h_desc); frh(h_xyz_in); frh(h_xyz_out); frh(h_n); frh(h_n_proj);
  W = H = 0;
}

void OrbWorkspace::reset_detector(int max_keypoints, int grid_res, int max_iters) {
  grid = grid_res;
  adjuster_iters = max_iters;
  const int mn = max_keypoints;            // features.cpp:47
  const int mx = (int)(mn * 1.5);          // :48
  const int cells = grid * grid;
  cell_min = (int)roundf(mn / (float)cells);  // :52
  cell_max = (int)roundf(mx / (float)cells);  // :53
  max_total = mx;
  for (int i = 0; i < 64; ++i) thresh[i] = 20.0;  // DetectorAdjuster("ORB", 20), features.cpp:99
  W = H = 0;  // geometry depends on the grid
}

// (re)build the geometry for a frame size / cell layout
int OrbWorkspace::prepare(int cols, int rows, bool use_grid, std::string& err, int n_frames, bool geometry_only) {
  // n_frames > 1: a SUPER-FRAME workspace (rgbdfe_detect_describe_batch): the images of up to n_frames frames live in one
  // pool and every stage's ONE launch covers all of them -- frame f's grid cell c is "cell" f * grid^2 + c of the kernels'
  // ImgDesc / OrbCtl tables (kOrbCtlMax entries: 28 frames of a 3 x 3 grid), its pyramid levels are frame images f * 8 + l.
  const int per_frame_cells = use_grid ? grid * grid : 1;
  const int want_cells = per_frame_cells * n_frames;
  if (want_cells > kOrbCtlMax || n_frames > kProjectFramesMax) { err = "super-frame: more (frame, cell) detectors than a launch's tables hold"; return RGBDFE_ERR_CAPACITY; }
  if (!geometry_only && cols == W && rows == H && n_cells == want_cells && frames == n_frames && d_pool) return RGBDFE_OK;
  release();
  W = cols; H = rows;
  n_cells = want_cells;
  frames = n_frames;
  const int G = use_grid ? grid : 1;
  const int edge = use_grid ? 31 : 0;  // VideoGridAdaptedFeatureDetector edgeThreshold (feature_adjuster.h)
  cells.clear();
  for (int f = 0; f < n_frames; ++f)
    for (int i = 0; i < G; ++i) {
      const int rowstart = std::max((i * rows) / G - edge, 0);
      const int rowend = std::min(rows, ((i + 1) * rows) / G + edge);
      for (int j = 0; j < G; ++j) {
        const int colstart = std::max((j * cols) / G - edge, 0);
        const int colend = std::min(cols, ((j + 1) * cols) / G + edge);
        cells.push_back(Cell{colstart, rowstart, colend - colstart, rowend - rowstart});
      }
    }
  // pool layout: [frame 0 gray | frame 0 mask | frame 1 gray | ...] first (one upload), the pyramid levels behind
  size_t off = (size_t)2 * W * H * n_frames;
  cell_imgs.assign((size_t)n_cells * kLevels, ImgDesc{});
  frame_imgs.assign((size_t)kLevels * n_frames, ImgDesc{});
  jobs.clear();
  level_job_begin.assign(kLevels + 1, 0);
  size_t score_off = 0, row_off = 0, keep_words = 0;
  max_w = max_h = 0;
  // geometry first
  std::vector<int> lw((size_t)n_cells * kLevels), lh((size_t)n_cells * kLevels);
  for (int c = 0; c < n_cells; ++c) {
    float sc[kLevels];
    level_geometry(cells[c].w, cells[c].h, kLevels, sc, &lw[(size_t)c * kLevels], &lh[(size_t)c * kLevels]);
  }
  level_geometry(W, H, kLevels, scale, flw, flh);
  for (int c = 0; c < n_cells; ++c)
    for (int l = 0; l < kLevels; ++l) {
      ImgDesc& d = cell_imgs[(size_t)c * kLevels + l];
      d.w = lw[(size_t)c * kLevels + l]; d.h = lh[(size_t)c * kLevels + l];
      d.cell = c; d.level = l; d.has_mask = 1;
      const uint32_t gray_off = (uint32_t)((size_t)2 * W * H * (c / per_frame_cells));
      const uint32_t mask_off = gray_off + (uint32_t)((size_t)W * H);
      if (l == 0) {
        d.off = gray_off + (uint32_t)((size_t)cells[c].y0 * W + cells[c].x0); d.stride = W;
        d.mask_off = mask_off + (uint32_t)((size_t)cells[c].y0 * W + cells[c].x0); d.mask_stride = W;
      } else {
        d.off = (uint32_t)off; off += (size_t)d.w * d.h; d.stride = d.w;
        d.mask_off = (uint32_t)off; off += (size_t)d.w * d.h; d.mask_stride = d.w;
      }
      d.score_off = (uint32_t)score_off; score_off += (size_t)d.w * d.h;
      d.row_off = (int32_t)row_off; row_off += (size_t)d.h;
      d.keep_off = (uint32_t)keep_words; keep_words += (size_t)d.h * (size_t)((d.w + 63) / 64);
      max_w = std::max(max_w, d.w); max_h = std::max(max_h, d.h);
    }
  size_t blur_off = 0;
  for (int f = 0; f < n_frames; ++f)
    for (int l = 0; l < kLevels; ++l) {
      ImgDesc& d = frame_imgs[(size_t)f * kLevels + l];
      d.w = flw[l]; d.h = flh[l]; d.cell = 0; d.level = l; d.has_mask = 0;
      if (l == 0) { d.off = (uint32_t)((size_t)2 * W * H * f); d.stride = W; }
      else { d.off = (uint32_t)off; off += (size_t)d.w * d.h; d.stride = d.w; }
      d.score_off = (uint32_t)blur_off; blur_off += (size_t)d.w * d.h;
    }
  pool_bytes = off + 256;
  // resize jobs, level by level (level l reads level l-1)
  for (int l = 1; l < kLevels; ++l) {
    level_job_begin[l] = (int)jobs.size();
    auto add = [&](const ImgDesc& s, const ImgDesc& d, bool is_mask) {
      ResizeJob j{};
      j.src_off = is_mask ? s.mask_off : s.off; j.dst_off = is_mask ? d.mask_off : d.off;
      j.sw = s.w; j.sh = s.h; j.sstride = is_mask ? s.mask_stride : s.stride; j.dw = d.w; j.dh = d.h;
      j.is_mask = is_mask ? 1 : 0;
      j.scale_x = 1. / ((double)d.w / s.w);
      j.scale_y = 1. / ((double)d.h / s.h);
      jobs.push_back(j);
    };
    for (int c = 0; c < n_cells; ++c) {
      add(cell_imgs[(size_t)c * kLevels + l - 1], cell_imgs[(size_t)c * kLevels + l], false);
      add(cell_imgs[(size_t)c * kLevels + l - 1], cell_imgs[(size_t)c * kLevels + l], true);
    }
    for (int f = 0; f < n_frames; ++f) add(frame_imgs[(size_t)f * kLevels + l - 1], frame_imgs[(size_t)f * kLevels + l], false);
  }
  level_job_begin[kLevels] = (int)jobs.size();
  // workgroup lists of the 2-D stages (orb_internal.h TileUnit): no empty workgroups for the small images of a step
  std::vector<TileUnit> units;
  auto add_tiles = [&](int img, int w, int h, int th = 4) {
    for (int by = 0; by < (h + th - 1) / th; ++by)
      for (int bx = 0; bx < (w + 63) / 64; ++bx) units.push_back(TileUnit{(uint16_t)img, (uint16_t)bx, (uint16_t)by, 0});
  };
  units_fast_off = (int)units.size();
  for (size_t i = 0; i < cell_imgs.size(); ++i) add_tiles((int)i, cell_imgs[i].w, cell_imgs[i].h, 16);  // orb_fast_nms_kernel: 64 x 16
  units_fast_n = (int)units.size() - units_fast_off;
  units_blur_off = (int)units.size();
  for (size_t i = 0; i < frame_imgs.size(); ++i) add_tiles((int)i, frame_imgs[i].w, frame_imgs[i].h, 16);  // orb_blur_kernel: 64 x 16
  units_blur_n = (int)units.size() - units_blur_off;
  units_rows_off = (int)units.size();
  for (size_t i = 0; i < cell_imgs.size(); ++i)   // orb_emit_kernel: one wave per 64 rows of an image
    for (int y = 0; y < cell_imgs[i].h; y += 64) units.push_back(TileUnit{(uint16_t)i, 0, (uint16_t)y, 0});
  units_rows_n = (int)units.size() - units_rows_off;
  for (int l = 1; l < kLevels; ++l) {
    units_resize_off[l] = (int)units.size();
    for (int k = level_job_begin[l]; k < level_job_begin[l + 1]; ++k) add_tiles(k - level_job_begin[l], jobs[k].dw, jobs[k].dh, 16);  // orb_resize_kernel: 64 x 16
    units_resize_n[l] = (int)units.size() - units_resize_off[l];
  }
  n_rows_total = (int)row_off;
  kp_cap = (int)(score_off / 4) + 64 * n_cells * kLevels;
  if (geometry_only) return RGBDFE_OK;
  ORB_HIP(hipMalloc((void**)&d_pool, pool_bytes));
  ORB_HIP(zero_fill_and_wait(d_pool, pool_bytes));
  ORB_HIP(hipMalloc((void**)&d_score, score_off + 256));
  ORB_HIP(hipMalloc((void**)&d_blur, blur_off + 256));
  blur_bytes = blur_off + 256;
  pool_set[0] = d_pool;
  blur_set[0] = d_blur;
  ORB_HIP(hipMalloc((void**)&d_cell_imgs, sizeof(ImgDesc) * cell_imgs.size()));
  ORB_HIP(hipMalloc((void**)&d_frame_imgs, sizeof(ImgDesc) * frame_imgs.size()));
  ORB_HIP(hipMalloc((void**)&d_jobs, sizeof(ResizeJob) * jobs.size()));
  ORB_HIP(hipMalloc((void**)&d_units, sizeof(TileUnit) * units.size()));
  ORB_HIP(upload_and_wait(d_units, units.data(), sizeof(TileUnit) * units.size()));
  ORB_HIP(hipMalloc((void**)&d_row_cnt, sizeof(int) * (row_off + 16)));
  ORB_HIP(zero_fill_and_wait(d_row_cnt, sizeof(int) * (row_off + 16)));   // accumulated by atomics, re-zeroed by the row scan
  ORB_HIP(hipMalloc((void**)&d_row_off, sizeof(int) * (row_off + 16)));
  ORB_HIP(hipMalloc((void**)&d_keep, sizeof(uint64_t) * (keep_words + 16)));
  // a pass's outputs in ONE buffer -- [per-image counts | keypoints] -- so that they come back in one copy
  passout_hdr = (sizeof(int) * (cell_imgs.size() + 1) + 255) & ~(size_t)255;  // per-image counts + their sum
  ORB_HIP(hipMalloc((void**)&d_passout, passout_hdr + sizeof(RawKp) * (size_t)kp_cap));
  d_img_total = reinterpret_cast<int*>(d_passout);
  d_kps = reinterpret_cast<RawKp*>(d_passout + passout_hdr);
  ORB_HIP(hipMalloc((void**)&d_desckp, sizeof(DescKp) * (size_t)kp_cap));
  ORB_HIP(hipMalloc((void**)&d_desc, (size_t)32 * kp_cap));
  ORB_HIP(hipMalloc((void**)&d_kpxy, sizeof(float) * 3 * (size_t)kp_cap));  // x, y and the looked-up depth per keypoint
  ORB_HIP(hipMalloc((void**)&d_kept, sizeof(int32_t) * (size_t)kp_cap));
  ORB_HIP(hipMalloc((void**)&d_xyz, sizeof(float4) * (size_t)kp_cap));
  ORB_HIP(hipMalloc((void**)&d_n, sizeof(int32_t)));
  ORB_HIP(hipMalloc((void**)&d_n_proj, sizeof(int32_t) * 64));
  pin_cap = std::min(kp_cap, n_frames > 1 ? 16384 * 8 : 16384);
  raw_cap = std::min(kp_cap, n_frames > 1 ? 65536 * 8 : 16384);   // a pass's scored corners (16 bytes each)
  ORB_HIP(hipHostMalloc((void**)&h_base, sizeof(int) * cell_imgs.size(), hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_passout, passout_hdr + sizeof(RawKp) * (size_t)raw_cap, hipHostMallocDefault));
  h_totals = reinterpret_cast<int*>(h_passout);
  h_raw = reinterpret_cast<RawKp*>(h_passout + passout_hdr);
  d_passout_slot[0] = d_passout; h_passout_slot[0] = h_passout; h_base_slot[0] = h_base;
  if (n_frames > 1) {
    for (int i = 1; i < kSets; ++i) {
      ORB_HIP(hipMalloc((void**)&d_passout_slot[i], passout_hdr + sizeof(RawKp) * (size_t)kp_cap));
      ORB_HIP(hipHostMalloc((void**)&h_passout_slot[i], passout_hdr + sizeof(RawKp) * (size_t)raw_cap, hipHostMallocDefault));
      ORB_HIP(hipHostMalloc((void**)&h_base_slot[i], sizeof(int) * cell_imgs.size(), hipHostMallocDefault));
    }
    for (int i = 0; i < kSets; ++i) ORB_HIP(hipEventCreateWithFlags(&ev_pass[i], hipEventDisableTiming));
    for (int i = 0; i < kStages; ++i)
      ORB_HIP(hipHostMalloc((void**)&himg_stage[i], (size_t)2 * W * H * n_frames, hipHostMallocDefault));
  }
  ORB_HIP(hipHostMalloc((void**)&h_desckp, sizeof(DescKp) * (size_t)pin_cap, hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_desc, (size_t)32 * pin_cap, hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_xyz_in, sizeof(float) * 3 * (size_t)pin_cap, hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_xyz_out, sizeof(float) * 4 * (size_t)pin_cap, hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_n, sizeof(int32_t), hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_n_proj, sizeof(int32_t) * 64, hipHostMallocDefault));
  ORB_HIP(hipHostMalloc((void**)&h_img, (size_t)2 * W * H * n_frames, hipHostMallocDefault));
  himg_set[0] = h_img;
  ORB_HIP(upload_and_wait(d_cell_imgs, cell_imgs.data(), sizeof(ImgDesc) * cell_imgs.size()));
  ORB_HIP(upload_and_wait(d_frame_imgs, frame_imgs.data(), sizeof(ImgDesc) * frame_imgs.size()));
  ORB_HIP(upload_and_wait(d_jobs, jobs.data(), sizeof(ResizeJob) * jobs.size()));
  {
    const char* pm = getenv("RGBDFE_ORB_PYRAMID");
    fused_pyramid = !(pm && std::string(pm) == "levels");
    std::vector<PyrTile> tiles;
    if (fused_pyramid && plan_pyramid(tiles, err) != RGBDFE_OK) {
      // a shape the plan does not cover (it says why in err): one launch per level, the path of rounds 1-3
      fused_pyramid = false;
      tiles.clear();
      err.clear();
    }
    n_pyr_tiles = (int)tiles.size();
    if (n_pyr_tiles) {
      ORB_HIP(hipMalloc((void**)&d_pyr_tiles, sizeof(PyrTile) * tiles.size()));
      ORB_HIP(upload_and_wait(d_pyr_tiles, tiles.data(), sizeof(PyrTile) * tiles.size()));
    }
  }
  if (!pattern_uploaded) { orb_upload_pattern(kOrbBitPattern31); pattern_uploaded = true; }
  return RGBDFE_OK;
}

// The workgroups of orb_pyramid_kernel.  A chain (the levels of one cell's gray image, of its mask, or of a frame) is cut
// into gx x gy tiles; tile (i, j) OWNS columns [i w_l / gx, (i + 1) w_l / gx) and the rows alike of every level l, and
// COMPUTES at level l the bounding box of what it owns and of the source pixels of what it computes at level l + 1
// (resize_tap_x / resize_tap_y: the taps the kernel will read).  LDS holds two consecutive levels of a tile.
int OrbWorkspace::plan_pyramid(std::vector<PyrTile>& tiles, std::string& err) {
  const int n_chains = level_job_begin[2] - level_job_begin[1];
  PyrPlan plan{};
  for (int l = 0; l < kLevels; ++l) plan.level_job_begin[l] = level_job_begin[l];
  int buf[2] = {0, 0}, max_rw = 0, max_rh = 0;
  for (int c = 0; c < n_chains; ++c) {
    const ResizeJob& j1 = jobs[(size_t)level_job_begin[1] + c];
    const int gx = std::max(1, (j1.dw + 127) / 128), gy = std::max(1, (j1.dh + 95) / 96);
    for (int ty = 0; ty < gy; ++ty)
      for (int tx = 0; tx < gx; ++tx) {
        PyrTile t{};
        t.chain = (uint16_t)c;
        int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;   // what the level below the current one has to provide
        for (int l = kLevels - 1; l >= 1; --l) {
          const ResizeJob& j = jobs[(size_t)level_job_begin[l] + c];
          const int ox0 = tx * j.dw / gx, ox1 = (tx + 1) * j.dw / gx, oy0 = ty * j.dh / gy, oy1 = (ty + 1) * j.dh / gy;
          const bool owns = ox1 > ox0 && oy1 > oy0, needed = nx1 > nx0 && ny1 > ny0;
          int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
          if (owns && needed) { x0 = std::min(ox0, nx0); x1 = std::max(ox1, nx1); y0 = std::min(oy0, ny0); y1 = std::max(oy1, ny1); }
          else if (owns) { x0 = ox0; x1 = ox1; y0 = oy0; y1 = oy1; }
          else if (needed) { x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1; }
          if (x1 > j.dw || y1 > j.dh) { err = "pyramid plan: a region leaves its level"; return RGBDFE_ERR_INTERNAL; }
          t.nx0[l] = (uint16_t)x0; t.nx1[l] = (uint16_t)x1; t.ny0[l] = (uint16_t)y0; t.ny1[l] = (uint16_t)y1;
          t.ox0[l] = (uint16_t)(owns ? ox0 : 0); t.ox1[l] = (uint16_t)(owns ? ox1 : 0);
          t.oy0[l] = (uint16_t)(owns ? oy0 : 0); t.oy1[l] = (uint16_t)(owns ? oy1 : 0);
          if (x1 > x0 && y1 > y0) {
            // the source pixels of the region (taps are monotone in the destination coordinate)
            nx0 = resize_tap_x(x0, j.scale_x, j.sw).s0; nx1 = resize_tap_x(x1 - 1, j.scale_x, j.sw).s1 + 1;
            ny0 = resize_tap_y(y0, j.scale_y, j.sh).r0; ny1 = resize_tap_y(y1 - 1, j.scale_y, j.sh).r1 + 1;
            const int bytes = ((x1 - x0) * (y1 - y0) + 15) & ~15;
            buf[(l + 1) & 1] = std::max(buf[(l + 1) & 1], bytes);
            max_rw = std::max(max_rw, x1 - x0); max_rh = std::max(max_rh, y1 - y0);
          } else {
            nx0 = nx1 = ny0 = ny1 = 0;
          }
        }
        // level 0: the part of the uploaded image the tile's level 1 reads, from a multiple of four columns on (the kernel
        // copies it to LDS in dwords); it shares the LDS buffer of the even levels
        if (nx1 > nx0 && ny1 > ny0) {
          nx0 &= ~3;
          t.nx0[0] = (uint16_t)nx0; t.nx1[0] = (uint16_t)nx1; t.ny0[0] = (uint16_t)ny0; t.ny1[0] = (uint16_t)ny1;
          buf[1] = std::max(buf[1], (((nx1 - nx0 + 3) & ~3) * (ny1 - ny0) + 15) & ~15);
        }
        // a tile that computes nothing at level 1 computes nothing at all (the kernel stops at the first empty level)
        bool any = false, gap = false;
        for (int l = 1; l < kLevels; ++l) {
          const bool e = !(t.nx1[l] > t.nx0[l] && t.ny1[l] > t.ny0[l]);
          if (e) any = true; else if (any) gap = true;
        }
        if (gap) { err = "pyramid plan: an empty level above a non-empty one"; return RGBDFE_ERR_INTERNAL; }
        if (t.nx1[1] > t.nx0[1] && t.ny1[1] > t.ny0[1]) tiles.push_back(t);
      }
  }
  plan.buf_bytes[0] = buf[0]; plan.buf_bytes[1] = buf[1];
  plan.max_rw = max_rw; plan.max_rh = max_rh;
  if ((size_t)buf[0] + buf[1] + 8u * (size_t)(max_rw + max_rh) > 64u * 1024u) { err = "pyramid plan: LDS"; return RGBDFE_ERR_INTERNAL; }
  pyr_plan = plan;
  return RGBDFE_OK;
}

// uploads the frame and builds every pyramid (cells + whole frame) and the blurred frame levels
int OrbWorkspace::ensure_alt(std::string& err) {
  if (pool_set[1]) return RGBDFE_OK;
  for (int i = 1; i < (frames > 1 ? kSets : 2); ++i) {  // the super-frame pipeline is kSets deep, the frame pipeline two
    ORB_HIP(hipMalloc((void**)&pool_set[i], pool_bytes));
    ORB_HIP(zero_fill_and_wait(pool_set[i], pool_bytes));
    ORB_HIP(hipMalloc((void**)&blur_set[i], blur_bytes));
  }
  ORB_HIP(hipHostMalloc((void**)&himg_set[1], (size_t)2 * W * H * frames, hipHostMallocDefault));
  return RGBDFE_OK;
}

void OrbWorkspace::use_set(int set) {
  d_pool = pool_set[set];
  d_blur = blur_set[set];
  h_img = himg_set[set];
}

// one launch per pyramid level (level l reads level l - 1).  Two levels per launch -- the second one's source pixels
// recomputed from the level below -- was built and measured: 4 x 10.7 us instead of 7 x 6.1 us, the same chain; so were
// host-side coefficient tables (cv::resize's xofs / alpha / yofs / beta) instead of the per-pixel double arithmetic:
// 6.4 us per launch.  A level's launch is a chain of dependent memory round trips, not arithmetic; both were dropped.
void OrbWorkspace::build_pyramids(uint8_t* pool, hipStream_t s) {
  if (fused_pyramid) {  // round 4: every level of every chain in one launch (orb_pyramid_kernel)
    launch_orb_pyramid(pool, d_jobs, d_pyr_tiles, n_pyr_tiles, pyr_plan, s);
    return;
  }
  for (int l = 1; l < kLevels; ++l) {
    launch_orb_resize(pool, d_jobs + level_job_begin[l], d_units + units_resize_off[l], units_resize_n[l], s);
  }
}

int OrbWorkspace::upload_and_build(const uint8_t* gray, const uint8_t* mask, hipStream_t s, std::string& err, int set,
                                   bool defer_blur) {
  uint8_t* const d_pool = set < 0 ? this->d_pool : pool_set[set];  // shadow the members: the code below is set-agnostic
  uint8_t* const d_blur = set < 0 ? this->d_blur : blur_set[set];
  uint8_t* const h_img = set < 0 ? this->h_img : himg_set[set];
  // Page-locked caller images (hipHostMalloc / hipHostRegister, e.g. through rgbdfe_host_register) go to the device
  // directly; the frame's results are waited for before the call returns, so the buffers are the caller's again then.
  auto page_locked = [](const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  };
  if (page_locked(gray) && (!mask || page_locked(mask))) {
    const size_t img = (size_t)W * H;
    ORB_HIP(hipMemcpyAsync(d_pool, gray, img, hipMemcpyHostToDevice, s));
    if (mask) ORB_HIP(hipMemcpyAsync(d_pool + img, mask, img, hipMemcpyHostToDevice, s));
    else ORB_HIP(hipMemsetAsync(d_pool + img, 255, img, s));
  } else
  // Pageable images: a plain hipMemcpyAsync of 0.6 MB stalls the host ~150 us while the runtime stages it.  Own staging
  // instead: memcpy a chunk into pinned memory, start its DMA, memcpy the next chunk meanwhile.
  {
    const size_t img = (size_t)W * H;
    const size_t total = mask ? 2 * img : img;
    const size_t chunk = 320 << 10;  // measured 64 KB ... 640 KB: fewer, larger copies win (an enqueue costs as much as copying 40 KB)
    for (size_t off = 0; off < total; off += chunk) {
      const size_t n = std::min(chunk, total - off);
      // [0, img) = gray, [img, 2 img) = mask: contiguous in the staging buffer and in d_pool alike
      if (off < img) {
        const size_t n0 = std::min(n, img - off);
        memcpy(h_img + off, gray + off, n0);
        if (n0 < n) memcpy(h_img + img, mask, n - n0);
      } else {
        memcpy(h_img + off, mask + (off - img), n);
      }
      ORB_HIP(hipMemcpyAsync(d_pool + off, h_img + off, n, hipMemcpyHostToDevice, s));
    }
    if (!mask) ORB_HIP(hipMemsetAsync(d_pool + img, 255, img, s));
  }
  build_pyramids(d_pool, s);
  // The blurred levels are the descriptor kernel's input, not the detector's: on the single-call path (one stream) the blur
  // is enqueued by the first detection pass BEHIND its read-back, so that the pass does not queue up behind it.
  if (defer_blur) blur_pending = true;
  else launch_orb_blur(d_pool, d_frame_imgs, d_units + units_blur_off, units_blur_n, d_blur, s);
  ORB_HIP(hipGetLastError());
  return RGBDFE_OK;
}

// upload_and_build in two halves for the batch entry point.  Two host threads inside the HIP runtime at once serialise on
// its locks (measured: the calling thread's five launches of a detection pass took 105 us instead of 16 while the helper
// thread was issuing the next frame's copies and pyramid launches), so the helper only does the CPU half -- the copy of the
// pageable images into the set's pinned staging buffer -- and the calling thread enqueues the device half while it would
// otherwise wait for a pass.
void OrbWorkspace::stage_images(const uint8_t* gray, const uint8_t* mask, int set) {
  uint8_t* const h = himg_set[set];
  const size_t img = (size_t)W * H;
  memcpy(h, gray, img);
  if (mask) memcpy(h + img, mask, img);
}

// super-frame variants: frame k of the super-frame at [k * 2WH, ...) of the staging buffer / pool, a missing mask as 255s
void OrbWorkspace::use_slot(int slot) {
  d_passout = d_passout_slot[slot];
  d_img_total = reinterpret_cast<int*>(d_passout);
  d_kps = reinterpret_cast<RawKp*>(d_passout + passout_hdr);
  h_passout = h_passout_slot[slot];
  h_totals = reinterpret_cast<int*>(h_passout);
  h_raw = reinterpret_cast<RawKp*>(h_passout + passout_hdr);
  h_base = h_base_slot[slot];
}

void OrbWorkspace::stage_image_at(const uint8_t* gray, const uint8_t* mask, int stage, int k) {
  uint8_t* const h = himg_stage[stage] + (size_t)2 * W * H * k;
  const size_t img = (size_t)W * H;
  memcpy(h, gray, img);
  if (mask) memcpy(h + img, mask, img);
  else memset(h + img, 255, img);
}

int OrbWorkspace::enqueue_staged_super(int nf, hipStream_t s, std::string& err, int set, int stage) {
  uint8_t* const d_pool = pool_set[set];
  uint8_t* const d_blur = blur_set[set];
  ORB_HIP(hipMemcpyAsync(d_pool, himg_stage[stage], (size_t)2 * W * H * nf, hipMemcpyHostToDevice, s));
  build_pyramids(d_pool, s);
  launch_orb_blur(d_pool, d_frame_imgs, d_units + units_blur_off, units_blur_n, d_blur, s);
  ORB_HIP(hipGetLastError());
  return RGBDFE_OK;
}

int OrbWorkspace::enqueue_staged(bool has_mask, hipStream_t s, std::string& err, int set) {
  uint8_t* const d_pool = pool_set[set];
  uint8_t* const d_blur = blur_set[set];
  const size_t img = (size_t)W * H;
  ORB_HIP(hipMemcpyAsync(d_pool, himg_set[set], has_mask ? 2 * img : img, hipMemcpyHostToDevice, s));
  if (!has_mask) ORB_HIP(hipMemsetAsync(d_pool + img, 255, img, s));
  build_pyramids(d_pool, s);
  launch_orb_blur(d_pool, d_frame_imgs, d_units + units_blur_off, units_blur_n, d_blur, s);
  ORB_HIP(hipGetLastError());
  return RGBDFE_OK;
}

// The device half of a detection pass over the active cells at the given FAST thresholds: every corner that survives
// NMS, the mask and the border filters, in raster order per (cell, level) image, with its FAST score, Harris response
// and orientation -- read back in one round trip (pass_raw, h_totals, h_base).
int OrbWorkspace::gpu_pass(const std::vector<int>& active, const std::vector<int>& thr, hipStream_t s, std::string& err) {
  const double tp0 = timing.on ? orb_now_us() : 0;
  OrbCtl ctl;  // thresholds and active flags are kernel arguments: no upload in front of the pass
  for (int c = 0; c < kOrbCtlMax; ++c) {
    ctl.thr[c] = c < n_cells ? thr[c] : 0;
    ctl.active[c] = c < n_cells ? active[c] : 0;
  }
  const int n_imgs = n_cells * kLevels;
  launch_orb_fast_nms(d_pool, d_cell_imgs, n_imgs, d_units + units_fast_off, units_fast_n, ctl, d_score, kDetectEdge, d_row_cnt,
                      d_row_off, d_img_total, d_keep, s);
  // One round trip per pass: the device scans the per-image counts itself, emits and measures the keypoints, and the
  // host reads counts and keypoints back together -- `bound` of them, a guess from the previous passes; a pass with
  // more keypoints than that pays a second round trip for the rest.
  const int bound = std::min(raw_cap, std::max(2048, 2 * last_n_total));
  launch_orb_emit(d_pool, d_cell_imgs, n_imgs, d_units + units_rows_off, units_rows_n, ctl, d_score, d_keep, d_row_off,
                  d_img_total, d_kps, bound, s);
  ORB_HIP(hipGetLastError());
  ORB_HIP(hipMemcpyAsync(h_passout, d_passout, passout_hdr + sizeof(RawKp) * (size_t)bound, hipMemcpyDeviceToHost, s));  // counts + keypoints
  bool wait_event = false;
  if (blur_pending) {  // (see upload_and_build) the host waits for the read-back only, the blur runs behind it
    blur_pending = false;
    if (!ev_readback) ORB_HIP(hipEventCreateWithFlags(&ev_readback, hipEventDisableTiming));
    ORB_HIP(hipEventRecord(ev_readback, s));
    launch_orb_blur(d_pool, d_frame_imgs, d_units + units_blur_off, units_blur_n, d_blur, s);
    ORB_HIP(hipGetLastError());
    wait_event = true;
  }
  const double tp1 = timing.on ? orb_now_us() : 0;  // the caller's hook below is its own time, not the pass's
  if (before_wait) {  // the caller's own host work (the next frame's upload) rides on this pass's device time
    std::function<int()> f = std::move(before_wait);
    before_wait = nullptr;
    const int rc = f();
    if (rc != RGBDFE_OK) { (void)hipStreamSynchronize(s); err = "prefetch of the next frame failed"; return rc; }
  }
  if (wait_event) ORB_HIP(hipEventSynchronize(ev_readback));
  else ORB_HIP(hipStreamSynchronize(s));
  const double tp2 = timing.on ? orb_now_us() : 0;
  int n_total = 0;
  for (int i = 0; i < n_imgs; ++i) { h_base[i] = n_total; n_total += h_totals[i]; }
  if (n_total > kp_cap) { err = "keypoint capacity exceeded"; return RGBDFE_ERR_CAPACITY; }
  last_n_total = n_total;
  pass_raw = h_raw;
  if (n_total > bound) {
    launch_orb_measure_rest(d_pool, d_cell_imgs, d_kps, d_img_total, n_imgs, bound, n_total - bound, s);
    ORB_HIP(hipGetLastError());
    pass_raw_big.resize((size_t)n_total);
    ORB_HIP(hipMemcpyAsync(pass_raw_big.data(), d_kps, sizeof(RawKp) * (size_t)n_total, hipMemcpyDeviceToHost, s));
    ORB_HIP(hipStreamSynchronize(s));
    pass_raw = pass_raw_big.data();
  }
  if (timing.on) {
    timing.passes++;
    timing.us[2] += tp1 - tp0;
    timing.us[3] += tp2 - tp1;
    timing.us[4] += orb_now_us() - tp2;
  }
  return RGBDFE_OK;
}

// The host half: out[c] receives the cell's keypoints (level coordinates scaled to the cell image, cell-local) after
// orb.cpp computeKeyPoints' per-level selection: retainBest(2*featuresNum) by FAST score, Harris responses,
// retainBest(featuresNum).  thr[c] may be HIGHER than the threshold the latest gpu_pass ran the cell with: cv::FAST at
// threshold t keeps the pixels whose best arc has min |difference| m > t that are strict 3x3 maxima of the score
// m - 1 (non-corners count as 0).  A corner kept at t therefore has score >= t and beats every neighbour whose own score
// is < t whether that neighbour counts as a corner (floor f <= its score) or as 0; and a pixel that loses against a
// neighbour at the floor loses against the same neighbour at t when its own score is >= t (the neighbour's is larger
// still).  So { corners at t } = { corners at f with score >= t }, in the same raster order; Harris response and angle
// do not depend on the threshold.
static void per_level_caps(int* per_level) {  // nfeaturesPerLevel (orb.cpp computeKeyPoints)
  const float factor = (float)(1.0 / (double)1.2f);
  float nd = kDetectFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
  int sum = 0;
  for (int l = 0; l < kLevels - 1; ++l) { per_level[l] = cv_round_f(nd); sum += per_level[l]; nd *= factor; }
  per_level[kLevels - 1] = std::max(kDetectFeatures - sum, 0);
}

// How many keypoints select_cell(c, t) would return, without building them: per level the corners with score >= t -- exact as
// long as no level reaches its retainBest cap (n <= nfeaturesPerLevel: neither retainBest(2n) nor retainBest(n) cuts, ties
// included); *capped is set otherwise and the caller runs the selection itself.
int OrbWorkspace::count_cell(const PassView& pv, int c, int thr_c, bool* capped) const {
  int per_level[kLevels];
  per_level_caps(per_level);
  const int t = std::min(std::max(thr_c, 0), 255);
  int found = 0;
  for (int l = 0; l < kLevels; ++l) {
    const int img = c * kLevels + l;
    const RawKp* r = pv.raw + pv.base[img];
    int n = 0;
    for (int k = 0; k < pv.totals[img]; ++k) n += (int)r[k].score >= t ? 1 : 0;
    if (n > per_level[l]) { *capped = true; return 0; }
    found += n;
  }
  return found;
}

void OrbWorkspace::select_cell(const PassView& pv, int c, int thr_c, std::vector<KpOut>& out) const {
  // nfeaturesPerLevel (orb.cpp computeKeyPoints)
  int per_level[kLevels];
  {
    const float factor = (float)(1.0 / (double)1.2f);
    float nd = kDetectFeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
    int sum = 0;
    for (int l = 0; l < kLevels - 1; ++l) { per_level[l] = cv_round_f(nd); sum += per_level[l]; nd *= factor; }
    per_level[kLevels - 1] = std::max(kDetectFeatures - sum, 0);
  }
  const int* totals = pv.totals;
  const int* base = pv.base;
  const RawKp* raw = pv.raw;
  out.clear();
  const int t = std::min(std::max(thr_c, 0), 255);  // the kernel's clamp
  float sc[kLevels]; int lw[kLevels], lh[kLevels];
  level_geometry(cells[c].w, cells[c].h, kLevels, sc, lw, lh);
  for (int l = 0; l < kLevels; ++l) {
    const int img = c * kLevels + l;
    static thread_local std::vector<KP> v;  // scratch: 72 (cell, level) images per pass
    v.clear();
    v.reserve((size_t)totals[img]);
    for (int k = 0; k < totals[img]; ++k) {
      const RawKp& r = raw[(size_t)base[img] + k];
      if ((int)r.score < t) continue;
      v.push_back(KP{(float)r.x, (float)r.y, 31 * sc[l], r.angle, r.harris, l, (float)r.score});
    }
    retain_best(v, 2 * per_level[l], [](const KP& k) { return k.score; });
    retain_best(v, per_level[l], [](const KP& k) { return k.response; });
    for (const KP& k : v) out.push_back(KpOut{k.x * sc[l], k.y * sc[l], k.size, k.angle, k.response, l});
  }
}

void OrbWorkspace::select_pass(const std::vector<int>& active, const std::vector<int>& thr,
                               std::vector<std::vector<KpOut>>& out) {
  const double tp0 = timing.on ? orb_now_us() : 0;
  const PassView pv = current_pass();
  for (int c = 0; c < n_cells; ++c)
    if (active[c]) select_cell(pv, c, thr[c], out[c]);
  if (timing.on) timing.us[4] += orb_now_us() - tp0;
}

// One detection pass over the active cells with their current thresholds (cv::ORB::detect on one image).
int OrbWorkspace::detect_pass(const std::vector<int>& active, const std::vector<int>& thr,
                              std::vector<std::vector<KpOut>>& out, hipStream_t s, std::string& err) {
  const int rc = gpu_pass(active, thr, s, err);
  if (rc != RGBDFE_OK) return rc;
  select_pass(active, thr, out);
  return RGBDFE_OK;
}

// VideoGridAdaptedFeatureDetector::detect over the whole frame (feature_adjuster.cpp:286-317)
int OrbWorkspace::grid_detect(std::vector<KpOut>& kps, hipStream_t s, std::string& err) {
  std::vector<std::vector<KpOut>> cellkp((size_t)n_cells);
  std::vector<int> active((size_t)n_cells, 1), iter_left((size_t)n_cells, adjuster_iters), thr((size_t)n_cells);
  std::vector<char> checked((size_t)n_cells, 0);
  // host copy of the mask is needed only for hasNonZero(): done lazily by the caller through mask_nonzero
  // A device pass runs every active cell at the threshold its NEXT adjuster iteration would use (x0.7, below) and
  // select_pass filters by score: the second iteration of a cell that finds too few keypoints -- the common case on
  // frames the thresholds have not settled on -- needs no second round trip.  The corners are those of separate passes.
  std::vector<int> floor_thr((size_t)n_cells, 0);
  std::vector<char> covered((size_t)n_cells, 0);
  bool any = true;
  while (any) {
    bool need_gpu = false;
    for (int c = 0; c < n_cells; ++c) {
      thr[c] = (int)thresh[c];  // static_cast<int>(thresh_)
      if (active[c] && (!covered[c] || thr[c] < floor_thr[c])) need_gpu = true;
    }
    if (need_gpu) {
      for (int c = 0; c < n_cells; ++c) {
        covered[c] = (char)active[c];
        double next = thresh[c] * 0.7;  // tooFew (:131-136)
        if (next < 2) next = 2;
        floor_thr[c] = lookahead ? std::min(thr[c], (int)next) : thr[c];
      }
      const int rc = gpu_pass(active, floor_thr, s, err);
      if (rc != RGBDFE_OK) return rc;
    }
    select_pass(active, thr, cellkp);
    any = false;
    for (int c = 0; c < n_cells; ++c) {
      if (!active[c]) continue;
      const int found = (int)cellkp[c].size();
      bool again = false;
      // VideoDynamicAdaptedFeatureDetector::detect (feature_adjuster.cpp:185-224)
      if (found < cell_min) {
        thresh[c] *= 0.7;                       // tooFew (:131-136)
        if (thresh[c] < 2) thresh[c] = 2;
        bool brk = false;
        if (found == 0 && !checked[c]) {
          checked[c] = 1;
          if (!cell_mask_nonzero[c]) brk = true;  // hasNonZero(mask) (:205-209)
        }
        if (!brk) {
          iter_left[c]--;
          again = iter_left[c] > 0 && (thresh[c] > 2 && thresh[c] < 10000);  // good() (:147-150)
        }
      } else if (found > cell_max) {
        thresh[c] *= 1.3;                       // tooMany (:138-143)
        if (thresh[c] > 10000) thresh[c] = 10000;
      }
      active[c] = again ? 1 : 0;
      any |= again;
    }
  }
  const int maxPerCell = max_total / (n_cells);  // :292
  kps.clear();
  for (int c = 0; c < n_cells; ++c) {
    std::vector<KP> v;
    v.reserve(cellkp[c].size());
    for (const KpOut& k : cellkp[c]) v.push_back(KP{k.x, k.y, k.size, k.angle, k.response, k.octave, 0.f});
    keep_strongest(v, maxPerCell, [](const KP& k) { return std::fabs(k.response); });  // :247-255
    for (const KP& k : v)  // aggregateKeypointsPerGridCell (:259-282)
      kps.push_back(KpOut{k.x + cells[c].x0, k.y + cells[c].y0, k.size, k.angle, k.response, k.octave});
  }
  return RGBDFE_OK;
}

// The device half of a super-frame's detection pass, enqueued only (no wait): the frames [0, nf) of image set `set` at floor
// thresholds derived from the detector's thresholds OF THIS MOMENT -- the replay of the previous super-frame may still
// move them; a cell that ends up below its floor is re-run by super_replay.  Outputs go to pass slot `slot`.
int OrbWorkspace::super_pass_enqueue(int nf, int set, int slot, hipStream_t s, std::string& err) {
  const int pc = grid * grid;
  OrbCtl ctl;
  std::vector<int>& fl = slot_floor[slot];
  fl.assign((size_t)n_cells, 0);
  for (int c = 0; c < kOrbCtlMax; ++c) {
    ctl.thr[c] = 0; ctl.active[c] = 0;
    if (c >= n_cells || c / pc >= nf) continue;
    double next = thresh[c % pc] * super_floor_factor;
    if (next < 2) next = 2;
    fl[(size_t)c] = std::min((int)thresh[c % pc], (int)next);
    ctl.thr[c] = fl[(size_t)c];
    ctl.active[c] = 1;
  }
  uint8_t* const pool = pool_set[set];
  uint8_t* const dpo = d_passout_slot[slot];
  int* const img_total = reinterpret_cast<int*>(dpo);
  RawKp* const kps = reinterpret_cast<RawKp*>(dpo + passout_hdr);
  const int n_imgs = n_cells * kLevels;
  launch_orb_fast_nms(pool, d_cell_imgs, n_imgs, d_units + units_fast_off, units_fast_n, ctl, d_score, kDetectEdge, d_row_cnt,
                      d_row_off, img_total, d_keep, s);
  const int bound = std::min(raw_cap, std::max(2048 * frames, 2 * last_n_total));
  slot_bound[slot] = bound;
  // the read-back: written by the measure kernel itself into the slot's page-locked buffer (RGBDFE_DETECT_HOSTWRITE=0: a copy
  // of the counts and `bound` records behind the kernels, the form of rounds 2-5)
  const char* const hw_env = getenv("RGBDFE_DETECT_HOSTWRITE");   // (read per pass: the tests switch it inside one process)
  const bool host_write = !(hw_env && atoi(hw_env) == 0);
  uint8_t* const hpo = h_passout_slot[slot];
  launch_orb_emit(pool, d_cell_imgs, n_imgs, d_units + units_rows_off, units_rows_n, ctl, d_score, d_keep, d_row_off,
                  img_total, kps, bound, s, host_write ? reinterpret_cast<int*>(hpo) : nullptr,
                  host_write ? reinterpret_cast<RawKp*>(hpo + passout_hdr) : nullptr);
  ORB_HIP(hipGetLastError());
  if (!host_write)
    ORB_HIP(hipMemcpyAsync(hpo, dpo, passout_hdr + sizeof(RawKp) * (size_t)bound, hipMemcpyDeviceToHost, s));
  ORB_HIP(hipEventRecord(ev_pass[slot], s));
  super_passes++;
  return RGBDFE_OK;
}

// The host half: waits for the slot's pass, then replays the adjuster frame by frame exactly as super_detect does (see
// there); re-passes (a threshold fell below its floor) run synchronously behind whatever the stream already holds.
int OrbWorkspace::super_replay(int nf, int set, int slot, std::vector<std::vector<KpOut>>& kps_per_frame, hipStream_t s,
                               std::string& err, Deferred* deferred) {
  ORB_HIP(hipEventSynchronize(ev_pass[slot]));
  use_slot(slot);
  use_set(set);
  const int n_imgs = n_cells * kLevels;
  int n_total = 0;
  for (int i = 0; i < n_imgs; ++i) { h_base[i] = n_total; n_total += h_totals[i]; }
  if (n_total > kp_cap) { err = "keypoint capacity exceeded"; return RGBDFE_ERR_CAPACITY; }
  last_n_total = n_total;
  pass_raw = h_raw;
  if (n_total > slot_bound[slot]) {  // more corners than the speculative read-back: the rest in a second trip
    launch_orb_measure_rest(d_pool, d_cell_imgs, d_kps, d_img_total, n_imgs, slot_bound[slot], n_total - slot_bound[slot], s);
    ORB_HIP(hipGetLastError());
    std::vector<RawKp>& big = pass_raw_big_slot[slot];
    big.resize((size_t)n_total);
    ORB_HIP(hipMemcpyAsync(big.data(), d_kps, sizeof(RawKp) * (size_t)n_total, hipMemcpyDeviceToHost, s));
    ORB_HIP(hipStreamSynchronize(s));
    pass_raw = big.data();
  }
  return super_detect(nf, kps_per_frame, s, err, &slot_floor[slot], deferred);
}

// VideoGridAdaptedFeatureDetector::detect for the frames [0, nf) of a super-frame, IN ORDER: frame f + 1 starts from the
// per-cell thresholds frame f leaves behind (feature_adjuster.cpp:185-224), exactly as nf calls of grid_detect would.
// What makes one device pass serve all of them: the corners at threshold t are the corners at any floor f <= t whose
// score is >= t (select_pass), so the pass runs every (frame, cell) at a floor below the threshold it is expected to
// end up with, and the adjuster is replayed on the host over the scored corners, frame by frame.  A cell whose threshold
// drops below its floor (rare: two x0.7 steps inside one super-frame) triggers another pass over the frames from there
// on, with floors taken from the thresholds of that moment -- results do not depend on the floors.
// The replay of super_detect when one pass at `floors` covers every frame -- the common case -- as independent work items:
// the adjuster of a grid cell only ever looks at its own cell (feature_adjuster.cpp:185-224 runs inside one cell's
// detector object), so the grid^2 chains "cell c9 over frames 0 .. nf - 1" do not interact, and neither do the nf per-frame
// merges (keepStrongest per cell + aggregate, :247-282) that follow.  Both run through parallel_for.  Returns 0 when a chain
// met a threshold below its floor: nothing has been committed then, and super_detect's sequential loop (which owns the
// re-pass logic) starts from the same state.  Same selections in the same order as that loop: identical keypoints.
int OrbWorkspace::replay_chains(int nf, const std::vector<int>& floors, std::vector<std::vector<KpOut>>& kps_per_frame) {
  const int pc = grid * grid;
  std::vector<std::vector<KpOut>> cellkp((size_t)nf * pc);
  std::vector<double> th_end((size_t)pc);
  std::vector<char> failed((size_t)pc, 0);
  const PassView pv = current_pass();
  parallel_for(pc, [&](int c9) {
    double th = thresh[c9];
    for (int f = 0; f < nf && !failed[c9]; ++f) {
      const int c = f * pc + c9;
      int iter_left = adjuster_iters;
      bool checked = false, active = true;
      while (active) {
        const int t = (int)th;  // static_cast<int>(thresh_)
        if (t < floors[(size_t)c]) { failed[c9] = 1; break; }
        select_cell(pv, c, t, cellkp[(size_t)c]);
        const int found = (int)cellkp[(size_t)c].size();
        bool again = false;
        if (found < cell_min) {
          th *= 0.7;                                 // tooFew (:131-136)
          if (th < 2) th = 2;
          bool brk = false;
          if (found == 0 && !checked) {
            checked = true;
            if (!cell_mask_nonzero[(size_t)c]) brk = true;  // hasNonZero(mask) (:205-209)
          }
          if (!brk) {
            iter_left--;
            again = iter_left > 0 && (th > 2 && th < 10000);  // good() (:147-150)
          }
        } else if (found > cell_max) {
          th *= 1.3;                                 // tooMany (:138-143)
          if (th > 10000) th = 10000;
        }
        active = again;
      }
    }
    th_end[(size_t)c9] = th;
  });
  for (int c9 = 0; c9 < pc; ++c9)
    if (failed[c9]) return 0;
  for (int c9 = 0; c9 < pc; ++c9) thresh[c9] = th_end[(size_t)c9];
  kps_per_frame.assign((size_t)nf, std::vector<KpOut>());
  const int maxPerCell = max_total / pc;  // :292
  parallel_for(nf, [&](int f) {
    std::vector<KpOut>& kps = kps_per_frame[(size_t)f];
    std::vector<KP> v;
    for (int c9 = 0; c9 < pc; ++c9) {
      const int c = f * pc + c9;
      v.clear();
      v.reserve(cellkp[(size_t)c].size());
      for (const KpOut& k : cellkp[(size_t)c]) v.push_back(KP{k.x, k.y, k.size, k.angle, k.response, k.octave, 0.f});
      keep_strongest(v, maxPerCell, [](const KP& k) { return std::fabs(k.response); });  // :247-255
      for (const KP& k : v)  // aggregateKeypointsPerGridCell (:259-282)
        kps.push_back(KpOut{k.x + cells[c].x0, k.y + cells[c].y0, k.size, k.angle, k.response, k.octave});
    }
  });
  return 1;
}

// The adjuster over the frames of a covered super-frame from counts alone (see orb_host.h): per grid cell the chain
// "threshold -> keypoints found -> too few: x0.7 and again / too many: x1.3 for the next frame" (feature_adjuster.cpp:185-224)
// needs `found` only, and found = the corners of the cell's 8 levels whose score is >= the threshold (count_cell; a level at
// its retainBest cap falls back to the real selection).  thr_final[c] = the threshold of the LAST detection of (frame, cell)
// c, the one whose keypoints the reference keeps.  Returns 0 -- nothing committed -- when a threshold falls below its floor.
int OrbWorkspace::replay_counts(int nf, const std::vector<int>& floors, const PassView& pv, std::vector<int>& thr_final) {
  const int pc = grid * grid;
  thr_final.assign((size_t)n_cells, 0);
  std::vector<double> th_end((size_t)pc);
  std::vector<KpOut> scratch;
  for (int c9 = 0; c9 < pc; ++c9) {
    double th = thresh[c9];
    for (int f = 0; f < nf; ++f) {
      const int c = f * pc + c9;
      int iter_left = adjuster_iters;
      bool checked = false, active = true;
      while (active) {
        const int t = (int)th;  // static_cast<int>(thresh_)
        if (t < floors[(size_t)c]) return 0;
        thr_final[(size_t)c] = t;
        bool capped = false;
        int found = count_cell(pv, c, t, &capped);
        if (capped) { select_cell(pv, c, t, scratch); found = (int)scratch.size(); }
        bool again = false;
        if (found < cell_min) {
          th *= 0.7;                                 // tooFew (:131-136)
          if (th < 2) th = 2;
          bool brk = false;
          if (found == 0 && !checked) {
            checked = true;
            if (!cell_mask_nonzero[(size_t)c]) brk = true;  // hasNonZero(mask) (:205-209)
          }
          if (!brk) {
            iter_left--;
            again = iter_left > 0 && (th > 2 && th < 10000);  // good() (:147-150)
          }
        } else if (found > cell_max) {
          th *= 1.3;                                 // tooMany (:138-143)
          if (th > 10000) th = 10000;
        }
        active = again;
      }
    }
    th_end[(size_t)c9] = th;
  }
  for (int c9 = 0; c9 < pc; ++c9) thresh[c9] = th_end[(size_t)c9];
  return 1;
}

// VideoGridAdaptedFeatureDetector::detect's output for one frame of a super-frame, given each cell's final threshold:
// the cell's keypoints at that threshold (select_cell), keepStrongest(maxPerCell) (:247-255), cell offsets added and the cells
// appended in order (aggregateKeypointsPerGridCell, :259-282).  Reads the pass view only: runs on any thread.
void OrbWorkspace::select_frame(const PassView& pv, int frame, const int* thr_final, std::vector<KpOut>& kps) const {
  const int pc = grid * grid;
  const int maxPerCell = max_total / pc;  // :292
  kps.clear();
  std::vector<KpOut> cell;
  std::vector<KP> v;
  for (int c9 = 0; c9 < pc; ++c9) {
    const int c = frame * pc + c9;
    select_cell(pv, c, thr_final[c], cell);
    v.clear();
    v.reserve(cell.size());
    for (const KpOut& k : cell) v.push_back(KP{k.x, k.y, k.size, k.angle, k.response, k.octave, 0.f});
    keep_strongest(v, maxPerCell, [](const KP& k) { return std::fabs(k.response); });
    for (const KP& k : v) kps.push_back(KpOut{k.x + cells[c].x0, k.y + cells[c].y0, k.size, k.angle, k.response, k.octave});
  }
}

int OrbWorkspace::super_detect(int nf, std::vector<std::vector<KpOut>>& kps_per_frame, hipStream_t s, std::string& err,
                               const std::vector<int>* covered_floors, Deferred* deferred) {
  const int pc = grid * grid;
  if (deferred) deferred->valid = false;
  if (covered_floors && deferred) {
    const double tp0 = timing.on ? orb_now_us() : 0;
    deferred->pv = current_pass();
    const int done = replay_counts(nf, *covered_floors, deferred->pv, deferred->thr_final);
    if (timing.on) timing.us[4] += orb_now_us() - tp0;
    if (done) {
      deferred->valid = true;
      kps_per_frame.assign((size_t)nf, std::vector<KpOut>());
      return RGBDFE_OK;
    }
    replay_fallbacks++;
  } else if (covered_floors && parallel_for) {
    const double tp0 = timing.on ? orb_now_us() : 0;
    const int done = replay_chains(nf, *covered_floors, kps_per_frame);
    if (timing.on) timing.us[4] += orb_now_us() - tp0;
    if (done) return RGBDFE_OK;
    replay_fallbacks++;
  }
  std::vector<std::vector<KpOut>> cellkp((size_t)n_cells);
  std::vector<int> act((size_t)n_cells, 0), thr((size_t)n_cells, 0), floor_thr((size_t)n_cells, 0);
  std::vector<char> covered((size_t)n_cells, 0);
  if (covered_floors)  // a pass over all nf frames at these floors has already been read back (super_replay)
    for (int c = 0; c < n_cells && c / pc < nf; ++c) { covered[c] = 1; floor_thr[c] = (*covered_floors)[(size_t)c]; }
  kps_per_frame.assign((size_t)nf, std::vector<KpOut>());
  auto run_pass = [&](int f_from) -> int {
    std::vector<int> pa((size_t)n_cells, 0);
    for (int c = 0; c < n_cells; ++c) {
      const int f = c / pc;
      covered[c] = 0;
      if (f < f_from || f >= nf) continue;
      pa[c] = 1;
      covered[c] = 1;
      double next = thresh[c % pc] * super_floor_factor;
      if (next < 2) next = 2;
      floor_thr[c] = std::min((int)thresh[c % pc], (int)next);
    }
    super_passes++;
    return gpu_pass(pa, floor_thr, s, err);
  };
  for (int f = 0; f < nf; ++f) {
    std::vector<int> iter_left((size_t)pc, adjuster_iters);
    std::vector<char> checked((size_t)pc, 0), active((size_t)pc, 1);
    bool any = true;
    while (any) {
      bool need_gpu = false;
      std::fill(act.begin(), act.end(), 0);
      for (int c9 = 0; c9 < pc; ++c9) {
        const int c = f * pc + c9;
        thr[c] = (int)thresh[c9];  // static_cast<int>(thresh_)
        act[c] = active[c9];
        if (active[c9] && (!covered[c] || thr[c] < floor_thr[c])) need_gpu = true;
      }
      if (need_gpu) {
        const int rc = run_pass(f);
        if (rc != RGBDFE_OK) return rc;
      }
      select_pass(act, thr, cellkp);
      any = false;
      for (int c9 = 0; c9 < pc; ++c9) {
        if (!active[c9]) continue;
        const int c = f * pc + c9;
        const int found = (int)cellkp[c].size();
        bool again = false;
        // VideoDynamicAdaptedFeatureDetector::detect (feature_adjuster.cpp:185-224)
        if (found < cell_min) {
          thresh[c9] *= 0.7;                       // tooFew (:131-136)
          if (thresh[c9] < 2) thresh[c9] = 2;
          bool brk = false;
          if (found == 0 && !checked[c9]) {
            checked[c9] = 1;
            if (!cell_mask_nonzero[c]) brk = true;  // hasNonZero(mask) (:205-209)
          }
          if (!brk) {
            iter_left[c9]--;
            again = iter_left[c9] > 0 && (thresh[c9] > 2 && thresh[c9] < 10000);  // good() (:147-150)
          }
        } else if (found > cell_max) {
          thresh[c9] *= 1.3;                       // tooMany (:138-143)
          if (thresh[c9] > 10000) thresh[c9] = 10000;
        }
        active[c9] = again ? 1 : 0;
        any |= again;
      }
    }
    const int maxPerCell = max_total / pc;  // :292
    std::vector<KpOut>& kps = kps_per_frame[(size_t)f];
    for (int c9 = 0; c9 < pc; ++c9) {
      const int c = f * pc + c9;
      std::vector<KP> v;
      v.reserve(cellkp[c].size());
      for (const KpOut& k : cellkp[c]) v.push_back(KP{k.x, k.y, k.size, k.angle, k.response, k.octave, 0.f});
      keep_strongest(v, maxPerCell, [](const KP& k) { return std::fabs(k.response); });  // :247-255
      for (const KP& k : v)  // aggregateKeypointsPerGridCell (:259-282)
        kps.push_back(KpOut{k.x + cells[c].x0, k.y + cells[c].y0, k.size, k.angle, k.response, k.octave});
    }
  }
  return RGBDFE_OK;
}

// The CPU half of cv::ORB::compute for one frame of a super-frame (see compute_enqueue): border filter, regroup by level,
// descriptor records with the frame's pyramid images (frame_imgs[frame * 8 + octave]).  Pure host code: runs on any thread.
void OrbWorkspace::compute_prepare(std::vector<KpOut>& kps, int frame, std::vector<int>& order, std::vector<DescKp>& dk) const {
  order.clear();
  order.reserve(kps.size());
  auto inside = [&](const KpOut& k) {
    return k.x >= kComputeEdge && k.x < W - kComputeEdge && k.y >= kComputeEdge && k.y < H - kComputeEdge;
  };
  for (int l = 0; l < kLevels; ++l)
    for (size_t i = 0; i < kps.size(); ++i) {
      const KpOut& k = kps[i];
      if (k.octave == l && inside(k)) order.push_back((int)i);
    }
  {
    std::vector<KpOut> t;
    t.reserve(order.size());
    for (int i : order) t.push_back(kps[(size_t)i]);
    kps.swap(t);
  }
  dk.resize(kps.size());
  for (size_t j = 0; j < kps.size(); ++j) {
    const KpOut& k = kps[j];
    const float sc = 1.f / scale[k.octave];
    float angle = k.angle;
    angle *= (float)(M_PI / 180.f);
    dk[j].cos_a = (float)std::cos((double)angle);
    dk[j].sin_a = (float)std::sin((double)angle);
    dk[j].cx = cv_round_f(k.x * sc);
    dk[j].cy = cv_round_f(k.y * sc);
    dk[j].level = frame * kLevels + k.octave;
  }
}

// cv::ORB::create()->compute (features.cpp:117-119): border filter, regroup by level, rBRIEF
// compute() in two halves, so that a caller can do other work between the enqueue and the wait (the batch entry point
// overlaps a frame's description with the next frame's detection): compute_enqueue filters and regroups the keypoints,
// enqueues the descriptor kernel and the read-back on `s` (plus the caller's enqueue_more); compute_finish waits and
// fills `desc` (sized by compute_enqueue).
int OrbWorkspace::compute_enqueue(std::vector<KpOut>& kps, std::vector<uint8_t>& desc, hipStream_t s, std::string& err,
                                  const std::function<int()>& enqueue_more, std::vector<int>* order_out) {
  // KeyPointsFilter::runByImageBorder(keypoints, image.size(), 31), then the stable regroup by level (orb.cpp:
  // !sortedByLevel branch); `order` = the surviving input positions in output order
  if (blur_pending) {  // no detection pass took it along (cannot happen on the paths that defer it)
    blur_pending = false;
    launch_orb_blur(d_pool, d_frame_imgs, d_units + units_blur_off, units_blur_n, d_blur, s);
  }
  const double tc0 = timing.on ? orb_now_us() : 0;
  std::vector<int> order;
  order.reserve(kps.size());
  int nlevels = 1;
  auto inside = [&](const KpOut& k) {
    return k.x >= kComputeEdge && k.x < W - kComputeEdge && k.y >= kComputeEdge && k.y < H - kComputeEdge;
  };
  for (const KpOut& k : kps)
    if (inside(k)) nlevels = std::max(nlevels, std::max(k.octave, 0) + 1);
  if (nlevels > kLevels) { err = "keypoint octave beyond the 8-level pyramid"; return RGBDFE_ERR_INVALID_ARG; }
  for (int l = 0; l < nlevels; ++l)
    for (size_t i = 0; i < kps.size(); ++i) {
      const KpOut& k = kps[i];
      if (k.octave == l && inside(k)) order.push_back((int)i);
    }
  {
    std::vector<KpOut> t;
    t.reserve(order.size());
    for (int i : order) t.push_back(kps[(size_t)i]);
    kps.swap(t);
  }
  if (order_out) *order_out = order;
  const int n = (int)kps.size();
  desc.assign((size_t)n * 32, 0);
  cmp_n = n;
  cmp_stage = nullptr;
  if (n == 0) return RGBDFE_OK;
  if (n > kp_cap) { err = "keypoint capacity exceeded"; return RGBDFE_ERR_CAPACITY; }
  DescKp* dk = h_desckp;
  uint8_t* desc_stage = h_desc;
  if (n > pin_cap) { cmp_dk_big.resize((size_t)n); dk = cmp_dk_big.data(); desc_stage = desc.data(); }
  cmp_stage = desc_stage;
  for (int j = 0; j < n; ++j) {
    const KpOut& k = kps[j];
    const float sc = 1.f / scale[k.octave];
    float angle = k.angle;
    angle *= (float)(M_PI / 180.f);
    dk[j].cos_a = (float)std::cos((double)angle);
    dk[j].sin_a = (float)std::sin((double)angle);
    dk[j].cx = cv_round_f(k.x * sc);
    dk[j].cy = cv_round_f(k.y * sc);
    dk[j].level = k.octave;
  }
  ORB_HIP(hipMemcpyAsync(d_desckp, dk, sizeof(DescKp) * (size_t)n, hipMemcpyHostToDevice, s));
  launch_orb_brief(d_pool, d_blur, d_frame_imgs, d_desckp, n, d_desc, s);
  ORB_HIP(hipGetLastError());
  ORB_HIP(hipMemcpyAsync(desc_stage, d_desc, (size_t)n * 32, hipMemcpyDeviceToHost, s));
  if (enqueue_more) {
    const int rc = enqueue_more();
    if (rc != RGBDFE_OK) { (void)hipStreamSynchronize(s); err = "enqueue after compute failed"; return rc; }
  }
  if (timing.on) timing.us[7] += orb_now_us() - tc0;
  return RGBDFE_OK;
}

int OrbWorkspace::compute_finish(std::vector<uint8_t>& desc, hipStream_t s, std::string& err) {
  if (cmp_n == 0) return RGBDFE_OK;
  const double tc1 = timing.on ? orb_now_us() : 0;
  ORB_HIP(hipStreamSynchronize(s));
  if (timing.on) timing.us[8] += orb_now_us() - tc1;
  if (cmp_stage && cmp_stage != desc.data()) memcpy(desc.data(), cmp_stage, (size_t)cmp_n * 32);
  return RGBDFE_OK;
}

int OrbWorkspace::compute(std::vector<KpOut>& kps, std::vector<uint8_t>& desc, hipStream_t s, std::string& err,
                          const std::function<int()>& enqueue_more, std::vector<int>* order_out) {
  const int rc = compute_enqueue(kps, desc, s, err, enqueue_more, order_out);
  if (rc != RGBDFE_OK) return rc;
  return compute_finish(desc, s, err);
}

}  // namespace rgbdfe

// ------------------------------------------------------------------------------------------------
// Host check of the fused pyramid kernel's plan (tests/test_pyramid_plan.py, no GPU): the geometry and the plan of a workspace
// of the given shape, a pool filled with pseudo-random level-0 images, then (A) one resize per level and image with
// resize_tap_x / resize_tap_y and (B) an emulation of orb_pyramid_kernel tile by tile -- the same regions, the same 16-bit
// table entries, the same LDS offsets, every LDS index checked against the buffer sizes of the plan.  Returns the number of
// pool bytes that differ (0 = every pixel of every level was written, with the value the per-level resize gives), or a
// negative number: -1 geometry, -2 plan, -3 an LDS index left its buffer.
// ------------------------------------------------------------------------------------------------
// runner (optional): called for the fused side instead of the built-in emulation -- tests/test_emu_orb_kernels.py passes the
// product's own kernel source compiled for the host (tests/emu/), so the KERNEL is checked against the per-level resize,
// not a restatement of it.
typedef void (*rgbdfe_pyramid_runner)(uint8_t* pool, const rgbdfe::ResizeJob* jobs, const rgbdfe::PyrTile* tiles, int n_tiles,
                                      const rgbdfe::PyrPlan* plan);
extern "C" int rgbdfe_debug_pyramid_plan_check2(int cols, int rows, int use_grid, int n_frames, unsigned seed, int* n_tiles_out,
                                                int* lds_bytes_out, rgbdfe_pyramid_runner runner);
extern "C" int rgbdfe_debug_pyramid_plan_check(int cols, int rows, int use_grid, int n_frames, unsigned seed, int* n_tiles_out,
                                               int* lds_bytes_out) {
  return rgbdfe_debug_pyramid_plan_check2(cols, rows, use_grid, n_frames, seed, n_tiles_out, lds_bytes_out, nullptr);
}
extern "C" int rgbdfe_debug_pyramid_plan_check2(int cols, int rows, int use_grid, int n_frames, unsigned seed, int* n_tiles_out,
                                                int* lds_bytes_out, rgbdfe_pyramid_runner runner) {
  using namespace rgbdfe;
  OrbWorkspace ws;
  std::string err;
  if (ws.prepare(cols, rows, use_grid != 0, err, n_frames, true) != RGBDFE_OK) return -1;
  std::vector<PyrTile> tiles;
  if (ws.plan_pyramid(tiles, err) != RGBDFE_OK) return -2;
  const PyrPlan plan = ws.pyr_plan;
  if (n_tiles_out) *n_tiles_out = (int)tiles.size();
  if (lds_bytes_out) *lds_bytes_out = plan.buf_bytes[0] + plan.buf_bytes[1] + 8 * (plan.max_rw + plan.max_rh);
  const size_t level0 = (size_t)2 * cols * rows * n_frames;
  std::vector<uint8_t> a(ws.pool_bytes, 0x55), b(ws.pool_bytes, 0xAA);
  uint32_t x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < level0; ++i) {
    x = x * 1664525u + 1013904223u;
    const bool is_mask = (i / ((size_t)cols * rows)) & 1;
    const uint8_t v = (uint8_t)(x >> 24);
    a[i] = is_mask ? (v < 40 ? 0 : (v < 60 ? v : 255)) : v;   // masks: mostly 255, holes, a few grey values
    b[i] = a[i];
  }
  auto pixel = [](int p00, int p01, int p10, int p11, const ResizeTapX& tx, const ResizeTapY& ty, bool is_mask) {
    const int h0 = p00 * tx.w0 + p01 * tx.w1, h1 = p10 * tx.w0 + p11 * tx.w1;
    int v = (((ty.b0 * (h0 >> 4)) >> 16) + ((ty.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = std::min(std::max(v, 0), 255);
    if (is_mask && v <= 254) v = 0;
    return (uint8_t)v;
  };
  // (A) level by level
  for (int l = 1; l < kLevels; ++l)
    for (int k = ws.level_job_begin[l]; k < ws.level_job_begin[l + 1]; ++k) {
      const ResizeJob& j = ws.jobs[k];
      for (int dy = 0; dy < j.dh; ++dy) {
        const ResizeTapY ty = resize_tap_y(dy, j.scale_y, j.sh);
        for (int dx = 0; dx < j.dw; ++dx) {
          const ResizeTapX tx = resize_tap_x(dx, j.scale_x, j.sw);
          const uint8_t* r0 = a.data() + j.src_off + (size_t)ty.r0 * j.sstride;
          const uint8_t* r1 = a.data() + j.src_off + (size_t)ty.r1 * j.sstride;
          a[j.dst_off + (size_t)dy * j.dw + dx] = pixel(r0[tx.s0], r0[tx.s1], r1[tx.s0], r1[tx.s1], tx, ty, j.is_mask != 0);
        }
      }
    }
  // (B) the kernel, tile by tile
  std::vector<uint8_t> lds((size_t)plan.buf_bytes[0] + plan.buf_bytes[1]);
  struct Tab { uint16_t a, b; int16_t w0, w1; };
  std::vector<Tab> xtab(plan.max_rw), ytab(plan.max_rh);
  const int buf_off[2] = {0, plan.buf_bytes[0]};
  const int buf_end[2] = {plan.buf_bytes[0], plan.buf_bytes[0] + plan.buf_bytes[1]};
  if (runner) runner(b.data(), ws.jobs.data(), tiles.data(), (int)tiles.size(), &plan);
  for (const PyrTile& t : tiles) {
    if (runner) break;
    int px0 = t.nx0[0], py0 = t.ny0[0], prw = (t.nx1[0] - t.nx0[0] + 3) & ~3;
    {
      const ResizeJob& j = ws.jobs[(size_t)plan.level_job_begin[1] + t.chain];
      const int rows0 = t.ny1[0] - py0;
      if (prw * rows0 > plan.buf_bytes[1]) return -3;
      for (int r = 0; r < rows0; ++r)
        for (int c = 0; c < prw; ++c) {
          const size_t src = j.src_off + (size_t)(py0 + r) * j.sstride + px0 + c;
          lds[buf_off[1] + r * prw + c] = src < b.size() ? b[src] : 0;   // (dwords may reach past the region: the pool's slack)
        }
    }
    for (int l = 1; l < kLevels; ++l) {
      const int x0 = t.nx0[l], x1 = t.nx1[l], y0 = t.ny0[l], y1 = t.ny1[l];
      const int rw = x1 - x0, rh = y1 - y0;
      if (rw <= 0 || rh <= 0) break;
      if (rw > plan.max_rw || rh > plan.max_rh) return -3;
      const ResizeJob& j = ws.jobs[(size_t)plan.level_job_begin[l] + t.chain];
      for (int i = 0; i < rw; ++i) {
        const ResizeTapX tx = resize_tap_x(x0 + i, j.scale_x, j.sw);
        xtab[i] = Tab{(uint16_t)(tx.s0 - px0), (uint16_t)(tx.s1 - px0), (int16_t)tx.w0, (int16_t)tx.w1};
      }
      for (int i = 0; i < rh; ++i) {
        const ResizeTapY ty = resize_tap_y(y0 + i, j.scale_y, j.sh);
        ytab[i] = Tab{(uint16_t)((ty.r0 - py0) * prw), (uint16_t)((ty.r1 - py0) * prw), (int16_t)ty.b0, (int16_t)ty.b1};
      }
      const int cur = buf_off[(l + 1) & 1], cur_end = buf_end[(l + 1) & 1], prev = buf_off[l & 1], prev_end = buf_end[l & 1];
      if (cur + rw * rh > cur_end) return -3;
      for (int y = 0; y < rh; ++y)
        for (int xx = 0; xx < rw; ++xx) {
          const Tab cx = xtab[xx], cy = ytab[y];
          const int i00 = prev + cx.a + cy.a, i01 = prev + cx.b + cy.a, i10 = prev + cx.a + cy.b, i11 = prev + cx.b + cy.b;
          if (std::max(std::max(i00, i01), std::max(i10, i11)) >= prev_end) return -3;
          ResizeTapX tx{}; tx.w0 = cx.w0; tx.w1 = cx.w1;
          ResizeTapY ty{}; ty.b0 = cy.w0; ty.b1 = cy.w1;
          const uint8_t v = pixel(lds[i00], lds[i01], lds[i10], lds[i11], tx, ty, j.is_mask != 0);
          lds[cur + y * rw + xx] = v;
          if (x0 + xx >= t.ox0[l] && x0 + xx < t.ox1[l] && y0 + y >= t.oy0[l] && y0 + y < t.oy1[l])
            b[j.dst_off + (size_t)(y0 + y) * j.dw + (x0 + xx)] = v;
        }
      px0 = x0; py0 = y0; prw = rw;
    }
  }
  // every byte behind the level-0 images that belongs to a level >= 1 must agree (A pre-filled 0x55, B 0xAA: a pixel nobody
  // wrote differs); bytes between images (none) or the slack are not compared
  int diff = 0;
  for (int l = 1; l < kLevels; ++l)
    for (int k = ws.level_job_begin[l]; k < ws.level_job_begin[l + 1]; ++k) {
      const ResizeJob& j = ws.jobs[k];
      for (size_t i = 0; i < (size_t)j.dw * j.dh; ++i) diff += a[j.dst_off + i] != b[j.dst_off + i];
    }
  return diff;
}

// tests (host only): the pool of a workspace of the given shape after `runner` (the fused pyramid kernel, see above) has run
// over the caller's level-0 images ([frame f gray | frame f mask] x n_frames, cols x rows bytes each), and the resize jobs
// that describe where every level of every image lies in it.  Returns the pool size in bytes (the caller's capacity must
// cover it), or a negative number.
extern "C" long rgbdfe_debug_pyramid_run(int cols, int rows, int use_grid, int n_frames, const uint8_t* level0,
                                         rgbdfe_pyramid_runner runner, uint8_t* pool_out, long pool_capacity,
                                         rgbdfe::ResizeJob* jobs_out, int jobs_capacity, int* n_jobs_out) {
  using namespace rgbdfe;
  OrbWorkspace ws;
  std::string err;
  if (!runner || ws.prepare(cols, rows, use_grid != 0, err, n_frames, true) != RGBDFE_OK) return -1;
  std::vector<PyrTile> tiles;
  if (ws.plan_pyramid(tiles, err) != RGBDFE_OK) return -2;
  if ((long)ws.pool_bytes > pool_capacity || (int)ws.jobs.size() > jobs_capacity) return -3;
  std::vector<uint8_t> pool(ws.pool_bytes, 0x5A);
  memcpy(pool.data(), level0, (size_t)2 * cols * rows * n_frames);
  runner(pool.data(), ws.jobs.data(), tiles.data(), (int)tiles.size(), &ws.pyr_plan);
  memcpy(pool_out, pool.data(), ws.pool_bytes);
  memcpy(jobs_out, ws.jobs.data(), sizeof(ResizeJob) * ws.jobs.size());
  if (n_jobs_out) *n_jobs_out = (int)ws.jobs.size();
  return (long)ws.pool_bytes;
}
