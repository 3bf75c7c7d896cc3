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
//   * a keypoint's orientation histogram is a wave-wide job (the reference gives it one thread), and so is a whole
//     DESCRIPTOR: the wave walks the feature's 5 x 5-cell support once and splits each pixel's vote over the cells and
//     directions around it (the reference walks every cell's own support, a pixel up to four times) -- see the two kernels.
// Arithmetic without transcendental functions (pyramid, extrema, sub-pixel offsets, lists) is exact against the
// reference's kernels compiled on the CPU emulation (oracle/_ref/libref_siftgpu.so); orientations and descriptors involve
// exp / atan2 / pow / sincos and a different order of summation: tests/test_gpu_sift_extract.py states the tolerances.
#include "sift_extract.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "rgbdfe_internal.h"
#include "sift_pyramid_kernels.h"

namespace rgbdfe {

namespace {

constexpr float kPi = 3.14159265358979323846f;

// ---- what the orientation and descriptor stages see of the image: central differences of a Gaussian plane ---------------------
// (the reference keeps a gradient-magnitude and an angle plane per level, written by its DoG kernel, ProgramCU.cu:466-473;
// here they are recomputed where they are read: half the difference vector's length, atan2 of it, 0 for a flat spot)
struct PolarGradient { float len, dir; };
__device__ __forceinline__ PolarGradient polar_gradient(const float* __restrict__ plane, int stride, int col, int row) {
  const float* p = plane + (size_t)row * stride + col;
  const float gh = p[1] - p[-1], gv = p[stride] - p[-stride];
  PolarGradient r;
  r.len = 0.5f * sqrtf(gh * gh + gv * gv);
  r.dir = r.len == 0.0f ? 0.0f : atan2f(gv, gh);
  return r;
}

// The same for the descriptor stage, whose outputs are continuous in the direction (a vote is SPLIT between the two nearest
// direction bins, none is moved whole): hardware square root (1 ulp) and an arctangent of |error| < 3e-7 rad -- one
// reciprocal, the octant folded to |t| <= tan(pi / 8) by (mn - mx) / (mn + mx), a four-term odd polynomial (the classic
// single-precision minimax set for that interval) -- in a quarter of libm's instructions.  3e-7 rad is 4e-7 of a
// direction bin; tests/test_gpu_sift_extract.py allows descriptors 1e-3.
#ifndef RGBDFE_SIFT_DESC_LIBM
#define RGBDFE_SIFT_DESC_LIBM 0   // 1: libm's sqrtf / atan2f / expf in the descriptor kernel (A/B: tools/sift_extract_ab.sh)
#endif
__device__ __forceinline__ PolarGradient polar_gradient_quick(const float* __restrict__ plane, int stride, int col, int row) {
#if RGBDFE_SIFT_DESC_LIBM
  return polar_gradient(plane, stride, col, row);
#else
  const float* p = plane + (size_t)row * stride + col;
  const float gh = p[1] - p[-1], gv = p[stride] - p[-stride];
  PolarGradient r;
  r.len = 0.5f * __builtin_amdgcn_sqrtf(gh * gh + gv * gv);
  const float ah = fabsf(gh), av = fabsf(gv);
  const float mn = fminf(ah, av), mx = fmaxf(ah, av);
  const bool upper = mn > 0.41421356f * mx;               // beyond pi / 8: measure from the diagonal instead
  const float t = (upper ? mn - mx : mn) * __builtin_amdgcn_rcpf(upper ? mn + mx : mx);
  const float t2 = t * t;
  float ang = ((((8.05374449538e-2f * t2 - 1.38776856032e-1f) * t2 + 1.99777106478e-1f) * t2 - 3.33329491539e-1f) * t2) * t + t;
  ang += upper ? 0.25f * kPi : 0.0f;
  ang = av > ah ? 0.5f * kPi - ang : ang;                 // first octant pair -> first quadrant
  ang = gh < 0.0f ? kPi - ang : ang;
  ang = gv < 0.0f ? -ang : ang;
  r.dir = mx == 0.0f ? 0.0f : ang;
  return r;
#endif
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

// the segment an item lies in: lane l holds begin[l + 1]; the segments at or before the item are a ballot's popcount
__device__ __forceinline__ int segment_of(const LevelJobs& jobs, int item, int lane) {
  const bool before = lane + 1 < jobs.n && item >= jobs.begin[lane + 1];
  return __popcll(__ballot(before));
}

// The pixel-centre window both stages scan around a point: every pixel whose centre (index + 0.5) lies within `reach` of the
// point (by the floor of the window's edges, as the reference's loops run), kept one pixel off the plane's border so that
// the central differences exist.
struct Window { int c0, r0, cols, rows; };
__device__ __forceinline__ Window window_around(float cx, float cy, float reach, int width, int height) {
  Window wd;
  wd.c0 = max(1, (int)floorf(cx - reach));
  wd.r0 = max(1, (int)floorf(cy - reach));
  wd.cols = max(0, min(width - 2, (int)floorf(cx + reach)) - wd.c0 + 1);
  wd.rows = max(0, min(height - 2, (int)floorf(cy + reach)) - wd.r0 + 1);
  return wd;
}

__device__ __forceinline__ float lane_value(float v, int src) { return __shfl(v, src); }

// The descriptor stage sums its votes in FIXED POINT: a non-negative vote becomes round(vote * 2^k) and goes into a 32-bit LDS
// counter by ds_add_u32.  Two reasons.  Integer sums do not depend on the order of the additions, so every histogram is the
// same bit pattern on every run and for every dealing of pixels to lanes, by construction.  And the LDS adds integers at the
// rate it writes, whereas ds_add_f32 was measured at ~35 cycles per wave instruction whatever the collisions (descriptor kernel
// 52 us per VGA frame with float adds, 18 with integer ones: profiles/r06_logs/sift_descriptor_forms.txt).  k is chosen per
// feature from a bound of the largest possible sum (gradient length <= 0.7072 for planes in [0, 1], weights <= 1): the
// counters cannot wrap, the grid is 2^-31 of that bound (~1e-7 absolute for a typical feature, where votes are 1e-3 .. 1).
// (The ORIENTATION stage keeps float sums: its peak tests are strict comparisons of neighbouring bins, and on mirror-symmetric
// patterns exact sums produce exact ties -- no peak at all -- where the reference's float rounding leaves one bin ahead:
// tests/test_gpu_sift_extract.py, the 0 / 255 block pattern, lost 0.7 % of its orientations to that.)
__device__ __forceinline__ float fixed_point_scale(float sum_bound) {   // the power of two with sum_bound * scale < 2^31
  int e;
  (void)frexpf(sum_bound, &e);                                          // sum_bound < 2^e
  return ldexpf(1.0f, 31 - e);
}
__device__ __forceinline__ void vote(unsigned* counter, float v) { atomicAdd(counter, (unsigned)rintf(v)); }

// ---- orientation assignment: what SiftGPU's ComputeOrientation_Kernel computes (ProgramCU.cu:774-935 with num_orientation = 2,
//      sub-pixel positions, detected keypoints), organised for a 64-wide wave ------------------------------------------------------
// One wave per keypoint candidate.
//   votes   the window's pixels are dealt to the lanes in raster order; a pixel inside the disc adds its gradient length, damped
//           by a Gaussian of the distance, to one of 36 direction bins -- each lane into its own column of a [36][64 + 1] LDS
//           array (plain read-add-write, no atomics); lane b < 36 then adds bin b's 64 partial sums in lane order: one fixed
//           order of float additions whatever the hardware does
//   smooth  lanes 0 .. 35 hold a bin each; six circular box filters are lane rotations
//   peaks   a ballot of the local maxima above 0.8 of the largest bin; the strongest and the runner-up by two wave-wide
//           arg-max reductions (the lower bin wins a tie, as a first-come scan would have it); each peak's position is refined
//           by the parabola through its neighbours and leaves as a 16-bit fraction of a turn, 0xFFFF = no such peak
constexpr int kDirBins = 36;
__global__ __launch_bounds__(64) void sift_orientation_kernel(const LevelJobs* __restrict__ jobs_of_frame,
                                                              const float* __restrict__ cand, float4* __restrict__ feat,
                                                              float level_ratio, float damping_scales, float disc_scales) {
  __shared__ float column[kDirBins][64 + 1];   // + 1: bin b's row starts in bank b, the final sums read 36 banks at a time
  const LevelJobs& jobs = jobs_of_frame[blockIdx.y];
  const int item = blockIdx.x;
  if (item >= jobs.begin[jobs.n]) return;   // the grid is sized for the batch's largest frame
  const int lane = threadIdx.x;
  const int seg = segment_of(jobs, item, lane);
  const float* rec = cand + (size_t)(jobs.src_off[seg] + item - jobs.begin[seg]) * 6;   // column, row, sign, d_col, d_row, d_level
  const float* __restrict__ plane = jobs.g[seg];
  const int stride = jobs.w[seg];
  // the refined keypoint in pixel-centre coordinates and its scale between the levels
  const float cx = (rec[0] + 0.5f) + rec[3], cy = (rec[1] + 0.5f) + rec[4];
  const float scale = jobs.sigma[seg] * powf(level_ratio, rec[5]);
  const float disc = fabsf(scale) * disc_scales;
  const float disc2 = disc * disc + 0.5f;
  const float spread = scale * damping_scales;
  const float damp = -0.5f / (spread * spread);
  const Window wd = window_around(cx, cy, disc, stride, jobs.h[seg]);
#pragma unroll
  for (int b = 0; b < kDirBins; ++b) column[b][lane] = 0.0f;
  const int pixels = wd.cols * wd.rows;
  const float per_col = 1.0f / (float)max(wd.cols, 1);
  for (int t = lane; t < pixels; t += 64) {
    // row = t / cols through the reciprocal: (t + 0.5) / cols is at least 0.5 / cols away from an integer and t, cols < 2^12
    // here, so the rounding of the product cannot cross one
    const int r = (int)(((float)t + 0.5f) * per_col), c = t - r * wd.cols;
    const float ox = ((float)(wd.c0 + c) + 0.5f) - cx, oy = ((float)(wd.r0 + r) + 0.5f) - cy;
    const float d2 = ox * ox + oy * oy;
    if (d2 >= disc2) continue;
    const PolarGradient g = polar_gradient(plane, stride, wd.c0 + c, wd.r0 + r);
    int bin = (int)floorf(g.dir * (18.0f / kPi));   // ten degrees per bin, -pi .. pi -> -18 .. 18
    bin += bin < 0 ? kDirBins : 0;
    column[bin][lane] += g.len * expf(d2 * damp);
  }
  __syncthreads();
  const bool owner = lane < kDirBins;
  float v = 0.0f;
  if (owner)
    for (int l = 0; l < 64; ++l) v += column[lane][l];
  const int left = owner ? (lane == 0 ? kDirBins - 1 : lane - 1) : lane;
  const int right = owner ? (lane == kDirBins - 1 ? 0 : lane + 1) : lane;
  const float third = 1.0 / 3.0;
  for (int pass = 0; pass < 6; ++pass) v = third * ((lane_value(v, left) + v) + lane_value(v, right));
  float top = owner ? v : -1.0f;
  for (int d = 32; d >= 1; d >>= 1) top = fmaxf(top, __shfl_xor(top, d));
  const float vl = lane_value(v, left), vr = lane_value(v, right);
  uint64_t peaks = __ballot(owner && v > 0.8f * top && v > vl && v > vr);
  // this bin's refined direction as a fraction of a turn, quantised (only read from peak lanes)
  float turn = ((float)lane + 0.5f * ((vr - vl) / (v + v - vr - vl)) + 0.5f) / (float)kDirBins;
  turn += turn < 0.0f ? 1.0f : 0.0f;
  const float quantised = floorf(turn * 65535.0f);
  unsigned code[2] = {0xFFFFu, 0xFFFFu};
#pragma unroll
  for (int rank = 0; rank < 2; ++rank) {
    if (peaks == 0) break;
    const bool in = (peaks >> lane) & 1;
    float best = in ? v : -1.0f;
    for (int d = 32; d >= 1; d >>= 1) best = fmaxf(best, __shfl_xor(best, d));
    const int who = __builtin_ctzll(__ballot(in && v == best));
    code[rank] = (unsigned)lane_value(quantised, who) & 0xFFFFu;
    peaks &= ~(1ull << who);
  }
  if (lane == 0) feat[jobs.base + item] = make_float4(cx, cy, scale, __uint_as_float(code[0] | (code[1] << 16)));
}

// ---- descriptors: what SiftGPU's ComputeDescriptor_Kernel<false> computes (ProgramCU.cu:967-1046, "-unn": raw histograms),
//      turned inside out ---------------------------------------------------------------------------------------------------------------
// The reference runs a thread per (feature, cell): each of the 16 cells walks the pixels of its own rotated 2 x 2-cell
// support, so a pixel's gradient (4 loads, a square root, an atan2) is evaluated by up to four cells, and the cell keeps 8 bins.
// Here a WAVE owns a feature and walks the feature's whole 5 x 5-cell support ONCE: a pixel is taken into the feature's
// frame -- (a, b) = M (pixel - feature) with the 2 x 2 matrix M = rotation / cell size, direction o relative to the
// feature's in units of 45 degrees -- and its vote, gradient length x Gaussian of the frame distance, is split trilinearly:
// (1 - fa, fa) x (1 - fb, fb) over the four cells around (a, b) that exist, (1 - fo, fo) over the two directions around o:
// eight ds_add_u32 into this wave's LDS histogram (fixed point, see above).  That is the same sum as the reference's:
// cell (i, j) there collects exactly the pixels with |a - i| < 1 and |b - j| < 1, weighted (1 - |a - i|)(1 - |b - j|).
// The wave then writes its 512 bytes of histogram in one coalesced store.  Per feature 25 instead of 64 cell areas of
// gradient evaluations, one wave prologue instead of 16.  Measured per VGA frame (~1550 features): 35 us for round 5's
// wave-per-cell kernel -> 18 us (libm) -> see profiles/r06_logs/sift_descriptor_forms.txt for the quick-math figure.
constexpr int kDescBins = 128;   // 4 x 4 cells x 8 directions
#ifndef RGBDFE_SIFT_DESC_COPIES
#define RGBDFE_SIFT_DESC_COPIES 4
#endif
// Neighbouring pixels mostly vote for the same (cell, direction).  The wave keeps kCopies interleaved histograms, lane l
// adding to copy l mod kCopies at [bin][copy]: fewer lanes of one instruction meet in a counter (measured 20.3 us with one
// copy, 18.2 with four, per VGA frame).
constexpr int kCopies = RGBDFE_SIFT_DESC_COPIES;
__global__ __launch_bounds__(64) void sift_descriptor_kernel(const LevelJobs* __restrict__ jobs_of_frame,
                                                             const float4* __restrict__ feat, float2* __restrict__ out,
                                                             float cell_scales) {
  __shared__ unsigned hist[kDescBins * kCopies];
  const LevelJobs& jobs = jobs_of_frame[blockIdx.y];
  const int item = blockIdx.x;
  if (item >= jobs.begin[jobs.n]) return;
  const int lane = threadIdx.x;
  const int seg = segment_of(jobs, item, lane);
  const float* __restrict__ plane = jobs.g[seg];
  const int stride = jobs.w[seg];
  const float4 f = feat[jobs.base + item];          // column, row, scale, direction in [0, 2 pi)
  const float cell = fabsf(f.z * cell_scales);      // a cell's side in pixels of this level
  float sn, cs;
  sincosf(f.w, &sn, &cs);
  const float facing = f.w > kPi ? f.w - 2.0f * kPi : f.w;
  const float m_c = cs / cell, m_s = sn / cell;     // M = [m_c m_s; -m_s m_c]
  // the support |a|, |b| < 2.5 is a rotated square; its bounding box in the image
  const Window wd = window_around(f.x, f.y, 2.5f * cell * (fabsf(cs) + fabsf(sn)), stride, jobs.h[seg]);
  // a bin's sum: at most 0.7072 x the lattice sum of a cell's tent weights, which is about cell^2 and below (cell + 1)^2
  const float to_fixed = fixed_point_scale((cell + 2.0f) * (cell + 2.0f));
#pragma unroll
  for (int i = 0; i < kDescBins * kCopies / 64; ++i) hist[i * 64 + lane] = 0u;
  __syncthreads();
  unsigned* mine = hist + (lane & (kCopies - 1));
  const int pixels = wd.cols * wd.rows;
  const float per_col = 1.0f / (float)max(wd.cols, 1);
  for (int t = lane; t < pixels; t += 64) {
    const int r = (int)(((float)t + 0.5f) * per_col), c = t - r * wd.cols;   // see the orientation kernel
    const float ox = ((float)(wd.c0 + c) + 0.5f) - f.x, oy = ((float)(wd.r0 + r) + 0.5f) - f.y;
    const float a = m_c * ox + m_s * oy, b = m_c * oy - m_s * ox;
    if (!(fabsf(a) < 2.5f && fabsf(b) < 2.5f)) continue;
    const PolarGradient g = polar_gradient_quick(plane, stride, wd.c0 + c, wd.r0 + r);
#if RGBDFE_SIFT_DESC_LIBM
    const float weight = g.len * expf(-0.125f * (a * a + b * b)) * to_fixed;
#else
    const float weight = g.len * __builtin_amdgcn_exp2f((-0.125f * 1.44269504f) * (a * a + b * b)) * to_fixed;
#endif
    // cells sit at -1.5, -0.5, 0.5, 1.5: the lower neighbour's index and the distance past it
    const float ai = floorf(a + 1.5f), bi = floorf(b + 1.5f);
    const float fa = (a + 1.5f) - ai, fb = (b + 1.5f) - bi;
    const int ia = (int)ai, ib = (int)bi;            // -1 .. 3
    float o = (facing - g.dir) * (4.0f / kPi);
    o += o < 0.0f ? 8.0f : 0.0f;
    const float oi = floorf(o);
    const float fo = o - oi;
    const int d0 = (int)oi & 7, d1 = ((int)oi + 1) & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ja = ia + (q & 1), jb = ib + (q >> 1);
      if ((unsigned)ja > 3u || (unsigned)jb > 3u) continue;
      const float share = weight * ((q & 1) ? fa : 1.0f - fa) * ((q >> 1) ? fb : 1.0f - fb);
      unsigned* cell_bins = mine + (jb * 4 + ja) * 8 * kCopies;
      vote(cell_bins + d0 * kCopies, share * (1.0f - fo));
      vote(cell_bins + d1 * kCopies, share * fo);
    }
  }
  __syncthreads();
  // the copies of a bin added up (integers: any order), back to float; the wave's 512 bytes leave in one store
  unsigned even = 0u, odd = 0u;
#pragma unroll
  for (int k = 0; k < kCopies; ++k) {
    even += hist[(2 * lane) * kCopies + k];
    odd += hist[(2 * lane + 1) * kCopies + k];
  }
  out[(size_t)(jobs.base + item) * (kDescBins / 2) + lane] = make_float2((float)even / to_fixed, (float)odd / to_fixed);
}

#define SIFT_HIP(expr)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RGBDFE_ERR_HIP; } \
  } while (0)

}  // namespace

SiftExtractor::~SiftExtractor() { release(); }

void SiftExtractor::release() {
  for (int i = 0; i <= kMaxBatch; ++i) {   // the captured launch chains hold the buffers' addresses
    if (begin_exec[i]) (void)hipGraphExecDestroy(begin_exec[i]);
    if (begin_graph[i]) (void)hipGraphDestroy(begin_graph[i]);
    begin_exec[i] = nullptr; begin_graph[i] = nullptr;
  }
  begin_capture_failed = false;
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
  if (d_key_tiles) (void)hipFree(d_key_tiles);
  d_key_tiles = nullptr; n_key_tiles = 0;
  if (d_octs) (void)hipFree(d_octs);
  d_octs = nullptr;
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

// the buffers of nf frames of this size, side by side (geometry: plan_geometry / bind_levels, sift_pyramid_kernels.h)
// (the tables go up on the caller's stream, followed by a wait: a synchronous hipMemcpy is an operation of the legacy stream,
// and the runtime refuses those -- "would make the legacy stream depend on a capturing blocking stream" -- while ANOTHER
// thread of the process has a stream capture open, e.g. another context recording its own launch chain:
// tests/test_gpu_sift_threads.py)
int SiftExtractor::prepare(int rows, int cols, int nf, hipStream_t s, std::string& err) {
  const char* const hw_env = getenv("RGBDFE_SIFT_HOSTWRITE");   // (read per call: the tests switch it inside one process)
  host_write = !(hw_env && atoi(hw_env) == 0);
  init_params();
  if (rows == H && cols == W && d_planes && nf <= frames_cap) return RGBDFE_OK;
  if (rows == H && cols == W && nf < frames_cap) nf = frames_cap;
  release();
  const int rc = plan_geometry(rows, cols, err);
  if (rc != RGBDFE_OK) return rc;
  const size_t F = (size_t)nf;
  SIFT_HIP(hipMalloc((void**)&d_gray, F * rows * cols));
  SIFT_HIP(hipMalloc((void**)&d_input, F * input_floats * 4));
  SIFT_HIP(hipMalloc((void**)&d_up, F * oct[0].plane * 4));
  SIFT_HIP(hipMalloc((void**)&d_planes, F * planes_floats * 4));
  SIFT_HIP(hipMalloc((void**)&d_flags, F * flags_bytes));
  // rowcnt [nf][total_rows] | rowoff [nf][total_rows] | row2lvl [total_rows] | lvltot [nf][64]
  SIFT_HIP(hipMalloc((void**)&d_rowcnt, sizeof(int) * ((size_t)total_rows * (2 * F + 1) + 64 * F)));
  SIFT_HIP(hipMemsetAsync(d_rowcnt, 0, sizeof(int) * (size_t)total_rows * F, s));
  d_rowoff = d_rowcnt + (size_t)total_rows * F;
  d_lvltot = d_rowcnt + (size_t)total_rows * (2 * F + 1);
  bind_levels();
  SIFT_HIP(hipMalloc((void**)&d_levels, sizeof(LevelDesc) * h_levels.size()));
  SIFT_HIP(hipMemcpyAsync(d_levels, h_levels.data(), sizeof(LevelDesc) * h_levels.size(), hipMemcpyHostToDevice, s));
  SIFT_HIP(hipMemcpyAsync(d_rowcnt + (size_t)total_rows * 2 * F, h_row2lvl.data(), sizeof(int) * (size_t)total_rows, hipMemcpyHostToDevice, s));
  SIFT_HIP(hipMalloc((void**)&d_octs, sizeof(OctDesc) * h_octs.size()));
  SIFT_HIP(hipMemcpyAsync(d_octs, h_octs.data(), sizeof(OctDesc) * h_octs.size(), hipMemcpyHostToDevice, s));
  n_key_tiles = (int)h_key_tiles.size();
  SIFT_HIP(hipMalloc((void**)&d_key_tiles, sizeof(KeyTile) * h_key_tiles.size()));
  SIFT_HIP(hipMemcpyAsync(d_key_tiles, h_key_tiles.data(), sizeof(KeyTile) * h_key_tiles.size(), hipMemcpyHostToDevice, s));
  SIFT_HIP(hipStreamSynchronize(s));   // the host vectors may change before the copies would otherwise have run
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
  for (int f = 0; f < nf; ++f) memcpy(h_gray + (size_t)f * gray_cap, gray[f], gray_cap);
  SIFT_HIP(hipMemcpyAsync(d_gray, h_gray, (size_t)nf * gray_cap, hipMemcpyHostToDevice, s));
  launch_pyramid(*this, nf, s);
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
  int rc = prepare(rows, cols, nf, s, err);
  if (rc != RGBDFE_OK) return rc;
  const auto t_in = std::chrono::steady_clock::now();
  for (int f = 0; f < nf; ++f) memcpy(h_gray + (size_t)f * gray_cap, gray[f], gray_cap);
  stage_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count();
  // Everything this half enqueues -- the frames' upload from the pinned stage, 50 pyramid launches, the extremum flags, the
  // scan, the ordered emit, the counts' download -- has the same arguments for every batch of nf frames of this size, so it
  // CAN be captured once per nf as a hipGraph and replayed with one hipGraphLaunch: RGBDFE_SIFT_GRAPH=1.  Measured: the
  // calling thread's time in this function falls from 0.86 to 0.27 ms per 32-frame call, the wall clock does not move (the
  // device is the bound: profiles/r05_logs/sift_host_steps.txt).  It is therefore NOT the default: while a capture is open --
  // even a relaxed one on a non-blocking stream -- the runtime refuses NULL-stream operations of every other thread of the
  // process ("operation would make the legacy stream depend on a capturing blocking stream"; tests/test_gpu_sift_threads.py
  // met it in this library's own allocation-time copies, which no longer use the NULL stream -- a caller's own code may).
  static const bool graph_env = getenv("RGBDFE_SIFT_GRAPH") && atoi(getenv("RGBDFE_SIFT_GRAPH")) != 0;
  bool launched = false;
  if (graph_env) {
    if (!begin_exec[nf] && !begin_capture_failed) {
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
        std::string cap_err;
        const int crc = enqueue_begin(nf, s, cap_err);
        hipError_t ce = hipStreamEndCapture(s, &g);
        if (crc == RGBDFE_OK && ce == hipSuccess) ce = hipGraphInstantiate(&begin_exec[nf], g, nullptr, nullptr, 0);
        if (crc != RGBDFE_OK || ce != hipSuccess) {
          if (begin_exec[nf]) (void)hipGraphExecDestroy(begin_exec[nf]);
          begin_exec[nf] = nullptr;
          if (g) (void)hipGraphDestroy(g);
          g = nullptr;
          begin_capture_failed = true;
        }
        begin_graph[nf] = g;
      } else {
        begin_capture_failed = true;
      }
      (void)hipGetLastError();
    }
    if (begin_exec[nf]) {
      SIFT_HIP(hipGraphLaunch(begin_exec[nf], s));
      launched = true;
    }
  }
  if (!launched) {
    rc = enqueue_begin(nf, s, err);
    if (rc != RGBDFE_OK) return rc;
  }
  pending_nf = nf;
  return RGBDFE_OK;
}

// the enqueues of begin_batch (frames staged in h_gray): directly, or under stream capture
int SiftExtractor::enqueue_begin(int nf, hipStream_t s, std::string& err) {
  FrameStrides st{};
  st.planes = planes_floats; st.flags = flags_bytes; st.cand = cand_cap * 6; st.rows = total_rows; st.lvltot = 64;
  SIFT_HIP(hipMemcpyAsync(d_gray, h_gray, (size_t)nf * gray_cap, hipMemcpyHostToDevice, s));
  launch_pyramid(*this, nf, s);
  // ---- DetectKeypointsEX + the list part of GenerateFeatureList: flags, row counts, scan, ordered emit ----------------------
  // (the row counts are zero here: zeroed once at allocation, and the emit kernel -- their last reader -- resets every count it
  //  has used; no memset between batches, hence no memset node in the RGBDFE_SIFT_GRAPH=1 capture)
  launch_key_flags(*this, nf, st, s);
  launch_key_lists(*this, nf, st, s);
  SIFT_HIP(hipGetLastError());
  if (!host_write) SIFT_HIP(hipMemcpyAsync(h_counts, d_lvltot, sizeof(int) * 64 * (size_t)nf, hipMemcpyDeviceToHost, s));
  return RGBDFE_OK;
}

int SiftExtractor::finish_batch(int max_features, std::vector<SiftKey>* keys, const float** desc, hipStream_t s, std::string& err) {
  int rc = finish_orientations(max_features, s, err);
  if (rc == RGBDFE_OK) rc = finish_descriptors(s, err);
  return rc == RGBDFE_OK ? finish_outputs(keys, desc, s, err) : rc;
}

// The data-dependent half of a batch in three steps, each of which WAITS for what the step before enqueued, works on the
// host and enqueues the next launch -- so that a caller with two batches in flight can put the other batch's host work
// (and the copy of finished results) between them instead of sitting in a wait (rgbdfe_sift_detect_batch, api_detect.hip):
//   finish_orientations  waits for begin_batch's half; feature-count limits; enqueues the orientation launch + its download
//   finish_descriptors   waits for that; one feature per orientation, limits again; enqueues the descriptor launch + download
//   finish_outputs       waits for that; keys[f] / desc[f] of every frame
int SiftExtractor::finish_orientations(int max_features_in, hipStream_t s, std::string& err) {
  const int nf = pending_nf;
  if (nf < 1) { err = "finish_batch without begin_batch"; return RGBDFE_ERR_INVALID_ARG; }
  pending_nf = 0;
  fin_nf = nf; fin_stage = 1; fin_max_features = max_features_in;
  fin_grand = fin_grand2 = 0;
  const int max_features = max_features_in;
  const int nlv = octave_num * kDogLevels;
  const unsigned NF = (unsigned)nf;
  SIFT_HIP(hipStreamSynchronize(s));
  // ---- per frame: which levels run -- GenerateFeatureList's "-tc2" order (coarse octaves first, PyramidCU.cpp:797-850) and
  //      SiftPyramid::LimitFeatureCount(0) (SiftPyramid.cpp:170-210, _TruncateMethod = 1).  A skipped level contributes
  //      nothing (the reference leaves the previous frame's list in it: DESIGN.md 4.11) ---------------------------------
  fs.assign((size_t)nf, FrameState{});
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
  if (grand == 0) return RGBDFE_OK;   // (fin_grand stays 0: the later steps have nothing to wait for)
  if ((size_t)grand * 4 > stage_floats) { err = "SIFT staging buffer too small"; return RGBDFE_ERR_CAPACITY; }
  const float sigma_step = powf(2.0f, 1.0f / kDogLevels);
  SIFT_HIP(hipMemcpyAsync(d_jobs, h_jobs, sizeof(LevelJobs) * (size_t)nf, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(sift_orientation_kernel, dim3(max_total, NF), dim3(64), 0, s, static_cast<const LevelJobs*>(d_jobs), d_cand,
                     host_write ? reinterpret_cast<float4*>(h_stage) : d_feat, sigma_step, 1.5f, 1.5f * 2.0f);
  SIFT_HIP(hipGetLastError());
  if (!host_write) SIFT_HIP(hipMemcpyAsync(h_stage, d_feat, (size_t)grand * 16, hipMemcpyDeviceToHost, s));
  fin_grand = grand;
  return RGBDFE_OK;
}

int SiftExtractor::finish_descriptors(hipStream_t s, std::string& err) {
  if (fin_stage != 1) { err = "finish_descriptors out of order"; return RGBDFE_ERR_INVALID_ARG; }
  fin_stage = 2;
  if (fin_grand == 0) return RGBDFE_OK;
  const int nf = fin_nf, max_features = fin_max_features;
  const int nlv = octave_num * kDogLevels;
  const unsigned NF = (unsigned)nf;
  LevelJobs* hj = static_cast<LevelJobs*>(h_jobs);
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
  SIFT_HIP(hipStreamSynchronize(s));
  // ---- ReshapeFeatureListCPU (PyramidCU.cpp:501-585, NO_DUPLICATE_DOWNLOAD) + LimitFeatureCount(1), per frame ------------------
  const double full_turn = 2.0 * 3.14159265358979323846;
  const double rad_per_code = 2.0 * 3.14159265358979323846 / 65535.0;
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
      int kept = 0;
      const float oss = os * (1 << (idx / kDogLevels));
      for (int k = 0; k < cnt; ++k, src += 4) {
        unsigned short orientations[2];
        memcpy(orientations, &src[3], 4);
        auto push = [&](unsigned short o) {
          const float fo = float(rad_per_code * o);
          F.list.push_back(src[0]); F.list.push_back(src[1]); F.list.push_back(src[2]); F.list.push_back(fo);
          F.keybuf.push_back(oss * (src[0] - 0.5f) + 0.5f);
          F.keybuf.push_back(oss * (src[1] - 0.5f) + 0.5f);
          F.keybuf.push_back(oss * src[2]);
          F.keybuf.push_back((float)fmod(full_turn - fo, full_turn));
          kept++;
        };
        if (orientations[0] != 65535) {
          push(orientations[0]);
          if (orientations[1] != 65535 && orientations[1] != orientations[0]) push(orientations[1]);
        }
      }
      F.level_num[(size_t)idx] = kept;
      F.feature_num += kept;
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
  if ((size_t)grand2 * 128 > h_desc_cap) {
    if (h_desc) (void)hipHostFree(h_desc);
    h_desc = nullptr; h_desc_cap = 0;
    SIFT_HIP(hipHostMalloc((void**)&h_desc, (size_t)grand2 * 128 * 4 * 2, hipHostMallocDefault));
    h_desc_cap = (size_t)grand2 * 128 * 2;
  }
  hipLaunchKernelGGL(sift_descriptor_kernel, dim3(max_total2, NF), dim3(64), 0, s, static_cast<const LevelJobs*>(d_jobs), d_feat,
                     (float2*)(host_write ? h_desc : d_desc), 3.0f);
  SIFT_HIP(hipGetLastError());
  if (!host_write) SIFT_HIP(hipMemcpyAsync(h_desc, d_desc, (size_t)grand2 * 128 * 4, hipMemcpyDeviceToHost, s));
  fin_grand2 = grand2;
  return RGBDFE_OK;
}

int SiftExtractor::finish_outputs(std::vector<SiftKey>* keys, const float** desc, hipStream_t s, std::string& err) {
  if (fin_stage != 2) { err = "finish_outputs out of order"; return RGBDFE_ERR_INVALID_ARG; }
  fin_stage = 0;
  const int nf = fin_nf;
  for (int f = 0; f < nf; ++f) { keys[f].clear(); desc[f] = nullptr; }
  if (fin_grand2 == 0) return RGBDFE_OK;
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
  int rc = prepare(rows, cols, 1, s, err);
  if (rc != RGBDFE_OK) return rc;
  if (n <= 0) return RGBDFE_OK;
  rc = enqueue_pyramid(&gray, 1, s, err);
  if (rc != RGBDFE_OK) return rc;
  const int nlv = octave_num * kDogLevels;
  const double full_turn = 2.0 * 3.14159265358979323846;
  const float sigma_half_step = powf(2.0f, 0.5f / kDogLevels);
  float octave_sigma = octave_min >= 0 ? float(1 << octave_min) : 1.0f / (1 << (-octave_min));
  const float offset = 0.5f;   // GlobalUtil::_LoweOrigin = 0
  std::vector<float> list;     // level coordinates, level by level
  std::vector<int> caller_row;      // _keypoint_index: the input position of every list entry
  LevelJobs* dj = static_cast<LevelJobs*>(h_jobs);
  memset(dj, 0, sizeof(LevelJobs));
  int total = 0;
  for (int i = 0; i < octave_num; ++i, octave_sigma *= 2.0f)
    for (int j = 0; j < kDogLevels; ++j) {
      const float level_sg = level_sigma(j) * octave_sigma;   // GetLevelSigma(j + level_min + 1)
      const float sigma_min = level_sg / sigma_half_step, sigma_max = level_sg * sigma_half_step;
      int kept = 0;
      for (int k = 0; k < n; ++k) {
        const float sigmak = keys_in[k].s;
        if ((sigmak >= sigma_min && sigmak < sigma_max) || (sigmak < sigma_min && i == 0 && j == 0) ||
            (sigmak > sigma_max && i == octave_num - 1 && j == kDogLevels - 1)) {
          list.push_back((keys_in[k].x - offset) / octave_sigma + 0.5f);
          list.push_back((keys_in[k].y - offset) / octave_sigma + 0.5f);
          list.push_back(keys_in[k].s / octave_sigma);
          list.push_back((float)fmod(full_turn - keys_in[k].o, full_turn));
          caller_row.push_back(k);
          ++kept;
        }
      }
      if (kept == 0) continue;
      const int m = dj->n++;
      dj->begin[m] = total;
      dj->g[m] = oct[i].g[j + 1];
      dj->w[m] = oct[i].w; dj->h[m] = oct[i].h;
      total += kept;
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
    hipLaunchKernelGGL(sift_descriptor_kernel, dim3(total, 1), dim3(64), 0, s, static_cast<const LevelJobs*>(d_jobs), d_feat,
                       (float2*)d_desc, 3.0f);
    SIFT_HIP(hipGetLastError());
    SIFT_HIP(hipMemcpyAsync(h_desc, d_desc, (size_t)total * 128 * 4, hipMemcpyDeviceToHost, s));
    SIFT_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < total; ++i) memcpy(ordered + (size_t)caller_row[(size_t)i] * 128, h_desc + (size_t)i * 128, 128 * 4);
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

int SiftExtractor::debug_candidates(int octave, int dog_level, std::vector<float>& out, hipStream_t s) {
  if (octave < 0 || octave >= octave_num || dog_level < 0 || dog_level >= kDogLevels || lvl_count.empty())
    return RGBDFE_ERR_INVALID_ARG;
  const int idx = octave * kDogLevels + dog_level;
  out.resize((size_t)lvl_count[(size_t)idx] * 6);
  if (!out.empty() &&
      (hipMemcpyAsync(out.data(), d_cand + (size_t)lvl_off[(size_t)idx] * 6, out.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
       hipStreamSynchronize(s) != hipSuccess))
    return RGBDFE_ERR_HIP;
  return RGBDFE_OK;
}

}  // namespace rgbdfe
