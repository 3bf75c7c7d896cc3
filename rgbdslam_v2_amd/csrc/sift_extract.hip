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
#include <chrono>

#include "rgbdfe_internal.h"
#include "sift_pyramid_kernels.h"

namespace rgbdfe {

namespace {

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
// (Round 5 tried a workgroup per feature that evaluates the gradients of the whole 4 x 4 window once into LDS -- a pixel
// lies in up to four cells' supports -- and lets the cells read them: bit-identical output, but 43 instead of 37 us per VGA
// frame.  The window's bounding box holds 1.6 x the pixels the cells' rotated supports cover, and 51 KB of LDS left 12
// waves per CU for a loop that lives on latency hiding: profiles/r05_logs/sift_descriptor_staged.txt.)
// the wave's sum of v, valid in lane 63 (rows of 16 by row_shr 1, 2, 4, 8; then row 0 -> 1, 2 -> 3 and 1 -> 2, 3)
#define SIFT_DPP_F32(x, ctrl, rows, bound) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rows, 0xF, bound))
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += SIFT_DPP_F32(v, 0x111, 0xF, true);    // row_shr:1
  v += SIFT_DPP_F32(v, 0x112, 0xF, true);    // row_shr:2
  v += SIFT_DPP_F32(v, 0x114, 0xF, true);    // row_shr:4
  v += SIFT_DPP_F32(v, 0x118, 0xF, true);    // row_shr:8
  v += SIFT_DPP_F32(v, 0x142, 0xA, false);   // row_bcast:15 into rows 1 and 3
  v += SIFT_DPP_F32(v, 0x143, 0xC, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
#undef SIFT_DPP_F32

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
  const float rnx = 1.0f / (float)nx;   // jy = t / nx through the reciprocal: (t + 0.5) / nx is at least 0.5 / nx away from
                                        // an integer and t, nx < 2^12 here, so the rounding of the product cannot cross one
  for (int t = lane; t < total; t += 64) {
    const int jy = (int)(((float)t + 0.5f) * rnx), jx = t - jy * nx;
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
  // the lanes' bins -> lane 63: four row_shr steps inside the rows of 16, then the rows' totals across (DPP adds: no LDS
  // round trip per step, as the xor butterfly of rounds 3 - 4 had -- a fifth of the kernel's instructions).  A fixed order.
#pragma unroll
  for (int i = 0; i < 9; ++i) des[i] = wave_sum_to_lane63(des[i]);
  if (lane != 63) return;
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
  // (under RGBDFE_SIFT_GRAPH=1 this becomes a memset NODE; the pair path once saw such a node not in effect on replay
  //  (ransac_split.hip, ransac_hyp_kernel) and zeroes its counters in a kernel since -- one more reason the graph is opt-in)
  SIFT_HIP(hipMemsetAsync(d_rowcnt, 0, sizeof(int) * (size_t)total_rows * nf, s));
  launch_key_flags(*this, nf, st, s);
  launch_key_lists(*this, nf, st, s);
  SIFT_HIP(hipGetLastError());
  SIFT_HIP(hipMemcpyAsync(h_counts, d_lvltot, sizeof(int) * 64 * (size_t)nf, hipMemcpyDeviceToHost, s));
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
  hipLaunchKernelGGL(sift_orientation_kernel, dim3(max_total, NF), dim3(64), 0, s, static_cast<const LevelJobs*>(d_jobs), d_cand, d_feat,
                     sigma_step, 1.5f, 1.5f * 2.0f);
  SIFT_HIP(hipGetLastError());
  SIFT_HIP(hipMemcpyAsync(h_stage, d_feat, (size_t)grand * 16, hipMemcpyDeviceToHost, s));
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
