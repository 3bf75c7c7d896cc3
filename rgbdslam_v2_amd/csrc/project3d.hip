// project3d.hip -- per-frame depth filter + back-projection on the device (gfx950).
//
// Replaces removeDepthless (src/node.cpp:67-97) + Node::projectTo3D, depth-image overload
// (src/node.cpp:900-965) + backProject (src/misc2.h:49-65): a keypoint survives iff it lies
// inside the image, its coordinates are not NaN and depth(round(y), round(x)) is not NaN;
// survivors keep their order and are cut at max_keypoints (node.cpp:957).
//
// One block of 256 lanes per frame: lane = keypoint, the order-preserving compaction is a
// __ballot / mbcnt prefix inside each wave plus a 4-entry wave-offset exchange in LDS per
// 256-keypoint chunk (a running base carries across chunks).  The depth gather is the only
// HBM traffic (one 4-byte load per keypoint).
//
// TRUNC = true is Node::projectTo3DSiftGPU (src/node.cpp:695-769, the SIFTGPU feature path): the
// lookup is depth.at<float>(p2d.y, p2d.x) -- float -> int by truncation (:733) -- and there is no
// inside-the-image test (indices are clamped to the image instead of read out of bounds).
// sift_pack_kernel re-packs the used descriptors densely (:752-766) and applies
// squareroot_descriptor_space (src/node.cpp:1557-1571, RootSIFT) to the feature_descriptors_ copy.
#include "rgbdfe_internal.h"

namespace rgbdfe {

template <bool TRUNC>
__global__ __launch_bounds__(256) void project_to_3d_kernel(
    const float2* __restrict__ kp, int n_kp, const float* __restrict__ depth, int rows, int cols,
    float fxinv, float fyinv, float cx, float cy, double depth_scaling, int max_keypoints,
    int32_t* __restrict__ kept_idx, float4* __restrict__ xyz1, int32_t* __restrict__ n_out,
    const float* __restrict__ z_gathered) {
  __shared__ uint32_t wave_cnt[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  uint32_t base = 0;
  for (int c0 = 0; c0 < n_kp; c0 += 256) {
    const int i = c0 + tid;
    bool keep = false;
    float px = 0.f, py = 0.f, Z = 0.f;
    if (i < n_kp) {
      const float2 p = kp[i];
      px = p.x; py = p.y;
      // node.cpp:931-937
      const bool bad = !TRUNC && (px >= (float)cols || px < 0.f || py >= (float)rows || py < 0.f ||
                                  __builtin_isnan(px) || __builtin_isnan(py));
      if (TRUNC) {
        // v_cvt_i32_f32 truncates, saturates and maps NaN to 0
        int r = (int)py, c = (int)px;
        r = min(max(r, 0), rows - 1);
        c = min(max(c, 0), cols - 1);
        // z_gathered: getMinDepthInNeighborhood's value under "use_feature_min_depth" (:731) instead of the pixel (:733)
        const float zraw = z_gathered ? z_gathered[i] : depth[(size_t)r * (size_t)cols + (size_t)c];
        Z = (float)((double)zraw * depth_scaling);
        keep = !__builtin_isnan(Z);  // :736
      } else if (!bad) {
        // depth.at<float>(round(y), round(x)): std::round = half away from zero (node.cpp:942).
        // round(y) can reach `rows` for y in [rows-0.5, rows): the reference reads out of
        // bounds there; clamp to the last row/column instead.
        int r = (int)roundf(py), c = (int)roundf(px);
        r = r >= rows ? rows - 1 : r;
        c = c >= cols ? cols - 1 : c;
        // z_gathered: the caller has looked depth(round(y), round(x)) up already (the image stays on the host)
        const float zraw = z_gathered ? z_gathered[i] : depth[(size_t)r * (size_t)cols + (size_t)c];
        Z = (float)((double)zraw * depth_scaling);
        keep = !__builtin_isnan(Z);  // node.cpp:947
      }
    }
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    const uint32_t chunk_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    const uint32_t pos = off + rank;
    if (keep && pos < (uint32_t)max_keypoints) {
      // backProject (misc2.h:62-64): ((u - cx) * z) * fxinv in float
      float4 o;
      o.x = (px - cx) * Z * fxinv;
      o.y = (py - cy) * Z * fyinv;
      o.z = Z;
      o.w = 1.0f;  // node.cpp:955
      xyz1[pos] = o;
      kept_idx[pos] = i;
    }
    base += chunk_total;
    __syncthreads();
    if (base >= (uint32_t)max_keypoints) break;  // node.cpp:957
  }
  if (tid == 0) *n_out = (int32_t)min(base, (uint32_t)max_keypoints);
}

// Node::projectTo3D, point-cloud overload (src/node.cpp:855-898; the ctor that is handed the sensor's organised cloud,
// :252-369): lookup point_cloud->at((int)x, (int)y) -- truncation (:877) --, drop when z > maximum_depth or a coordinate
// is NaN (:880), keep the cloud's own (x, y, z, 1) (:887), cut at max_keypoints (:889).  `pts` holds the looked-up cloud
// point of every keypoint (gathered by the caller: 16 bytes per keypoint cross PCIe instead of the 4.9 MB cloud), or the
// whole cloud when `gathered` is 0.  Same order-preserving compaction as project_to_3d_kernel.
__global__ __launch_bounds__(256) void project_cloud_kernel(const float2* __restrict__ kp, int n_kp,
                                                           const float4* __restrict__ pts, int gathered, int rows,
                                                           int cols, double maximum_depth, int max_keypoints,
                                                           int32_t* __restrict__ kept_idx, float4* __restrict__ xyz1,
                                                           int32_t* __restrict__ n_out) {
  __shared__ uint32_t wave_cnt[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  uint32_t base = 0;
  for (int c0 = 0; c0 < n_kp; c0 += 256) {
    const int i = c0 + tid;
    bool keep = false;
    float4 p3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n_kp) {
      const float2 p = kp[i];
      const bool bad = p.x >= (float)cols || p.x < 0.f || p.y >= (float)rows || p.y < 0.f || __builtin_isnan(p.x) ||
                       __builtin_isnan(p.y);  // :868-875
      if (!bad) {
        p3 = gathered ? pts[i] : pts[(size_t)(int)p.y * (size_t)cols + (size_t)(int)p.x];  // :877
        keep = !(((double)p3.z > maximum_depth) || __builtin_isnan(p3.x) || __builtin_isnan(p3.y) || __builtin_isnan(p3.z));
      }
    }
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    const uint32_t chunk_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    const uint32_t pos = off + rank;
    if (keep && pos < (uint32_t)max_keypoints) {
      xyz1[pos] = make_float4(p3.x, p3.y, p3.z, 1.0f);  // :887
      kept_idx[pos] = i;
    }
    base += chunk_total;
    __syncthreads();
    if (base >= (uint32_t)max_keypoints) break;  // :889
  }
  if (tid == 0) *n_out = (int32_t)min(base, (uint32_t)max_keypoints);
}

// One wave per kept keypoint: row y of `raw` (siftgpu_descriptors) and of `feat`
// (feature_descriptors_) <- descriptors_in[kept_idx[y]]; `feat` is RootSIFT-normalised when asked.
// Lane l holds columns 2l, 2l+1.  The L1 norm follows cv::reduce's float accumulation order
// (a0 over columns 0,2,..,124,126,127; a1 over 1,3,..,125; a0 + a1): a strictly sequential chain, fed
// by v_readlane so that the whole wave computes it uniformly.
__global__ __launch_bounds__(256) void sift_pack_kernel(const float2* __restrict__ in,
                                                       const int32_t* __restrict__ kept_idx,
                                                       const int32_t* __restrict__ n_ptr, int root_sift,
                                                       float2* __restrict__ raw, float2* __restrict__ feat) {
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y >= *n_ptr) return;
  const float2 v = in[(size_t)kept_idx[y] * 64 + lane];
  raw[(size_t)y * 64 + lane] = v;
  if (!feat) return;
  float2 o = v;
  if (root_sift) {
    o.x = fabsf(v.x);  // cv::abs (:1561)
    o.y = fabsf(v.y);
    auto lane_f = [](float f, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), l)); };
    float a0 = lane_f(o.x, 0), a1 = lane_f(o.y, 0);
#pragma unroll
    for (int l = 1; l < 63; ++l) {
      a0 = a0 + lane_f(o.x, l);
      a1 = a1 + lane_f(o.y, l);
    }
    a0 = a0 + lane_f(o.x, 63);
    a0 = a0 + lane_f(o.y, 63);
    const float sum = a0 + a1;
    if (sum != 0.0f) {  // :1565
      o.x = sqrtf(o.x / sum);  // :1569
      o.y = sqrtf(o.y / sum);
    }
  }
  feat[(size_t)y * 64 + lane] = o;
}

void launch_project_to_3d(const float* kp_xy, int n_kp, const float* depth, int rows, int cols,
                          float fxinv, float fyinv, float cx, float cy, double depth_scaling,
                          int max_keypoints, int32_t* kept_idx, float4* xyz1, int32_t* n_out,
                          hipStream_t stream, bool truncate, const float* z_gathered) {
  if (truncate)
    hipLaunchKernelGGL(project_to_3d_kernel<true>, dim3(1), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(kp_xy), n_kp, depth, rows, cols, fxinv, fyinv,
                       cx, cy, depth_scaling, max_keypoints, kept_idx, xyz1, n_out, z_gathered);
  else
    hipLaunchKernelGGL(project_to_3d_kernel<false>, dim3(1), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(kp_xy), n_kp, depth, rows, cols, fxinv, fyinv,
                       cx, cy, depth_scaling, max_keypoints, kept_idx, xyz1, n_out, z_gathered);
}

// Several frames in one launch (the super-frame batch path of rgbdfe_detect_describe_batch): block f projects the keypoints
// [off[f], off[f] + n[f]) of the concatenated buffers -- a frame's input block holds 2 n (x, y) floats followed by its n
// depths depth.at<float>(round(y), round(x)), looked up by the caller.  project_to_3d_kernel<false>'s arithmetic and
// order-preserving compaction (node.cpp:931-957, misc2.h:62-64).
__global__ __launch_bounds__(256) void project_to_3d_frames_kernel(ProjectFrames fr, const float* __restrict__ kpxy, int rows,
                                                                   int cols, float fxinv, float fyinv, float cx, float cy,
                                                                   double depth_scaling, int max_keypoints,
                                                                   int32_t* __restrict__ kept_all, float4* __restrict__ xyz_all,
                                                                   int32_t* __restrict__ n_out) {
  __shared__ uint32_t wave_cnt[4];
  const int f = blockIdx.x;
  const int n_kp = fr.n[f];
  const float* __restrict__ in = kpxy + (size_t)3 * fr.off[f];
  const float* __restrict__ zg = in + (size_t)2 * n_kp;
  int32_t* __restrict__ kept_idx = kept_all + fr.off[f];
  float4* __restrict__ xyz1 = xyz_all + fr.off[f];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  uint32_t base = 0;
  for (int c0 = 0; c0 < n_kp; c0 += 256) {
    const int i = c0 + tid;
    bool keep = false;
    float px = 0.f, py = 0.f, Z = 0.f;
    if (i < n_kp) {
      px = in[2 * i]; py = in[2 * i + 1];
      const bool bad = px >= (float)cols || px < 0.f || py >= (float)rows || py < 0.f || __builtin_isnan(px) ||
                       __builtin_isnan(py);
      if (!bad) {
        Z = (float)((double)zg[i] * depth_scaling);
        keep = !__builtin_isnan(Z);
      }
    }
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    const uint32_t chunk_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    const uint32_t pos = off + rank;
    if (keep && pos < (uint32_t)max_keypoints) {
      float4 o;
      o.x = (px - cx) * Z * fxinv;
      o.y = (py - cy) * Z * fyinv;
      o.z = Z;
      o.w = 1.0f;
      xyz1[pos] = o;
      kept_idx[pos] = i;
    }
    base += chunk_total;
    __syncthreads();
    if (base >= (uint32_t)max_keypoints) break;
  }
  if (tid == 0) n_out[f] = (int32_t)min(base, (uint32_t)max_keypoints);
}

void launch_project_to_3d_frames(const ProjectFrames& fr, const float* kpxy, int rows, int cols, float fxinv, float fyinv,
                                 float cx, float cy, double depth_scaling, int max_keypoints, int32_t* kept_idx, float4* xyz1,
                                 int32_t* n_out, hipStream_t stream) {
  if (fr.n_frames > 0)
    hipLaunchKernelGGL(project_to_3d_frames_kernel, dim3(fr.n_frames), dim3(256), 0, stream, fr, kpxy, rows, cols, fxinv,
                       fyinv, cx, cy, depth_scaling, max_keypoints, kept_idx, xyz1, n_out);
}

// getMinDepthInNeighborhood (misc.cpp:774-793), the depth lookup of "use_feature_min_depth" (parameter_server.cpp:90):
// the smallest non-NaN depth of the keypoint's neighbourhood -- rows [int(y - r), int(y + r)) x cols [int(x - r),
// int(x + r)), r = int((size - 1) / 2), clamped to the image; no comparable value or a minimum of 0 gives NaN.
// One wave per keypoint, lanes along the window's columns; a minimum is order independent: exact.
__global__ __launch_bounds__(64) void min_depth_kernel(const float* __restrict__ kp, int n_kp, const float* __restrict__ depth,
                                                       int rows, int cols, float* __restrict__ z_out) {
  const int i = blockIdx.x;
  if (i >= n_kp) return;
  const float cx = kp[3 * i], cy = kp[3 * i + 1], diameter = kp[3 * i + 2];
  const int radius = (int)((diameter - 1) / 2);
  int top = (int)(cy - (float)radius); top = top < 0 ? 0 : top;
  int left = (int)(cx - (float)radius); left = left < 0 ? 0 : left;
  int bot = (int)(cy + (float)radius); bot = bot > rows ? rows : bot;
  int right = (int)(cx + (float)radius); right = right > cols ? cols : right;
  float mn = 3.402823466e+38f;
  bool found = false;
  for (int r = top; r < bot; ++r)
    for (int c = left + (int)threadIdx.x; c < right; c += 64) {
      const float v = depth[(size_t)r * (size_t)cols + (size_t)c];
      if (v < mn) { mn = v; found = true; }  // NaN never compares less: skipped, as in cv::minMaxLoc
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mn = fminf(mn, __shfl_xor(mn, d));  // no NaN among the partial minima
  found = __ballot(found) != 0ull;
  if (threadIdx.x == 0) z_out[i] = (found && mn != 0.0f) ? mn : __builtin_nanf("");
}

// kp: n_kp x (x, y, size); z_out[i] = getMinDepthInNeighborhood(depth, (x, y), size)
void launch_min_depth(const float* kp_xys, int n_kp, const float* depth, int rows, int cols, float* z_out,
                      hipStream_t stream) {
  if (n_kp > 0) hipLaunchKernelGGL(min_depth_kernel, dim3(n_kp), dim3(64), 0, stream, kp_xys, n_kp, depth, rows, cols, z_out);
}

void launch_project_cloud(const float* kp_xy, int n_kp, const float4* pts, bool gathered, int rows, int cols,
                          double maximum_depth, int max_keypoints, int32_t* kept_idx, float4* xyz1, int32_t* n_out,
                          hipStream_t stream) {
  hipLaunchKernelGGL(project_cloud_kernel, dim3(1), dim3(256), 0, stream, reinterpret_cast<const float2*>(kp_xy), n_kp,
                     pts, gathered ? 1 : 0, rows, cols, maximum_depth, max_keypoints, kept_idx, xyz1, n_out);
}

void launch_sift_pack(const float* desc_in, const int32_t* kept_idx, const int32_t* n_ptr, int max_rows,
                      bool root_sift, float* raw, float* feat, hipStream_t stream) {
  if (max_rows <= 0) return;
  hipLaunchKernelGGL(sift_pack_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, stream,
                     reinterpret_cast<const float2*>(desc_in), kept_idx, n_ptr, root_sift ? 1 : 0,
                     reinterpret_cast<float2*>(raw), reinterpret_cast<float2*>(feat));
}

}  // namespace rgbdfe
