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
#include "rgbdfe_internal.h"

namespace rgbdfe {

__global__ __launch_bounds__(256) void project_to_3d_kernel(
    const float2* __restrict__ kp, int n_kp, const float* __restrict__ depth, int rows, int cols,
    float fxinv, float fyinv, float cx, float cy, double depth_scaling, int max_keypoints,
    int32_t* __restrict__ kept_idx, float4* __restrict__ xyz1, int32_t* __restrict__ n_out) {
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
      const bool bad = px >= (float)cols || px < 0.f || py >= (float)rows || py < 0.f ||
                       __builtin_isnan(px) || __builtin_isnan(py);
      if (!bad) {
        // depth.at<float>(round(y), round(x)): std::round = half away from zero (node.cpp:942).
        // round(y) can reach `rows` for y in [rows-0.5, rows): the reference reads out of
        // bounds there; clamp to the last row/column instead.
        int r = (int)roundf(py), c = (int)roundf(px);
        r = r >= rows ? rows - 1 : r;
        c = c >= cols ? cols - 1 : c;
        Z = (float)((double)depth[(size_t)r * (size_t)cols + (size_t)c] * depth_scaling);
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

void launch_project_to_3d(const float* kp_xy, int n_kp, const float* depth, int rows, int cols,
                          float fxinv, float fyinv, float cx, float cy, double depth_scaling,
                          int max_keypoints, int32_t* kept_idx, float4* xyz1, int32_t* n_out,
                          hipStream_t stream) {
  hipLaunchKernelGGL(project_to_3d_kernel, dim3(1), dim3(256), 0, stream,
                     reinterpret_cast<const float2*>(kp_xy), n_kp, depth, rows, cols, fxinv, fyinv,
                     cx, cy, depth_scaling, max_keypoints, kept_idx, xyz1, n_out);
}

}  // namespace rgbdfe
