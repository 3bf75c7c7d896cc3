// emm.hip -- the frame-level data either side of the pair path, gfx950.
//
//   depth_to_mono8_*      depthToCV8UC1            (src/misc.cpp:414-430)  detection mask from depth
//   create_cloud_kernel   createXYZRGBPointCloud   (src/misc.cpp:467-556)  structured cloud of a frame
//   emm_kernel            observationLikelihood    (src/misc.cpp:814-969)  environment measurement model,
//                         the check matchNodePair applies to a RANSAC edge when observability_threshold > 0
//                         (src/node.cpp:1340-1343, pairwiseObservationLikelihood :1520-1554)
//
// All three are per-pixel streaming / gather kernels: lane = pixel (or sampled point), no LDS tiles.  The
// float / double operation order is the oracle's (oracle/rgbd_oracle.c), compiled with -ffp-contract=off:
// results are bit-identical.  The EMM's two cdf tests compare the depth difference against boundaries found
// on the host (bisection with the host's libm erf, then the exact pre-image under the IEEE division by
// sigma * SQRT_2; api_frame.hip), so neither a device erf nor a division enters the decision.
#include "rgbdfe_internal.h"

namespace rgbdfe {

// saturate_cast<uchar>(cvRound(t)): round half to even; NaN / out-of-int-range -> 0 (cvtss2si indefinite)
__device__ __forceinline__ uint8_t sat_u8_rne(float t) {
  if (!(t > -2147483648.0f && t < 2147483648.0f)) return 0;
  const int r = (int)rintf(t);
  return (uint8_t)min(max(r, 0), 255);
}

__global__ __launch_bounds__(256) void depth_to_mono8_f32_kernel(const float* __restrict__ depth, size_t n,
                                                                uint8_t* __restrict__ mono8) {
  // 4 pixels per lane: one 16-byte load, one 4-byte store
  const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const float4 d = *reinterpret_cast<const float4*>(depth + i4);
    const uint32_t o = (uint32_t)sat_u8_rne(d.x * 100.0f + 0.0f) | ((uint32_t)sat_u8_rne(d.y * 100.0f + 0.0f) << 8) |
                       ((uint32_t)sat_u8_rne(d.z * 100.0f + 0.0f) << 16) | ((uint32_t)sat_u8_rne(d.w * 100.0f + 0.0f) << 24);
    *reinterpret_cast<uint32_t*>(mono8 + i4) = o;
  } else {
    for (size_t i = i4; i < n; ++i) mono8[i] = sat_u8_rne(depth[i] * 100.0f + 0.0f);
  }
}

__global__ __launch_bounds__(256) void depth_u16_kernel(const uint16_t* __restrict__ depth_mm, size_t n,
                                                       uint8_t* __restrict__ mono8, float* __restrict__ depth_m) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (float)depth_mm[i];
  mono8[i] = sat_u8_rne(v * 0.05f + -25.0f);  // misc.cpp:423
  depth_m[i] = v * 0.001f + 0.0f;             // misc.cpp:424
}

// One lane per cloud point (vi, ui) <- depth pixel (vi*s, ui*s); the reference's running color_idx /
// depth_idx equal channels*(v*cols+u) and v*cols+u when s divides both image dimensions (checked by the host).
__global__ __launch_bounds__(256) void create_cloud_kernel(
    const float* __restrict__ depth, int rows, int cols, const uint8_t* __restrict__ rgb, int channels,
    int encoding_bgr, float fxinv, float fyinv, float cx, float cy, double depth_scaling, float min_depth,
    int s, int ch, int cw, float4* __restrict__ cloud, float* __restrict__ zplane) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ch * cw) return;
  const int vi = k / cw, ui = k - vi * cw;
  const int v = vi * s, u = ui * s;
  const size_t pix = (size_t)v * (size_t)cols + (size_t)u;
  const float Z = (float)((double)depth[pix] * depth_scaling);  // misc.cpp:522
  float4 pt;
  if (!(Z >= min_depth)) {  // :525 (also NaN)
    pt.x = (float)((double)((float)u - cx) * 1.0 * (double)fxinv);  // :527
    pt.y = (float)((double)((float)v - cy) * 1.0 * (double)fyinv);
    pt.z = __builtin_nanf("");
  } else {  // backProject (misc2.h:62-64)
    pt.x = ((float)u - cx) * Z * fxinv;
    pt.y = ((float)v - cy) * Z * fyinv;
    pt.z = Z;
  }
  uint32_t bits = 0u;
  if (rgb && k > 0) {  // `color_idx > 0` (:536): the first point's colour is never written
    const size_t ci = pix * (size_t)channels;
    uint32_t r, g, b;
    if (channels == 3) {
      r = rgb[ci + (encoding_bgr ? 2 : 0)];
      g = rgb[ci + 1];
      b = rgb[ci + (encoding_bgr ? 0 : 2)];
    } else {
      r = g = b = rgb[ci];
    }
    bits = b | (g << 8) | (r << 16);  // RGBValue {Blue, Green, Red, Alpha = 0}
  }
  pt.w = __uint_as_float(bits);
  cloud[k] = pt;
  zplane[k] = pt.z;  // dense depth plane for the EMM's neighbourhood reads (4 B instead of 16 B per point)
}

// The points observationLikelihood visits (every skip-th row and column, misc.cpp:882-883) copied into a dense
// array: the EMM then reads 16 contiguous bytes per lane instead of 16 bytes per 128-byte line.  Built once per
// (node, skip step) and cached next to the cloud.
__global__ __launch_bounds__(256) void decimate_cloud_kernel(const float4* __restrict__ cloud, int cw, int skip_step,
                                                            int nsx, int total, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int sy = i / nsx, sx = i - sy * nsx;
  out[i] = cloud[(size_t)(sy * skip_step) * cw + sx * skip_step];
}

// One 256-lane block per job (= one direction of one edge), lane = sampled point of the new cloud.
__global__ __launch_bounds__(256) void emm_kernel(const EmmJob* __restrict__ jobs, int ch, int cw, int skip_step,
                                                 double d_lo, double d_hi, uint32_t* __restrict__ counts) {
  __shared__ uint32_t acc[3];
  const EmmJob jb = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid < 3) acc[tid] = 0u;
  __syncthreads();
  const float4* __restrict__ new_pc = jb.new_samples;  // the sampled points, contiguous (decimate_cloud_kernel)
  const float* __restrict__ old_z = jb.old_z;
  const int nsx = (cw + skip_step - 1) / skip_step, nsy = (ch + skip_step - 1) / skip_step;
  const int total = nsx * nsy;
  uint32_t good = 0, bad = 0, occ = 0;
  for (int i = tid; i < total; i += 256) {
    const float4 q = new_pc[i];
    float px = q.x, py = q.y, pz = q.z;
    if (isfinite(px) && isfinite(py) && isfinite(pz)) {  // pcl::transformPointCloud, non-dense cloud
      const float x = px, y = py, z = pz;
      px = jb.T[0] * x + jb.T[1] * y + jb.T[2] * z + jb.T[3];
      py = jb.T[4] * x + jb.T[5] * y + jb.T[6] * z + jb.T[7];
      pz = jb.T[8] * x + jb.T[9] * y + jb.T[10] * z + jb.T[11];
    }
    if (pz != pz) continue;   // misc.cpp:886
    if (pz < 0.0f) continue;  // :887
    const double dx = floor((double)((px / pz) * jb.fx + jb.cx) + 0.5);  // round(), :804-807
    const double dy = floor((double)((py / pz) * jb.fy + jb.cy) + 0.5);
    if (!(dx >= 0.0 && dx < (double)cw && dy >= 0.0 && dy < (double)ch)) continue;  // :891-896
    const int xc = (int)dx, yc = (int)dy;
    const int startx = max(0, xc - 2), starty = max(0, yc - 2);
    const int endx = min(cw, xc + 3), endy = min(ch, yc + 3);
    // The 3 x 3 samples of the 5 x 5 neighbourhood (step 2, misc.cpp:902-908): all loads are issued before
    // the first use; the three flags are order-independent ORs.
    const double pzd = (double)pz;
    float oz[9];
    bool in[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int oy = starty + 2 * j, ox = startx + 2 * k;
        in[j * 3 + k] = (oy < endy) && (ox < endx);
        oz[j * 3 + k] = old_z[(size_t)min(oy, ch - 1) * cw + min(ox, cw - 1)];
      }
    bool good_point = false, occluded_point = false, bad_point = false;
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      const float z = oz[n];
      if (!in[n] || z != z) continue;  // :911
      // cdf tests (:924-938) on d = old_z - new_z: p < 0.001 <=> d / denom < q_lo <=> d < d_lo (the division
      // by a positive constant is monotone; d_lo, d_hi are the exact boundaries found on the host)
      const double d = (double)z - pzd;
      if (d < d_lo) occluded_point = true;
      else if (d < d_hi) good_point = true;
      else bad_point = true;
    }
    if (good_point) good++;
    else if (occluded_point) occ++;
    else if (bad_point) bad++;
  }
  // wave reduction, then 3 LDS atomics per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    good += __shfl_xor(good, off);
    bad += __shfl_xor(bad, off);
    occ += __shfl_xor(occ, off);
  }
  if ((tid & 63) == 0) {
    atomicAdd(&acc[0], good);
    atomicAdd(&acc[1], bad);
    atomicAdd(&acc[2], occ);
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t* o = counts + (size_t)blockIdx.x * 4;
    o[0] = acc[0];  // inliers
    o[1] = acc[1];  // outliers
    o[2] = acc[2];  // occluded
    o[3] = (uint32_t)total;  // `all` counts every sampled raster position (:883)
  }
}

void launch_depth_to_mono8_f32(const float* depth, size_t n, uint8_t* mono8, hipStream_t stream) {
  if (n == 0) return;
  const size_t lanes = (n + 3) / 4;
  hipLaunchKernelGGL(depth_to_mono8_f32_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream,
                     depth, n, mono8);
}
void launch_depth_u16(const uint16_t* depth_mm, size_t n, uint8_t* mono8, float* depth_m, hipStream_t stream) {
  if (n == 0) return;
  hipLaunchKernelGGL(depth_u16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, depth_mm, n,
                     mono8, depth_m);
}
void launch_create_cloud(const float* depth, int rows, int cols, const uint8_t* rgb, int channels,
                         int encoding_bgr, float fxinv, float fyinv, float cx, float cy, double depth_scaling,
                         float min_depth, int s, int ch, int cw, float4* cloud, float* zplane, hipStream_t stream) {
  const int n = ch * cw;
  if (n <= 0) return;
  hipLaunchKernelGGL(create_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, depth, rows, cols, rgb,
                     channels, encoding_bgr, fxinv, fyinv, cx, cy, depth_scaling, min_depth, s, ch, cw, cloud, zplane);
}
void launch_decimate_cloud(const float4* cloud, int ch, int cw, int skip_step, float4* out, hipStream_t stream) {
  const int nsx = (cw + skip_step - 1) / skip_step, nsy = (ch + skip_step - 1) / skip_step;
  const int total = nsx * nsy;
  if (total <= 0) return;
  hipLaunchKernelGGL(decimate_cloud_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, cloud, cw, skip_step, nsx,
                     total, out);
}
void launch_emm(const EmmJob* jobs, int n_jobs, int ch, int cw, int skip_step, double d_lo, double d_hi,
                uint32_t* counts, hipStream_t stream) {
  if (n_jobs <= 0) return;
  hipLaunchKernelGGL(emm_kernel, dim3(n_jobs), dim3(256), 0, stream, jobs, ch, cw, skip_step, d_lo, d_hi, counts);
}

}  // namespace rgbdfe
