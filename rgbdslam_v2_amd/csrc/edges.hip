// edges.hip -- stream compaction of the accepted edges of a shard (an all-pairs loop-closure sweep rejects
// most pairs, id1 == -1 (node.cpp:1419), and those records need not cross xGMI).  Stable: the surviving records keep
// their shard order, so every device -- and every run -- sees the same list.
#include <cstddef>

#include "rgbdfe_internal.h"

namespace rgbdfe {

namespace {

// one block: flags -> exclusive positions (dst[k] = position of record k among the accepted ones, or -1), total -> *count
__global__ __launch_bounds__(1024) void edge_scan_kernel(const rgbdfe_match_result* __restrict__ in, uint32_t n,
                                                         int32_t* __restrict__ dst, int32_t* __restrict__ count) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += 1024u) {
    const uint32_t k = k0 + tid;
    const bool keep = k < n && in[k].id1 >= 0;
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_tot[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (uint32_t w = 0; w < wv; ++w) off += wave_tot[w];
    if (k < n) dst[k] = keep ? (int32_t)(off + rank) : -1;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
      for (int w = 0; w < 16; ++w) t += wave_tot[w];
      base += t;
    }
    __syncthreads();
  }
  if (tid == 0) *count = (int32_t)base;
}

// one wave per record: 1744 bytes = 109 x 16 bytes
__global__ __launch_bounds__(64) void edge_copy_kernel(const rgbdfe_match_result* __restrict__ in,
                                                       const int32_t* __restrict__ dst, uint32_t n,
                                                       rgbdfe_match_result* __restrict__ out,
                                                       int32_t* __restrict__ out_index, int32_t index_scale,
                                                       int32_t index_offset) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  const int32_t d = dst[k];
  if (d < 0) return;
  static_assert(sizeof(rgbdfe_match_result) % 16 == 0, "records move as 16-byte chunks");
  constexpr uint32_t chunks = sizeof(rgbdfe_match_result) / 16;
  const uint4* __restrict__ s = reinterpret_cast<const uint4*>(in + k);
  uint4* __restrict__ o = reinterpret_cast<uint4*>(out + d);
  for (uint32_t c = threadIdx.x; c < chunks; c += 64u) o[c] = s[c];
  if (out_index && threadIdx.x == 0) out_index[d] = index_offset + index_scale * (int32_t)k;  // position in the caller's list
}

// rgbdfe_compact_result = a record without its all_q / all_t / all_hd lists: 13 + 5 words of 8 bytes out of 218
__global__ __launch_bounds__(256) void compact_pack_kernel(const uint64_t* __restrict__ in, uint32_t n,
                                                           uint64_t* __restrict__ out) {
  static_assert(sizeof(rgbdfe_match_result) == 218 * 8 && sizeof(rgbdfe_compact_result) == 18 * 8, "compact record layout");
  static_assert(offsetof(rgbdfe_match_result, all_q) == 13 * 8 && offsetof(rgbdfe_match_result, inlier_mask) == 213 * 8,
                "compact record layout");
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n * 18u) return;
  const uint32_t k = i / 18u, w = i - k * 18u;
  out[i] = in[(size_t)k * 218u + (w < 13u ? w : 200u + w)];
}

// ---- the inlier stream of a shard (rgbdfe_inlier_header): headers + one list of (query row, train row) per pair ----------
constexpr uint32_t kHeaderWords = sizeof(rgbdfe_inlier_header) / 4;                      // 26
constexpr uint32_t kFirstInlierWord = offsetof(rgbdfe_inlier_header, first_inlier) / 4;  // 21
static_assert(sizeof(rgbdfe_inlier_header) == offsetof(rgbdfe_match_result, all_q), "an inlier header = a record's leading fields");
static_assert(offsetof(rgbdfe_inlier_header, first_inlier) == offsetof(rgbdfe_match_result, pad0), "first_inlier sits in pad0");

// one block: inlier counts -> exclusive positions (headers' first_inlier), the padding headers, the total
__global__ __launch_bounds__(1024) void inlier_scan_kernel(const rgbdfe_match_result* __restrict__ in, uint32_t n,
                                                           uint32_t n_headers, rgbdfe_inlier_header* __restrict__ hdr,
                                                           int32_t* __restrict__ total) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += 1024u) {
    const uint32_t k = k0 + tid;
    uint32_t c = 0;
    if (k < n)
      for (int w = 0; w < RGBDFE_MASK_WORDS; ++w) c += (uint32_t)__popcll(in[k].inlier_mask[w]);
    uint32_t incl = c;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t off = base;
    for (uint32_t w = 0; w < wv; ++w) off += wave_tot[w];
    if (k < n) hdr[k].first_inlier = off + incl - c;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
      for (int w = 0; w < 16; ++w) t += wave_tot[w];
      base += t;
    }
    __syncthreads();
  }
  for (uint32_t k = n + tid; k < n_headers; k += 1024u) {   // padding headers: "no edge", no inliers
    uint32_t* __restrict__ h = reinterpret_cast<uint32_t*>(hdr + k);
    for (uint32_t w = 0; w < kHeaderWords; ++w) h[w] = 0u;
    hdr[k].id1 = -1; hdr[k].id2 = -1; hdr[k].first_inlier = base;
  }
  if (tid == 0) *total = (int32_t)base;
}

// one wave per pair: the header (its first_inlier is already there) and the pair's list
__global__ __launch_bounds__(64) void inlier_list_kernel(const rgbdfe_match_result* __restrict__ in, uint32_t n,
                                                         rgbdfe_inlier_header* __restrict__ hdr, uint32_t* __restrict__ list) {
  const uint32_t k = blockIdx.x, lane = threadIdx.x;
  if (k >= n) return;
  const rgbdfe_match_result& r = in[k];
  if (lane < kHeaderWords && lane != kFirstInlierWord)
    reinterpret_cast<uint32_t*>(hdr + k)[lane] = reinterpret_cast<const uint32_t*>(&r)[lane];
  uint32_t at = hdr[k].first_inlier;
#pragma unroll
  for (int w = 0; w < RGBDFE_MASK_WORDS; ++w) {
    const uint64_t m = r.inlier_mask[w];
    if ((m >> lane) & 1ull) {
      const uint32_t i = (uint32_t)w * 64u + lane;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      list[at + rank] = (uint32_t)r.all_q[i] | ((uint32_t)r.all_t[i] << 16);
    }
    at += (uint32_t)__popcll(m);
  }
}

}  // namespace

void launch_pack_inliers(const rgbdfe_match_result* in, uint32_t n, uint32_t n_headers, void* stream_out, int32_t* d_total,
                         hipStream_t stream) {
  rgbdfe_inlier_header* hdr = reinterpret_cast<rgbdfe_inlier_header*>(stream_out);
  uint32_t* list = reinterpret_cast<uint32_t*>(hdr + n_headers);
  hipLaunchKernelGGL(inlier_scan_kernel, dim3(1), dim3(1024), 0, stream, in, n, n_headers, hdr, d_total);
  if (n > 0) hipLaunchKernelGGL(inlier_list_kernel, dim3(n), dim3(64), 0, stream, in, n, hdr, list);
}

void launch_compact_pack(const rgbdfe_match_result* in, uint32_t n, rgbdfe_compact_result* out, hipStream_t stream) {
  if (n > 0)
    hipLaunchKernelGGL(compact_pack_kernel, dim3((n * 18u + 255u) / 256u), dim3(256), 0, stream,
                       reinterpret_cast<const uint64_t*>(in), n, reinterpret_cast<uint64_t*>(out));
}

// in[0..n) -> out[0..count): the records with id1 >= 0 in order; out_index[j] (optional) = index_offset + index_scale * k of
// the j-th survivor (the caller's global pair index of a shard: offset = device, scale = devices).
void launch_compact_edges(const rgbdfe_match_result* in, uint32_t n, rgbdfe_match_result* out, int32_t* out_index,
                          int32_t index_scale, int32_t index_offset, int32_t* d_dst, int32_t* d_count, hipStream_t stream) {
  hipLaunchKernelGGL(edge_scan_kernel, dim3(1), dim3(1024), 0, stream, in, n, d_dst, d_count);
  if (n > 0)
    hipLaunchKernelGGL(edge_copy_kernel, dim3(n), dim3(64), 0, stream, in, d_dst, n, out, out_index, index_scale,
                       index_offset);
}

}  // namespace rgbdfe
