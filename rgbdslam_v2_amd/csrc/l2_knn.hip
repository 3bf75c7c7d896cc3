// l2_knn.hip -- Node::featureMatching's FLANN branch for float descriptors (src/node.cpp:610-667) with EXACT neighbours.
//
// The reference builds 4 randomised kd-trees over the older node's descriptors (getFlannIndex, node.cpp:493-505) and asks
// for the 2 nearest neighbours of every descriptor of the newer node with 16 checks (:634) -- an approximate search that
// cannot be reproduced bit for bit (tree construction is randomised).  What the branch does WITH the neighbours is
// reproduced exactly: ratio = dists[2i] / dists[2i+1] over FLANN's squared-L2 distances (:645), accepted when
// nn_distance_ratio > ratio (:648), every train index used once, first come first served in query order (:650-653),
// DMatch.distance = the ratio (:657).  The neighbours themselves are the exact 2 nearest (a superset-quality
// replacement): the squared distance is accumulated in the order of flann::L2<float>::operator()
// (four differences per step: result += ((d0*d0 + d1*d1) + d2*d2) + d3*d3), strict <, so the lowest train row wins ties.
//
//   l2_knn2_kernel     lane = query descriptor (its 128 floats live in VGPRs), the train row is wave-uniform: scalar
//                      loads through the constant cache, consumed as the SGPR operand of v_sub_f32 -- the structure of
//                      hamming_nn_kernel with 12 float VALU instructions per 4 dimensions instead of xor + popcount;
//                      no LDS, no vector-memory instruction in the loop;
//   l2_ratio_kernel    one block per pair: ratio test, the train-unique rule as an atomicMin of the claiming query per
//                      train row (= first come first served in query order), compaction in query order.
// The match list goes the way of the SIFTGPU matcher's (sift_sort_kernel -> pair_prep_kernel<true> -> RANSAC).
#include "rgbdfe_internal.h"

namespace rgbdfe {

namespace {

constexpr int kDim = 128;       // descriptor slab width (64-d descriptors are zero padded: adds +0.0 terms)
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void l2_knn2_kernel(const float* __restrict__ f32_pool,
                                                           const PairWork* __restrict__ work, uint32_t max_kp,
                                                           uint32_t n_pairs, uint32_t tiles,
                                                           uint32_t* __restrict__ knn /* [pair][row][3] */) {
  const uint32_t L = blockIdx.x;
  const uint32_t xcd = L & 7u, j = L >> 3;
  const uint32_t pair = (j / tiles) * 8u + xcd;  // whole pairs per XCD (block b runs on XCD b % 8)
  if (pair >= n_pairs) return;
  const uint32_t tile = j % tiles;
  const PairWork w = work[pair];
  const uint32_t nq = w.nq, nt = w.nt;
  if (tile * kThreads >= nq) return;
  uint32_t qi = tile * kThreads + threadIdx.x;
  const bool live = qi < nq;
  qi = live ? qi : nq - 1u;
  float q[kDim];
  {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(f32_pool + ((size_t)w.q_slot * max_kp + qi) * kDim);
#pragma unroll
    for (int k = 0; k < kDim / 4; ++k) {
      const float4 v = src[k];
      q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w;
    }
  }
  float b1 = __builtin_inff(), b2 = __builtin_inff();
  uint32_t i1 = 0xFFFFFFFFu;
  const float* __restrict__ tp = f32_pool + (size_t)w.t_slot * max_kp * kDim;
  for (uint32_t t = 0; t < nt; ++t) {
    const float* __restrict__ row = tp + (size_t)t * kDim;  // wave-uniform: s_load
    float result = 0.0f;
#pragma unroll
    for (int k = 0; k < kDim; k += 4) {  // flann::L2<float>::operator(), four differences per step
      const float d0 = q[k] - row[k], d1 = q[k + 1] - row[k + 1], d2 = q[k + 2] - row[k + 2], d3 = q[k + 3] - row[k + 3];
      result += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
    }
    if (result < b1) { b2 = b1; b1 = result; i1 = t; }
    else if (result < b2) b2 = result;
  }
  if (live) {
    uint32_t* o = knn + ((size_t)pair * max_kp + qi) * 3u;
    o[0] = __float_as_uint(b1);
    o[1] = __float_as_uint(b2);
    o[2] = i1;
  }
}

__global__ __launch_bounds__(kThreads) void l2_ratio_kernel(const PairWork* __restrict__ work, uint32_t max_kp,
                                                            const uint32_t* __restrict__ knn,
                                                            uint32_t* __restrict__ claim /* [pair][row] */,
                                                            double max_ratio, uint16_t* __restrict__ sm_q,
                                                            uint16_t* __restrict__ sm_t, float* __restrict__ sm_d,
                                                            int32_t* __restrict__ sm_n) {
  __shared__ uint32_t wave_cnt[4];
  const uint32_t pair = blockIdx.x;
  const PairWork w = work[pair];
  const int nq = (int)w.nq, nt = (int)w.nt;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint16_t* __restrict__ oq = sm_q + (size_t)pair * max_kp;
  uint16_t* __restrict__ ot = sm_t + (size_t)pair * max_kp;
  float* __restrict__ od = sm_d + (size_t)pair * max_kp;
  if (nq <= 0 || nt < 2) {  // knnSearch with k = 2 needs two train rows
    if (tid == 0) sm_n[pair] = 0;
    return;
  }
  const uint32_t* __restrict__ kb = knn + (size_t)pair * max_kp * 3u;
  uint32_t* __restrict__ cl = claim + (size_t)pair * max_kp;
  for (int t = tid; t < nt; t += kThreads) cl[t] = 0xFFFFFFFFu;
  __syncthreads();
  // the query that claims a train row is the first one (in query order) that passes the ratio test with it (:650-653)
  for (int i = tid; i < nq; i += kThreads) {
    const float ratio = __uint_as_float(kb[(size_t)i * 3]) / __uint_as_float(kb[(size_t)i * 3 + 1]);  // :645
    if (max_ratio > (double)ratio) atomicMin(&cl[kb[(size_t)i * 3 + 2]], (uint32_t)i);                 // :648
  }
  __syncthreads();
  uint32_t base = 0;
  for (int i0 = 0; i0 < nq; i0 += kThreads) {
    const int i = i0 + tid;
    bool keep = false;
    float ratio = 0.f;
    uint32_t tr = 0;
    if (i < nq) {
      ratio = __uint_as_float(kb[(size_t)i * 3]) / __uint_as_float(kb[(size_t)i * 3 + 1]);
      tr = kb[(size_t)i * 3 + 2];
      keep = (max_ratio > (double)ratio) && cl[tr] == (uint32_t)i;
    }
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    if (keep) {
      oq[off + rank] = (uint16_t)i;
      ot[off + rank] = (uint16_t)tr;
      od[off + rank] = ratio;  // match.distance = dist_ratio_fac (:657)
    }
    base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) sm_n[pair] = (int32_t)base;
}

}  // namespace

void launch_l2_knn2(const float* f32_pool, const PairWork* work, uint32_t max_kp, uint32_t n_pairs, uint32_t max_nq,
                    uint32_t* knn, hipStream_t stream) {
  if (n_pairs == 0 || max_nq == 0) return;
  const uint32_t tiles = (max_nq + kThreads - 1) / kThreads;
  const uint32_t pairs8 = (n_pairs + 7u) / 8u * 8u;
  hipLaunchKernelGGL(l2_knn2_kernel, dim3(pairs8 * tiles), dim3(kThreads), 0, stream, f32_pool, work, max_kp, n_pairs,
                     tiles, knn);
}

void launch_l2_ratio(const PairWork* work, uint32_t max_kp, uint32_t n_pairs, const uint32_t* knn, uint32_t* claim,
                     double max_ratio, uint16_t* sm_q, uint16_t* sm_t, float* sm_d, int32_t* sm_n, hipStream_t stream) {
  if (n_pairs == 0) return;
  hipLaunchKernelGGL(l2_ratio_kernel, dim3(n_pairs), dim3(kThreads), 0, stream, work, max_kp, knn, claim, max_ratio,
                     sm_q, sm_t, sm_d, sm_n);
}

}  // namespace rgbdfe
