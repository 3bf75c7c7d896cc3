// place_recognition.hip -- descriptor-vote prefilter for loop-closure candidates (GPU half).
//
// What the reference attempted in loop_closing.cpp (GraphManager::getNeighbours, :190-277, compiled only with
// DO_LOOP_CLOSING and never wired into nodeComparisons): every descriptor of the new node votes for the nodes that hold
// its nearest descriptors, "linear decreasing score" value = neighbour_cnt - rank (:241), the votes of a node are
// normalised by its descriptor count (:263) and the nodes are ranked by that score (:269).  There the neighbours come
// from an approximate kd-tree over float descriptors; here they are EXACT and binary: the Hamming kernel has just
// produced, for every (query descriptor, candidate node), the candidate's best match (hd, row) -- the keys of the
// pair path -- and a query descriptor's k nearest NODES are the k candidates with the smallest such hd (ties: the
// candidate listed first).  Votes are integers, so the sums are exact and order-independent (atomicAdd on u32).
//
// A batch holds many query nodes at once (an offline loop-closure sweep: every frame against all earlier frames is ONE
// Hamming launch + ONE vote launch); the online call shape is the batch of one query.
// One thread per query descriptor walks the candidates (the keys of neighbouring threads are neighbouring words, so a
// wave reads 256 contiguous bytes per candidate) and keeps its K best in registers.
#include "rgbdfe_internal.h"

namespace rgbdfe {

namespace {

constexpr int kMaxK = 8;

template <int K>
__global__ __launch_bounds__(256) void place_vote_kernel(const uint32_t* __restrict__ keys, uint32_t planes,
                                                         uint32_t max_kp, const PairWork* __restrict__ work,
                                                         const uint32_t* __restrict__ seg /* [n_queries + 1] */,
                                                         uint32_t k_use, uint32_t max_hd, uint32_t* __restrict__ votes) {
  // blockIdx.y = query node s: its candidates are the pairs seg[s] .. seg[s+1]-1 of the batch (all with the same query)
  const uint32_t s = blockIdx.y;
  const uint32_t p0 = seg[s], p1 = seg[s + 1];
  if (p0 >= p1) return;
  const uint32_t nq = work[p0].nq;
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= nq) return;
  // best[j] = (hd << 16 | candidate position in the segment), ascending; 0xFFFFFFFF = empty
  uint32_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 0xFFFFFFFFu;
  for (uint32_t p = p0; p < p1; ++p) {
    uint32_t key = keys[((size_t)p * planes) * max_kp + q];
    for (uint32_t pl = 1; pl < planes; ++pl) key = min(key, keys[((size_t)p * planes + pl) * max_kp + q]);
    const uint32_t hd = key >> 16;
    if (hd >= max_hd) continue;  // (257 = nothing searched is always >= max_hd <= 257)
    uint32_t v = (hd << 16) | (p - p0);  // a query has at most 65535 candidates
#pragma unroll
    for (int j = 0; j < K; ++j) {  // sorted insert
      const uint32_t lo = min(best[j], v);
      v = max(best[j], v);
      best[j] = lo;
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j)
    if ((uint32_t)j < k_use && best[j] != 0xFFFFFFFFu) atomicAdd(&votes[p0 + (best[j] & 0xFFFFu)], k_use - (uint32_t)j);  // :241
}

}  // namespace

void launch_place_votes(const uint32_t* keys, uint32_t planes, uint32_t max_kp, const PairWork* work, const uint32_t* seg,
                        uint32_t n_queries, uint32_t max_nq, uint32_t k_neighbours, uint32_t max_hd, uint32_t* votes,
                        hipStream_t stream) {
  if (max_nq == 0 || n_queries == 0) return;
  const dim3 grid((max_nq + 255u) / 256u, n_queries), block(256);
  if (k_neighbours <= 2)
    hipLaunchKernelGGL(place_vote_kernel<2>, grid, block, 0, stream, keys, planes, max_kp, work, seg, k_neighbours, max_hd, votes);
  else if (k_neighbours <= 4)
    hipLaunchKernelGGL(place_vote_kernel<4>, grid, block, 0, stream, keys, planes, max_kp, work, seg, k_neighbours, max_hd, votes);
  else
    hipLaunchKernelGGL(place_vote_kernel<kMaxK>, grid, block, 0, stream, keys, planes, max_kp, work, seg, k_neighbours, max_hd, votes);
}

}  // namespace rgbdfe
