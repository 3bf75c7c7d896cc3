// hamming_nn.hip -- batched 256-bit brute-force Hamming nearest neighbour for gfx950.
//
// Replaces the reference's hot loop 1: Node::featureMatching's ORB branch calling
// bruteForceSearchORB for every query row (src/node.cpp:561-576, src/features.cpp:163-182).
//
// Mapping (CDNA4, wave64):
//   * every lane owns QPL query descriptors in VGPRs (8 dwords each);
//   * the train descriptor of the current row is WAVE-UNIFORM: it is fetched with scalar
//     loads (s_load_dwordx8 through the scalar cache) and consumed as the SGPR operand of
//     v_xor_b32, so the inner loop is pure VALU: 8 x v_xor_b32 + 8 x v_bcnt_u32_b32
//     (accumulating form) + v_lshl_or_b32 + v_min_u32 per 256-bit compare -- no LDS, no
//     vector-memory traffic in the loop;
//   * (hd << 16 | train_row) packed keys turn "strict <, first minimum wins"
//     (features.cpp:176) into a single unsigned min; when the train rows of a SMALL batch
//     are split over several blocks every split writes its own key plane
//     (keys[pair][split][row]) and the consumer takes the min over the planes -- no
//     atomics, no pre-initialised buffer;
//   * the reference never visits the last train row (`i < size-1`, features.cpp:174):
//     rows [0, nt-1) are searched;
//   * blocks of one pair are placed on one XCD (block b runs on XCD b % 8) so the
//     pair's descriptors are pulled into one L2 only.
#include "rgbdfe_internal.h"

namespace rgbdfe {

constexpr int kHamThreads = 256;
constexpr int kQPL = 2;                                // queries per lane
constexpr int kQueriesPerBlock = kHamThreads * kQPL;   // 512

template <bool SPLIT>
__global__ __launch_bounds__(kHamThreads) void hamming_nn_kernel(
    const uint32_t* __restrict__ desc_pool, const PairWork* __restrict__ work,
    uint32_t* __restrict__ keys, uint32_t max_kp, uint32_t n_pairs, uint32_t tiles,
    uint32_t tsplit) {
  // XCD-aware decode: consecutive block ids round-robin over the 8 XCDs; give every XCD
  // whole pairs.
  const uint32_t L = blockIdx.x;
  const uint32_t xcd = L & 7u;
  const uint32_t j = L >> 3;
  const uint32_t subs = tiles * tsplit;
  const uint32_t pair = (j / subs) * 8u + xcd;
  if (pair >= n_pairs) return;
  const uint32_t sub = j % subs;
  const uint32_t tile = sub / tsplit;
  const uint32_t split = sub % tsplit;

  const PairWork w = work[pair];
  const uint32_t nq = w.nq;
  const uint32_t nt_search = w.nt > 0 ? w.nt - 1u : 0u;  // features.cpp:174 (and D4)

  // train row range of this block
  uint32_t t0 = 0, t1 = nt_search;
  if (SPLIT) {
    uint32_t chunk = (nt_search + tsplit - 1u) / tsplit;
    t0 = min(split * chunk, nt_search);
    t1 = min(t0 + chunk, nt_search);  // may be empty: the plane is then filled with "no match"
  }

  const uint32_t qbase = tile * kQueriesPerBlock + threadIdx.x;
  if (tile * kQueriesPerBlock >= nq) return;

  // query descriptors -> VGPRs (rows beyond nq are clamped; their result is discarded)
  uint32_t q[kQPL][8];
  const uint4* qpool = reinterpret_cast<const uint4*>(desc_pool + (size_t)w.q_slot * max_kp * 8u);
#pragma unroll
  for (int k = 0; k < kQPL; ++k) {
    uint32_t qi = qbase + k * kHamThreads;
    qi = qi < nq ? qi : nq - 1u;
    uint4 a = qpool[2u * qi], b = qpool[2u * qi + 1u];
    q[k][0] = a.x; q[k][1] = a.y; q[k][2] = a.z; q[k][3] = a.w;
    q[k][4] = b.x; q[k][5] = b.y; q[k][6] = b.z; q[k][7] = b.w;
  }

  uint32_t best[kQPL];
#pragma unroll
  for (int k = 0; k < kQPL; ++k) best[k] = 0xFFFFFFFFu;

  const uint32_t* __restrict__ tpool = desc_pool + (size_t)w.t_slot * max_kp * 8u;
  // Software pipeline over groups of 4 train rows: the scalar loads of group g+1 are
  // issued before the VALU work on group g (SMEM returns out of order, so the only
  // legal wait is lgkmcnt(0); issuing early moves that wait behind 144 VALU ops).
  // Prefetch may run up to 4 rows past t1: the slab is padded, the rows are not used.
  uint32_t t = t0;
  uint32_t cur[32];
  {
    const uint32_t* __restrict__ row = tpool + (size_t)t * 8u;
#pragma unroll
    for (int i = 0; i < 32; ++i) cur[i] = row[i];
  }
#pragma unroll 2
  for (; t + 4u <= t1; t += 4u) {
    uint32_t nxt[32];
    const uint32_t* __restrict__ row = tpool + (size_t)(t + 4u) * 8u;
#pragma unroll
    for (int i = 0; i < 32; ++i) nxt[i] = row[i];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int k = 0; k < kQPL; ++k) {
        uint32_t hd = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) hd += __builtin_popcount(q[k][i] ^ cur[g * 8 + i]);
        uint32_t key = (hd << 16) | (t + g);
        best[k] = min(best[k], key);
      }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) cur[i] = nxt[i];
  }
  // tail rows (< 4), already in `cur`
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (t + g < t1) {
#pragma unroll
      for (int k = 0; k < kQPL; ++k) {
        uint32_t hd = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) hd += __builtin_popcount(q[k][i] ^ cur[g * 8 + i]);
        uint32_t key = (hd << 16) | (t + g);
        best[k] = min(best[k], key);
      }
    }
  }

  uint32_t* kout = keys + ((size_t)pair * tsplit + split) * max_kp;
#pragma unroll
  for (int k = 0; k < kQPL; ++k) {
    uint32_t qi = qbase + k * kHamThreads;
    if (qi < nq) kout[qi] = best[k] == 0xFFFFFFFFu ? kNoMatchKey : best[k];
  }
}

HammingGeometry hamming_nn_geometry(uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_planes_capacity) {
  HammingGeometry g{0, 1};
  if (n_pairs == 0 || max_nq == 0) return g;
  g.qblocks = (max_nq + kQueriesPerBlock - 1) / kQueriesPerBlock;
  // Enough blocks to fill 256 CUs several times over; split the train rows when the
  // batch is small (live SLAM: ~20 pairs per frame).  Every split owns a key plane.
  const uint32_t blocks1 = n_pairs * g.qblocks;
  if (blocks1 < 2048 && max_nt > 64) {
    uint32_t tsplit = (2048 + blocks1 - 1) / blocks1;
    const uint32_t max_split = (max_nt + 63) / 64;  // at least 64 rows per block
    if (tsplit > max_split) tsplit = max_split;
    if (tsplit > 32) tsplit = 32;
    const uint32_t fit = key_planes_capacity / n_pairs;  // planes the keys buffer can hold
    if (tsplit > fit) tsplit = fit;
    if (tsplit < 1) tsplit = 1;
    g.tsplit = tsplit;
  }
  return g;
}

uint32_t launch_hamming_nn(const uint32_t* desc_pool, const PairWork* work, uint32_t* keys,
                           uint32_t max_kp, uint32_t n_pairs, HammingGeometry geom, hipStream_t stream) {
  if (n_pairs == 0 || geom.qblocks == 0) return 1;
  const uint32_t tiles = geom.qblocks, tsplit = geom.tsplit;
  const uint32_t pairs8 = (n_pairs + 7u) / 8u * 8u;
  const uint32_t grid = pairs8 * tiles * tsplit;
  if (tsplit > 1)
    hipLaunchKernelGGL(hamming_nn_kernel<true>, dim3(grid), dim3(kHamThreads), 0, stream,
                       desc_pool, work, keys, max_kp, n_pairs, tiles, tsplit);
  else
    hipLaunchKernelGGL(hamming_nn_kernel<false>, dim3(grid), dim3(kHamThreads), 0, stream,
                       desc_pool, work, keys, max_kp, n_pairs, tiles, tsplit);
  return tsplit;
}

}  // namespace rgbdfe
