// hamming_mfma.hip -- the 256-bit Hamming distance matrix as an exact fp4 contraction on the gfx950 matrix cores.
//
// Same contract as hamming_nn.hip (the popcount kernel): for every query row of the newer node the nearest train row
// of the older node under bruteForceSearchORB's rules (features.cpp:163-182: rows [0, nt-1) only, strict <, first
// minimum wins), delivered as (hd << 16 | row) keys.  The arithmetic is different:
//
//   * every descriptor bit becomes one fp4 (E2M1) operand nibble: +1.0 (0x2) for a 0 bit, -1.0 (0xA) for a 1 bit
//     (hamming_expand_kernel, once per node upload; 128 bytes per descriptor, stored in MFMA fragment order);
//     the query side flips the sign bit of every nibble on load, so one product is -1 for equal bits and +1 for
//     different bits and the 256-term dot product is  2*hd - 256  -- small integers, exact in the f32 accumulator;
//   * v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 4: both operands fp4) does 32 train rows x 32 queries x 64 bits per
//     instruction, four of them per 32x32 tile: 2*32*32*256 FLOP at the fp4 rate (4x the bf16 rate) instead of
//     16 VALU instructions per 64 compares;
//   * the accumulator is initialised with  256 + row_in_tile / 2^14, so an accumulator element is the complete sort key
//     2*hd + row/2^14  (24 significant bits at most: exact in f32), non-negative, hence ordered like its bit pattern:
//     "first minimum wins" is an unsigned integer minimum; train rows run along the accumulator registers of a lane,
//     queries along the lanes, so the running minimum is ONE register per 32-query tile and the epilogue is a
//     v_min3_u32 tree: 10 VALU instructions per 16 matrix elements;
//   * a 256-thread block holds 256 queries (two 32-query operand tiles per wave) and streams the train tiles of the
//     pair through LDS (16 KB stages, double buffered, one barrier per stage); the four waves share every tile.
//
// MODE 1 adds the row term inside the MFMA (C operand); MODE 2 starts from C = 0 and adds it with v_add_f32 (kept as
// the conservative variant: it does not rely on the matrix core adding a 2^-14-granular C to the integer sum exactly);
// MODE 3 (hamming_mfma_pipe_kernel below, the default since round 4) is mode 1's arithmetic as a software pipeline inside
// every wave: reductions and LDS reads between the MFMAs, train tiles by global_load_lds through three LDS buffers.
// Keys are identical to the popcount kernel's, bit for bit (tests/test_gpu_hamming.py runs every mode against the
// golden vectors of the reference function).  Usable while max_keypoints <= 32768 (15 index bits next to 9 distance
// bits in a 24-bit significand); the host falls back to the popcount kernel above that.
#include "rgbdfe_internal.h"

namespace rgbdfe {

// The train tiles of ONE block of hamming_mfma_pipe_kernel (mode 3) as slots of four-tile stages: the block's full tiles
// first, then phantoms, the pair's ragged tile -- if it is this block's -- in the LAST slot of the last stage.  Host and
// device (rgbdfe_debug_hamming_slots, tests/test_hamming_slots.py check the enumeration without a GPU).
constexpr uint32_t kHammingPhantomTile = 0x7FFFFFFFu;
struct HammingSlots {
  uint32_t nt_search, n_ttiles, tile0, tile1, n_full, n_stages, last_slot;
  bool has_ragged;
  __host__ __device__ static HammingSlots make(uint32_t nt, uint32_t tsplit, uint32_t split, uint32_t tiles_per_stage) {
    HammingSlots s;
    s.nt_search = nt > 0 ? nt - 1u : 0u;   // features.cpp:174 (and D4): the node's last row is never a candidate
    s.n_ttiles = (s.nt_search + 31u) >> 5;
    s.tile0 = 0; s.tile1 = s.n_ttiles;
    if (tsplit > 1u) {
      const uint32_t chunk = (s.n_ttiles + tsplit - 1u) / tsplit;
      s.tile0 = split * chunk < s.n_ttiles ? split * chunk : s.n_ttiles;
      s.tile1 = s.tile0 + chunk < s.n_ttiles ? s.tile0 + chunk : s.n_ttiles;
    }
    const uint32_t full_tiles = s.nt_search >> 5;
    const uint32_t fend = s.tile1 < full_tiles ? s.tile1 : full_tiles;
    s.n_full = fend > s.tile0 ? fend - s.tile0 : 0u;
    s.has_ragged = (s.nt_search & 31u) != 0u && s.tile1 == s.n_ttiles && s.tile1 > s.tile0;
    s.n_stages = (s.n_full + (s.has_ragged ? 1u : 0u) + tiles_per_stage - 1u) / tiles_per_stage;
    s.last_slot = s.n_stages * tiles_per_stage - 1u;
    return s;
  }
  __host__ __device__ uint32_t tile(uint32_t slot) const {
    const uint32_t behind = (has_ragged & (slot == last_slot)) ? n_ttiles - 1u : kHammingPhantomTile;   // (selects, no branches)
    return slot < n_full ? tile0 + slot : behind;
  }
};

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
#ifndef RGBDFE_HAMMING_QT
#define RGBDFE_HAMMING_QT 2
#endif
#ifndef RGBDFE_HAMMING_STAGE
#define RGBDFE_HAMMING_STAGE 4
#endif
constexpr int kQT = RGBDFE_HAMMING_QT;          // 32-query operand tiles per wave
constexpr int kQueriesPerBlock = 4 * kQT * 32;  // 256
constexpr int kStage = RGBDFE_HAMMING_STAGE;    // train tiles (of 32 rows, 4 KB each) per LDS stage
constexpr float kNone = 3.0e38f;
constexpr float kRowUnit = 1.0f / 16384.0f;  // 2^-14
constexpr float kBias = 256.0f;  // element = 2*hd + row / 2^14 >= 0 (24 significant bits at most: 2*hd <= 512, row < 2^15)

// 8 descriptor bits -> 8 fp4 nibbles (0x2 = +1.0 for a 0 bit, 0xA = -1.0 for a 1 bit)
__device__ __forceinline__ uint32_t spread8(uint32_t x) {
  uint32_t t = (x | (x << 12)) & 0x000F000Fu;
  t = (t | (t << 6)) & 0x03030303u;
  t = (t | (t << 3)) & 0x11111111u;
  return 0x22222222u | (t << 3);
}

// One thread per (row of a 32-row tile, descriptor dword): writes the 16 bytes lane (half*32 + row) feeds to k-step s,
// with dword = 2*s + half.  Rows >= n of a touched tile are written as zeros (0.0 operands).
__global__ __launch_bounds__(256) void hamming_expand_kernel(const uint32_t* __restrict__ rows, uint4* __restrict__ tiles,
                                                             uint32_t n) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;  // tile*256 + s*64 + half*32 + row
  const uint32_t tile = gid >> 8, s = (gid >> 6) & 3u, half = (gid >> 5) & 1u, r = gid & 31u;
  const uint32_t row = tile * 32u + r;
  if (tile * 32u >= n) return;
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (row < n) {
    const uint32_t w = rows[(size_t)row * 8u + 2u * s + half];
    o.x = spread8(w & 0xFFu);
    o.y = spread8((w >> 8) & 0xFFu);
    o.z = spread8((w >> 16) & 0xFFu);
    o.w = spread8(w >> 24);
  }
  tiles[gid] = o;
}

__device__ __forceinline__ v8i as_operand(uint4 v) {
  v8i o;
  o[0] = (int)v.x; o[1] = (int)v.y; o[2] = (int)v.z; o[3] = (int)v.w;
  o[4] = 0; o[5] = 0; o[6] = 0; o[7] = 0;
  return o;
}

// Accumulator elements are non-negative (the C operand carries a +256 bias next to the row term), and non-negative
// floats order like their bit patterns: the minimum is an INTEGER min3 tree -- fminf would add a canonicalising v_max
// in front of every v_min3_f32 (44 of ~190 VALU instructions per four tiles).
__device__ __forceinline__ uint32_t min16(const v16f& a) {
  uint32_t x[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = __float_as_uint(a[r]);
  const uint32_t m0 = min(min(x[0], x[1]), x[2]);
  const uint32_t m1 = min(min(x[3], x[4]), x[5]);
  const uint32_t m2 = min(min(x[6], x[7]), x[8]);
  const uint32_t m3 = min(min(x[9], x[10]), x[11]);
  const uint32_t m4 = min(min(x[12], x[13]), x[14]);
  const uint32_t m5 = min(min(m0, m1), x[15]);
  const uint32_t m6 = min(min(m2, m3), m4);
  return min(m5, m6);
}

template <int MODE, bool SPLIT>
__global__ __launch_bounds__(kThreads) void hamming_mfma_kernel(const uint4* __restrict__ slab,
                                                                const PairWork* __restrict__ work,
                                                                uint32_t* __restrict__ keys, uint32_t max_kp,
                                                                uint32_t tiles_per_slot, uint32_t n_pairs,
                                                                uint32_t qblocks, uint32_t tsplit) {
  __shared__ uint4 lds[2][kStage * 256];
  // whole pairs per XCD (block b runs on XCD b % 8), as in hamming_nn_kernel
  const uint32_t L = blockIdx.x;
  const uint32_t xcd = L & 7u;
  const uint32_t j = L >> 3;
  const uint32_t subs = qblocks * tsplit;
  const uint32_t pair = (j / subs) * 8u + xcd;
  if (pair >= n_pairs) return;
  const uint32_t sub = j % subs;
  const uint32_t qblock = sub / tsplit;
  const uint32_t split = sub % tsplit;

  const PairWork w = work[pair];
  const uint32_t nq = w.nq;
  if (qblock * kQueriesPerBlock >= nq) return;
  const uint32_t nt_search = w.nt > 0 ? w.nt - 1u : 0u;  // features.cpp:174 (and D4)
  const uint32_t n_ttiles = (nt_search + 31u) >> 5;
  uint32_t tile0 = 0, tile1 = n_ttiles;
  if (SPLIT) {
    const uint32_t chunk = (n_ttiles + tsplit - 1u) / tsplit;
    tile0 = min(split * chunk, n_ttiles);
    tile1 = min(tile0 + chunk, n_ttiles);
  }

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t half = lane >> 5;

  // query operands: kQT tiles x 4 k-steps, sign-flipped
  v8i bq[kQT][4];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    uint32_t qt = qblock * (kQueriesPerBlock / 32) + wave * kQT + t;
    qt = min(qt, tiles_per_slot - 1u);  // tiles beyond the node's rows: results are discarded
    const uint4* __restrict__ qs = slab + ((size_t)w.q_slot * tiles_per_slot + qt) * 256u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 v = qs[s * 64 + lane];
      v.x ^= 0x88888888u; v.y ^= 0x88888888u; v.z ^= 0x88888888u; v.w ^= 0x88888888u;
      bq[t][s] = as_operand(v);
    }
  }

  // accumulator register r of this lane belongs to train row (r & 3) + 8 * (r >> 2) + 4 * half of the tile
  v16f crow;
#pragma unroll
  for (int r = 0; r < 16; ++r) crow[r] = kBias + (float)((r & 3) + 8 * (r >> 2) + 4 * (int)half) * kRowUnit;
  v16f czero;
#pragma unroll
  for (int r = 0; r < 16; ++r) czero[r] = 0.f;
  // The pair's last tile is ragged (rows >= nt_search do not take part: at least the node's last row, features.cpp:174):
  // its row term is kNone for those rows, so their accumulator elements can never be a minimum -- the exclusion costs
  // nothing in the epilogue, the ragged tile just starts from another C vector (3e38 + a dot product of at most 256 in
  // magnitude stays 3e38).
  v16f crow_ragged;
  {
    const uint32_t rag0 = (nt_search >> 5) << 5;  // first row of the ragged tile (no ragged tile when nt_search % 32 == 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t row = rag0 + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
      crow_ragged[r] = row < nt_search ? crow[r] : kNone;
    }
  }

  uint32_t best[kQT];  // float bit patterns
#pragma unroll
  for (int t = 0; t < kQT; ++t) best[t] = __float_as_uint(kNone);

  const uint4* __restrict__ ts = slab + (size_t)w.t_slot * tiles_per_slot * 256u;
  const uint32_t n_tiles = tile1 - tile0;
  const uint32_t n_stages = (n_tiles + kStage - 1u) / kStage;
  const uint32_t last_tile = tiles_per_slot - 1u;

  // stage 0 -> LDS
  {
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const uint32_t tl = min(tile0 + (uint32_t)i, last_tile);
      lds[0][i * 256 + threadIdx.x] = ts[(size_t)tl * 256u + threadIdx.x];
    }
  }
  __syncthreads();

  for (uint32_t st = 0; st < n_stages; ++st) {
    // the next stage's tiles travel through registers while this stage is computed (behind the last stage the
    // clamped addresses are simply read again: no branch around the loads, nothing is done with them)
    uint4 nxt[kStage];
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
      const uint32_t tl = min(tile0 + (st + 1u) * kStage + (uint32_t)i, last_tile);
      nxt[i] = ts[(size_t)tl * 256u + threadIdx.x];
    }
    const uint4* __restrict__ buf = lds[st & 1u];
    // one train tile against this wave's query tiles; CROW = the row term (C operand in mode 1, added after in mode 2)
    auto do_tile = [&](int i, uint32_t tile, const v16f& CROW) {
      v8i a[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = as_operand(buf[i * 256 + s * 64 + lane]);
      const float base = (float)(tile * 32u) * kRowUnit;
      // both query tiles' MFMA chains are issued before the first epilogue: the matrix pipe works on tile 1 while
      // the VALU reduces tile 0
      v16f acc[kQT];
#pragma unroll
      for (int t = 0; t < kQT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], bq[t][0], MODE == 1 ? CROW : czero, 4, 4, 0, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], bq[t][1], acc[t], 4, 4, 0, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[2], bq[t][2], acc[t], 4, 4, 0, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[3], bq[t][3], acc[t], 4, 4, 0, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < kQT; ++t) {
        if (MODE != 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += CROW[r];
        }
        const float m = __uint_as_float(min16(acc[t]));
        best[t] = min(best[t], __float_as_uint(m + base));  // kNone + base stays huge
      }
    };
    const uint32_t stage_tile0 = tile0 + st * kStage;
    if (stage_tile0 + kStage <= tile1 && (stage_tile0 + kStage) * 32u <= nt_search) {
      // a full stage without the ragged tile: ONE basic block, so the LDS reads of tile i + 1 can be scheduled beside
      // the MFMAs of tile i
#pragma unroll
      for (int i = 0; i < kStage; ++i) do_tile(i, stage_tile0 + (uint32_t)i, crow);
    } else {
#pragma unroll
      for (int i = 0; i < kStage; ++i) {
        const uint32_t tile = stage_tile0 + (uint32_t)i;
        if (tile < tile1) {  // block-uniform
          if (tile * 32u + 32u > nt_search) do_tile(i, tile, crow_ragged);  // block-uniform: the pair's last tile only
          else do_tile(i, tile, crow);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kStage; ++i) lds[(st + 1u) & 1u][i * 256 + threadIdx.x] = nxt[i];
    __syncthreads();
  }

  uint32_t* kout = keys + ((size_t)pair * tsplit + split) * max_kp;
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    uint32_t bb = best[t];
    bb = min(bb, (uint32_t)__shfl_xor((int)bb, 32));  // the two row halves of the tiles
    const float b = __uint_as_float(bb);
    const uint32_t qi = qblock * kQueriesPerBlock + (wave * kQT + (uint32_t)t) * 32u + (lane & 31u);
    if (half == 0 && qi < nq) {
      uint32_t key = kNoMatchKey;
      if (b < 1.0e30f) {
        // 2*hd + row / 2^14  ->  hd * 2^15 + row   (exact: < 2^24)
        const uint32_t k = (uint32_t)(b * 16384.0f);
        key = ((k >> 15) << 16) | (k & 32767u);
      }
      kout[qi] = key;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// MODE 3: the same contraction as a software pipeline inside every wave (round 4).
//
// In the kernel above a wave issues the eight MFMAs of a train tile in one burst and reduces the two accumulators
// afterwards: the matrix pipe only stays busy while ANOTHER wave of the SIMD happens to be in its burst, and bursts of
// co-resident waves drift into step (measured: matrix pipe 0.61 + VALU 0.40 of the kernel time, i.e. no overlap).  Here the
// overlap is built into the instruction stream of each wave:
//   * the unit of work is one chain of four MFMAs (one train tile x one of the wave's two query tiles); the v_min3 tree of
//     unit u - 1 and the LDS reads of the next train tile are placed in the gaps between the MFMAs of unit u
//     (sched_group_barrier pins the order): ~3.3 single-issue instructions per 32-cycle MFMA;
//   * the two accumulators alternate (unit u writes acc[u & 1] while acc[(u - 1) & 1] is reduced): no more registers than
//     the burst form;
//   * train tiles come in stages of four through THREE LDS buffers filled by global_load_lds (no staging registers, no
//     ds_write); the one barrier per stage sits in the MIDDLE of the stage's instruction stream -- it publishes the NEXT
//     stage (loaded a whole stage ago) and frees the buffer of the previous one -- so the stream of MFMAs does not drain at
//     a stage boundary;
//   * every stage is the same straight-line code: the pair's ragged tile is moved to the last slot of the last stage
//     (the minimum does not care about the order of the tiles) where the C operand is `c3` (= the row term with 3e38 for
//     the excluded rows in that one stage), missing tiles are phantoms: they read some valid tile and add a base of
//     2^22, which no real key reaches (real keys are < 1024).
// Keys are those of the other kernels, bit for bit (tests/test_gpu_hamming.py runs all four).
// ------------------------------------------------------------------------------------------------
#ifndef RGBDFE_HAMMING_PIPE_BURST
#define RGBDFE_HAMMING_PIPE_BURST 0
#endif
#ifndef RGBDFE_HAMMING_PIPE_ALT4
#define RGBDFE_HAMMING_PIPE_ALT4 0
#endif
#ifndef RGBDFE_HAMMING_PIPE_WAVES
#define RGBDFE_HAMMING_PIPE_WAVES 3          // waves per SIMD the register allocation aims at (48 KB of LDS: three blocks per CU)
#endif
#ifndef RGBDFE_HAMMING_PIPE_VALU_GROUPS
#define RGBDFE_HAMMING_PIPE_VALU_GROUPS 334  // reduction instructions behind the 2nd / 3rd / 4th MFMA of a unit (diagnostics)
#endif
#ifndef RGBDFE_HAMMING_PIPE_DIAG
#define RGBDFE_HAMMING_PIPE_DIAG 0   // TIMING-ONLY builds (keys are forced to "no match"): 1 no reductions, 2 no LDS reads in the
#endif                               // loop, 4 no global_load_lds / barriers in the loop; tools/build_hamming_pipe_variants.sh
constexpr int kPipeStage = 4;  // train tiles per stage (the unrolled stream below is written for four)
constexpr float kRowUnitPerTile = 32.0f / 16384.0f;  // the tile's first row, in key units

#define HP_MFMA(ACC, A, B, C) ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 4, 4, 0, 0, 0, 0);
#define HP_SGB(MASK, N) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
#define HP_FENCE() __builtin_amdgcn_sched_barrier(0);

template <bool SPLIT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(RGBDFE_HAMMING_PIPE_WAVES, RGBDFE_HAMMING_PIPE_WAVES)))
void hamming_mfma_pipe_kernel(const uint4* __restrict__ slab,
                                                                     const PairWork* __restrict__ work,
                                                                     uint32_t* __restrict__ keys, uint32_t max_kp,
                                                                     uint32_t tiles_per_slot, uint32_t n_pairs,
                                                                     uint32_t qblocks, uint32_t tsplit) {
  // three stage buffers [slot][fragment order of a tile] as three VARIABLES: the compiler tells LDS objects apart by
  // variable, and only then does it let the LDS reads of one buffer pass the global_load_lds towards another
  __shared__ uint4 lds0[kPipeStage][256], lds1[kPipeStage][256], lds2[kPipeStage][256];
  const uint32_t L = blockIdx.x;
  const uint32_t xcd = L & 7u;
  const uint32_t j = L >> 3;
  const uint32_t subs = qblocks * tsplit;
  const uint32_t pair = (j / subs) * 8u + xcd;
  if (pair >= n_pairs) return;
  const uint32_t sub = j % subs;
  const uint32_t qblock = sub / tsplit;
  const uint32_t split = sub % tsplit;

  const PairWork w = work[pair];
  const uint32_t nq = w.nq;
  if (qblock * kQueriesPerBlock >= nq) return;
  // this block's tiles as slots: the full tiles first, phantoms, the ragged tile (if it is this block's) in the last slot
  const HammingSlots slots = HammingSlots::make(w.nt, SPLIT ? tsplit : 1u, SPLIT ? split : 0u, kPipeStage);
  const uint32_t nt_search = slots.nt_search, n_stages = slots.n_stages;
  const bool has_ragged = slots.has_ragged;
  auto slot_tile = [&](uint32_t s) -> uint32_t { return slots.tile(s); };  // block-uniform

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t half = lane >> 5;

  v8i bq[kQT][4];
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    uint32_t qt = qblock * (kQueriesPerBlock / 32) + wave * kQT + t;
    qt = min(qt, tiles_per_slot - 1u);
    const uint4* __restrict__ qs = slab + ((size_t)w.q_slot * tiles_per_slot + qt) * 256u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 v = qs[s * 64 + lane];
      v.x ^= 0x88888888u; v.y ^= 0x88888888u; v.z ^= 0x88888888u; v.w ^= 0x88888888u;
      bq[t][s] = as_operand(v);
    }
  }

  v16f crow, c3;
#pragma unroll
  for (int r = 0; r < 16; ++r) crow[r] = kBias + (float)((r & 3) + 8 * (r >> 2) + 4 * (int)half) * kRowUnit;
  c3 = crow;
  auto set_c3_ragged = [&]() {
    uint32_t nts = nt_search;
    asm volatile("" : "+v"(nts));  // formed where it is needed (once per block), not kept in 16 registers from the start
    const uint32_t rag0 = (nts >> 5) << 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t row = rag0 + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * half;
      c3[r] = row < nts ? crow[r] : kNone;
    }
  };

  uint32_t best0 = __float_as_uint(kNone), best1 = __float_as_uint(kNone);

  const uint32_t last_tile = tiles_per_slot - 1u;
  const char* __restrict__ ts = reinterpret_cast<const char*>(slab + (size_t)w.t_slot * tiles_per_slot * 256u);
  const uint64_t ts_u = reinterpret_cast<uint64_t>(ts);
  const uint32_t ts_lo = __builtin_amdgcn_readfirstlane((uint32_t)ts_u);
  const uint32_t ts_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ts_u >> 32));
  const uint32_t vo16 = threadIdx.x * 16u;

  // stage `st` of this block -> LDS buffer `b` (this wave's quarter of each of the four tiles)
  auto load_stage = [&](uint32_t st, uint4 (*buf)[256]) {
#pragma unroll
    for (int i = 0; i < kPipeStage; ++i) {
      const uint32_t tl = min(slot_tile(st * kPipeStage + (uint32_t)i), last_tile);
      const char* tb = reinterpret_cast<const char*>((((uint64_t)ts_hi << 32) | ts_lo) + (uint64_t)tl * 4096u);
      uint32_t vo = vo16;
      asm volatile("" : "+v"(vo));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + vo),
                                       (__attribute__((address_space(3))) void*)&buf[i][wave * 64],
                                       16, 0, 0);
    }
  };

  if (n_stages > 0) {
    static_assert(kQT == 2 && kPipeStage == 4, "the unrolled stream is written for two query tiles and four-tile stages");
    load_stage(0, lds0);
    if (n_stages > 1) load_stage(1, lds1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();

    v8i aA[4], aB[4];  // A operands of the even / odd slots of a stage
    // (BUF = 0, 1, 2 is a literal everywhere: the compiler must SEE that a stage's LDS reads and the global_load_lds of the
    // stage after the next one touch different buffers, or it waits for the loads in front of the next LDS read)
#if RGBDFE_HAMMING_PIPE_DIAG & 2
#define HP_READ(DST, BUF, SLOT, S) asm volatile("" : "+v"(DST[S]));
#else
#define HP_READ(DST, BUF, SLOT, S) DST[S] = as_operand(lds##BUF[SLOT][(S) * 64 + lane]);
#endif
#define HP_BASE(SLOT) ((float)slot_tile(st * kPipeStage + (SLOT)) * kRowUnitPerTile)
    // reduction of one accumulator into the running minimum of its query tile
#if RGBDFE_HAMMING_PIPE_DIAG & 1
#define HP_EPI(ACC, BEST, BASE) asm volatile("" : : "v"(ACC));
#else
#define HP_EPI(ACC, BEST, BASE)                                                 \
    BEST = min(BEST, __float_as_uint(__uint_as_float(min16(ACC)) + (BASE))); \
    asm volatile("" : "+v"(BEST));   /* the reduction is finished HERE, not merged into a later one */
#endif
    // one unit: four MFMAs with the previous unit's reduction (10 VALU) and two LDS reads in their gaps
#if RGBDFE_HAMMING_PIPE_BURST   /* diagnostics: the chain back to back, everything else behind it */
#define HP_UNIT_SCHED() HP_SGB(0x008, 4) HP_SGB(0x100, 2) HP_SGB(0x002, 12)
#else
#define HP_UNIT_SCHED()                                                                             \
    HP_SGB(0x008, 1) HP_SGB(0x100, 2)                                                               \
    HP_SGB(0x008, 1) HP_SGB(0x002, RGBDFE_HAMMING_PIPE_VALU_GROUPS / 100)                            \
    HP_SGB(0x008, 1) HP_SGB(0x002, RGBDFE_HAMMING_PIPE_VALU_GROUPS / 10 % 10)                        \
    HP_SGB(0x008, 1) HP_SGB(0x002, RGBDFE_HAMMING_PIPE_VALU_GROUPS % 10)
#endif
#define HP_UNIT(ACC, A, Q, C, PACC, PBEST, PBASE, R0, R1)                                           \
    HP_MFMA(ACC, A[0], bq[Q][0], C)                                                                 \
    R0 R1                                                                                           \
    HP_MFMA(ACC, A[1], bq[Q][1], ACC)                                                               \
    HP_EPI(PACC, PBEST, PBASE)                                                                      \
    HP_MFMA(ACC, A[2], bq[Q][2], ACC)                                                               \
    HP_MFMA(ACC, A[3], bq[Q][3], ACC)                                                               \
    HP_UNIT_SCHED()                                                                                 \
    HP_FENCE()
    // A stage out of buffer CUR in two halves; the barrier between them publishes the NEXT stage (this wave's
    // global_load_lds of it was issued a whole stage ago) and frees the buffer of the PREVIOUS one for the stage after the
    // next.  The loop below turns at that barrier, where nothing is in flight (the compiler's bookkeeping of what an LDS
    // read may have to wait for is exact in straight-line code only).
#if RGBDFE_HAMMING_PIPE_DIAG & 4
#define HP_TURN()
#define HP_LOADS false
#else
    /* vmcnt(0): this wave's share of the next stage has landed */
#define HP_TURN() __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads();
#define HP_LOADS true
#endif
#if RGBDFE_HAMMING_PIPE_ALT4
    // Diagnostics (2 waves per SIMD: four accumulators): a unit is a whole train tile, the chains of the two query tiles
    // ALTERNATE (no instruction ever sits between two MFMAs on the same accumulator without an MFMA on another one), the
    // previous tile's two reductions fill the gaps.
#define HP_TILE(X0, X1, A, C, P0, P1, PBASE, R0, R1, R2, R3)                                         \
    HP_MFMA(X0, A[0], bq[0][0], C) HP_MFMA(X1, A[0], bq[1][0], C)                                    \
    R0 R1 R2 R3                                                                                      \
    HP_MFMA(X0, A[1], bq[0][1], X0) HP_MFMA(X1, A[1], bq[1][1], X1)                                  \
    HP_EPI(P0, best0, PBASE)                                                                         \
    HP_MFMA(X0, A[2], bq[0][2], X0) HP_MFMA(X1, A[2], bq[1][2], X1)                                  \
    HP_EPI(P1, best1, PBASE)                                                                         \
    HP_MFMA(X0, A[3], bq[0][3], X0) HP_MFMA(X1, A[3], bq[1][3], X1)                                  \
    HP_SGB(0x008, 1) HP_SGB(0x100, 1) HP_SGB(0x008, 1) HP_SGB(0x100, 1)                              \
    HP_SGB(0x008, 1) HP_SGB(0x100, 1) HP_SGB(0x002, 2) HP_SGB(0x008, 1) HP_SGB(0x100, 1) HP_SGB(0x002, 3) \
    HP_SGB(0x008, 1) HP_SGB(0x002, 4) HP_SGB(0x008, 1) HP_SGB(0x002, 4)                              \
    HP_SGB(0x008, 1) HP_SGB(0x002, 4) HP_SGB(0x008, 1) HP_SGB(0x002, 4)                              \
    HP_FENCE()
#define HP_HALF_A(CUR)                                                                                               \
    {                                                                                                                \
      b0 = HP_BASE(0); b1 = HP_BASE(1);                                                                              \
      HP_FENCE()                                                                                                     \
      HP_TILE(acc0, acc1, aA, crow, acc2, acc3, pend, HP_READ(aB, CUR, 1, 0), HP_READ(aB, CUR, 1, 1),                \
              HP_READ(aB, CUR, 1, 2), HP_READ(aB, CUR, 1, 3))                                                        \
      HP_TILE(acc2, acc3, aB, crow, acc0, acc1, b0, HP_READ(aA, CUR, 2, 0), HP_READ(aA, CUR, 2, 1),                  \
              HP_READ(aA, CUR, 2, 2), HP_READ(aA, CUR, 2, 3))                                                        \
      HP_TURN()                                                                                                      \
    }
#define HP_HALF_B(CUR, NXT, NN)                                                                                      \
    {                                                                                                                \
      if (HP_LOADS && st + 2u < n_stages) load_stage(st + 2u, lds##NN);                                              \
      const float b2 = HP_BASE(2), b3 = HP_BASE(3);                                                                  \
      if (has_ragged && st + 1u == n_stages) {                                                                       \
        set_c3_ragged();                                                                                             \
        asm volatile("" : "+v"(c3));                                                                                 \
      }                                                                                                              \
      HP_FENCE()                                                                                                     \
      HP_TILE(acc0, acc1, aA, crow, acc2, acc3, b1, HP_READ(aB, CUR, 3, 0), HP_READ(aB, CUR, 3, 1),                  \
              HP_READ(aB, CUR, 3, 2), HP_READ(aB, CUR, 3, 3))                                                        \
      HP_TILE(acc2, acc3, aB, c3, acc0, acc1, b2, HP_READ(aA, NXT, 0, 0), HP_READ(aA, NXT, 0, 1),                    \
              HP_READ(aA, NXT, 0, 2), HP_READ(aA, NXT, 0, 3))                                                        \
      pend = b3;                                                                                                     \
    }
#else
#define HP_HALF_A(CUR)                                                                                               \
    {                                                                                                                \
      b0 = HP_BASE(0); b1 = HP_BASE(1);                                                                              \
      HP_FENCE()                                                                                                     \
      HP_UNIT(acc0, aA, 0, crow, acc1, best1, pend, HP_READ(aB, CUR, 1, 0), HP_READ(aB, CUR, 1, 1))                  \
      HP_UNIT(acc1, aA, 1, crow, acc0, best0, b0, HP_READ(aB, CUR, 1, 2), HP_READ(aB, CUR, 1, 3))                    \
      HP_UNIT(acc0, aB, 0, crow, acc1, best1, b0, HP_READ(aA, CUR, 2, 0), HP_READ(aA, CUR, 2, 1))                    \
      HP_UNIT(acc1, aB, 1, crow, acc0, best0, b1, HP_READ(aA, CUR, 2, 2), HP_READ(aA, CUR, 2, 3))                    \
      HP_TURN()                                                                                                      \
    }
#define HP_HALF_B(CUR, NXT, NN)                                                                                      \
    {                                                                                                                \
      if (HP_LOADS && st + 2u < n_stages) load_stage(st + 2u, lds##NN); /* (nothing may be in flight at the end) */   \
      const float b2 = HP_BASE(2), b3 = HP_BASE(3);                                                                  \
      if (has_ragged && st + 1u == n_stages) {                                                                       \
        set_c3_ragged();                                                                                             \
        asm volatile("" : "+v"(c3)); /* a real (block-uniform) branch: not sixteen selects in every stage */        \
      }                                                                                                              \
      HP_FENCE()                                                                                                     \
      HP_UNIT(acc0, aA, 0, crow, acc1, best1, b1, HP_READ(aB, CUR, 3, 0), HP_READ(aB, CUR, 3, 1))                    \
      HP_UNIT(acc1, aA, 1, crow, acc0, best0, b2, HP_READ(aB, CUR, 3, 2), HP_READ(aB, CUR, 3, 3))                    \
      HP_UNIT(acc0, aB, 0, c3, acc1, best1, b2, HP_READ(aA, NXT, 0, 0), HP_READ(aA, NXT, 0, 1))                      \
      HP_UNIT(acc1, aB, 1, c3, acc0, best0, b3, HP_READ(aA, NXT, 0, 2), HP_READ(aA, NXT, 0, 3))                      \
      pend = b3;                                                                                                     \
    }

#endif
    HP_READ(aA, 0, 0, 0) HP_READ(aA, 0, 0, 1) HP_READ(aA, 0, 0, 2) HP_READ(aA, 0, 0, 3)
#if RGBDFE_HAMMING_PIPE_ALT4
    v16f acc0, acc1, acc2 = crow, acc3 = crow;
#else
    v16f acc0, acc1 = crow;
#endif
    float pend = 4.0e6f;  // acc1 = the row terms (>= 256), + 4e6: a phantom for the first unit's reduction slot
    float b0, b1;
    uint32_t st = 0;
    HP_HALF_A(0)
    for (;;) {
      HP_HALF_B(0, 1, 2)
      if (++st == n_stages) break;
      HP_HALF_A(1)
      HP_HALF_B(1, 2, 0)
      if (++st == n_stages) break;
      HP_HALF_A(2)
      HP_HALF_B(2, 0, 1)
      if (++st == n_stages) break;
      HP_HALF_A(0)
    }
#if RGBDFE_HAMMING_PIPE_ALT4
    HP_EPI(acc2, best0, pend)
    HP_EPI(acc3, best1, pend)
#else
    HP_EPI(acc1, best1, pend)
#endif
#undef HP_HALF_A
#undef HP_HALF_B
#undef HP_TURN
#undef HP_LOADS
#undef HP_UNIT
#undef HP_EPI
#undef HP_BASE
#undef HP_READ
  }

  uint32_t* kout = keys + ((size_t)pair * tsplit + split) * max_kp;
  const uint32_t bests[2] = {best0, best1};
#pragma unroll
  for (int t = 0; t < kQT; ++t) {
    uint32_t bb = bests[t];
    bb = min(bb, (uint32_t)__shfl_xor((int)bb, 32));
    const float b = __uint_as_float(bb);
    const uint32_t qi = qblock * kQueriesPerBlock + (wave * kQT + (uint32_t)t) * 32u + (lane & 31u);
    if (half == 0 && qi < nq) {
      uint32_t key = kNoMatchKey;
      if (RGBDFE_HAMMING_PIPE_DIAG ? b < -1.0f : b < 1024.0f) {  // real keys: 2*hd + row / 2^14 < 514; phantoms and excluded rows are >= 2^22
        const uint32_t k = (uint32_t)(b * 16384.0f);
        key = ((k >> 15) << 16) | (k & 32767u);
      }
      kout[qi] = key;
    }
  }
}
#undef HP_MFMA
#undef HP_SGB
#undef HP_FENCE

}  // namespace

}  // namespace rgbdfe

// tests: the slot enumeration of one block of the pipelined kernel (host only; tiles_out[s] = train tile of slot s or
// 0x7FFFFFFF for a phantom); returns the number of slots (4 x stages) or -1 when they do not fit
extern "C" int rgbdfe_debug_hamming_slots(uint32_t nt, uint32_t tsplit, uint32_t split, uint32_t* tiles_out, int capacity,
                                          int* has_ragged, uint32_t* n_ttiles) {
  const rgbdfe::HammingSlots s = rgbdfe::HammingSlots::make(nt, tsplit, split, 4u);
  if (has_ragged) *has_ragged = s.has_ragged ? 1 : 0;
  if (n_ttiles) *n_ttiles = s.n_ttiles;
  const int n = (int)(s.n_stages * 4u);
  if (n > capacity) return -1;
  for (int i = 0; i < n; ++i) tiles_out[i] = s.tile((uint32_t)i);
  return n;
}

namespace rgbdfe {

uint32_t hamming_mfma_tiles_per_slot(uint32_t max_kp) { return (max_kp + 31u) / 32u; }

size_t hamming_mfma_slab_bytes(uint32_t max_nodes, uint32_t max_kp) {
  return ((size_t)max_nodes * hamming_mfma_tiles_per_slot(max_kp) + 1u) * 4096u;
}

void launch_hamming_expand(const uint32_t* node_rows, uint32_t* slab, uint32_t slot, uint32_t max_kp, uint32_t n,
                           hipStream_t stream) {
  if (n == 0) return;
  const uint32_t tiles = (n + 31u) / 32u;
  uint4* dst = reinterpret_cast<uint4*>(slab) + (size_t)slot * hamming_mfma_tiles_per_slot(max_kp) * 256u;
  hipLaunchKernelGGL(hamming_expand_kernel, dim3(tiles), dim3(256), 0, stream, node_rows, dst, n);
}

HammingGeometry hamming_mfma_geometry(uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_planes_capacity) {
  HammingGeometry g{0, 1};
  if (n_pairs == 0 || max_nq == 0) return g;
  g.qblocks = (max_nq + kQueriesPerBlock - 1) / kQueriesPerBlock;
  // small batches (live SLAM: ~20 pairs per frame) split the train tiles over several blocks, one key plane each
  const uint32_t blocks1 = n_pairs * g.qblocks;
  const uint32_t ttiles = (max_nt + 31u) / 32u;
  if (blocks1 < 1024 && ttiles > kStage) {
    uint32_t tsplit = (1024 + blocks1 - 1) / blocks1;
    const uint32_t max_split = (ttiles + kStage - 1) / kStage;  // at least one LDS stage per block
    if (tsplit > max_split) tsplit = max_split;
    if (tsplit > 32) tsplit = 32;
    const uint32_t fit = key_planes_capacity / n_pairs;
    if (tsplit > fit) tsplit = fit;
    if (tsplit < 1) tsplit = 1;
    g.tsplit = tsplit;
  }
  return g;
}

uint32_t launch_hamming_mfma(const uint32_t* slab, const PairWork* work, uint32_t* keys, uint32_t max_kp,
                             uint32_t n_pairs, HammingGeometry geom, int mode, hipStream_t stream) {
  if (n_pairs == 0 || geom.qblocks == 0) return 1;
  const uint32_t qblocks = geom.qblocks, tsplit = geom.tsplit;
  const uint32_t tiles_per_slot = hamming_mfma_tiles_per_slot(max_kp);
  const uint32_t pairs8 = (n_pairs + 7u) / 8u * 8u;
  const uint32_t grid = pairs8 * qblocks * tsplit;
  const uint4* s4 = reinterpret_cast<const uint4*>(slab);
#define RGBDFE_LAUNCH_HM(M, S)                                                                                   \
  hipLaunchKernelGGL((hamming_mfma_kernel<M, S>), dim3(grid), dim3(kThreads), 0, stream, s4, work, keys, max_kp, \
                     tiles_per_slot, n_pairs, qblocks, tsplit)
  if (mode == 3) {
    if (tsplit > 1)
      hipLaunchKernelGGL((hamming_mfma_pipe_kernel<true>), dim3(grid), dim3(kThreads), 0, stream, s4, work, keys, max_kp,
                         tiles_per_slot, n_pairs, qblocks, tsplit);
    else
      hipLaunchKernelGGL((hamming_mfma_pipe_kernel<false>), dim3(grid), dim3(kThreads), 0, stream, s4, work, keys, max_kp,
                         tiles_per_slot, n_pairs, qblocks, tsplit);
  } else if (mode == 2) {
    if (tsplit > 1) RGBDFE_LAUNCH_HM(2, true); else RGBDFE_LAUNCH_HM(2, false);
  } else {
    if (tsplit > 1) RGBDFE_LAUNCH_HM(1, true); else RGBDFE_LAUNCH_HM(1, false);
  }
#undef RGBDFE_LAUNCH_HM
  return tsplit;
}

}  // namespace rgbdfe
