// sift_match.hip -- 128-d float-descriptor (SIFT) matcher on the bf16 matrix cores (gfx950).
//
// Replaces SiftGPUWrapper::match (src/sift_gpu_wrapper.cpp:169-227) over SiftMatchGPU, whose
// CUDA back end is the behavioural reference:
//   SiftMatchCU.cpp:87-100   u8 quantisation  pub[i] = int(512*f + 0.5)
//   ProgramCU.cu:1405-1482   MultiplyDescriptor_Kernel: u8 dot-product matrix + per-8-row column
//                            (max, argmax, second) partials
//   ProgramCU.cu:1689-1743   RowMatch_Kernel: per query row best / second best / acos tests
//   ProgramCU.cu:1764-1782   ColMatch_Kernel: per train column best row
//   SiftMatchCU.cpp:148-177  GetBestMatch: mutual best
//
// The dot-product matrix is the one dense contraction of the front end, so it runs on MFMA:
// the u8 values (0..255) are exactly representable in bf16 and a 128-term sum of u8*u8
// products is < 2^24, so v_mfma_f32_32x32x16_bf16 computes the INTEGER dot products exactly
// (fp32 accumulation never rounds).  The matrix is never written to memory: every 32x64
// accumulator chunk is reduced in registers to running (best, second best) keys per row.
// Keys pack (dot << 7 | 127 - sequence) so that "strict >, first wins" is one unsigned max;
// the cross-lane merge reproduces RowMatch_Kernel's 32-thread butterfly (lower thread wins
// ties).  The per-train-column result (ColMatch_Kernel, "lowest row wins") is a second pass of
// the same kernel with the operands swapped: recomputing the products on the matrix cores is
// cheaper than exchanging column partials through LDS and HBM.
//
// Two key formats.  INTEGER keys (sift_row_top2_kernel) hold for any input: dot < 2^23, up to 4096 rows.  FLOAT keys
// (sift_top2_fast_kernel) apply when the host has checked (PairWork::pad bit 0) that both nodes hold at most 1024 rows
// and that every descriptor's quantised squared norm is < 2^19 -- true for unit-length SIFT descriptors (512^2 = 2^18) --
// so that every dot product is < 2^19 by Cauchy-Schwarz: a ninth k-step adds (31 - sequence) / 32 to every accumulator
// element and the element IS the key (24 significant bits: exact in f32); non-negative floats order like their bit
// patterns, so the running top-2 is v_med3_u32 + v_max_u32 on the raw accumulator: 2 VALU per element instead of 4.
#include "rgbdfe_internal.h"

namespace rgbdfe {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kSiftDim = 128;
constexpr int kTile = 128;     // rows per block, columns per tile
constexpr int kSiftThreads = 256;

// running (best, second) with mx >= nx: the new second is the median of {mx, nx, key} (v_med3_u32), the new
// best their maximum -- two VALU operations per accumulator element
__device__ __forceinline__ void top2_insert(uint32_t& mx, uint32_t& nx, uint32_t key) {
  const uint32_t med = max(min(mx, nx), min(max(mx, nx), key));  // the shape the back end matches to v_med3_u32
  mx = max(mx, key);
  nx = med;
}

__device__ __forceinline__ int row_of_reg(int reg, int lane) {
  // v_mfma_f32_32x32x16 C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// u8 quantisation of float descriptors, stored as bf16 (exact) -- SiftMatchCU.cpp:96-99
__global__ void sift_quantise_kernel(const float* __restrict__ f, uint16_t* __restrict__ q, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    // int(512 * d + 0.5): float product, double add, truncation, then the unsigned char store
    const float prod = 512 * f[i];
    const int v = (int)((double)prod + 0.5);
    const unsigned char u = (unsigned char)v;
    const float uf = (float)u;                       // 0..255: exact in bf16 (8 significant bits)
    q[i] = (uint16_t)(__float_as_uint(uf) >> 16);
  }
}

void launch_sift_quantise(const float* f32, uint16_t* bf16, size_t n_elems, hipStream_t stream) {
  if (n_elems == 0) return;
  int blocks = (int)((n_elems + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sift_quantise_kernel, dim3(blocks), dim3(256), 0, stream, f32, bf16, n_elems);
}

// Best / second-best dot product of every row of node X against all rows of node Y.
//   SWAP = false: X = query (newer) node, Y = train node  -> RowMatch_Kernel's per-query result
//   SWAP = true : X = train node, Y = query node          -> ColMatch_Kernel's per-train-column result
// The second pass recomputes the (cheap, MFMA) dot products instead of exchanging per-tile column
// partials between waves through HBM: no partial buffers, no cross-wave reduction.
//
// Block = 4 waves = 128 rows of X (each wave keeps its 32 rows x K=128 as A fragments in 32 VGPRs).
// Y streams through LDS in 128-row tiles (32 KB), double buffered: the tile is fetched with fully
// coalesced 16-byte loads (a wave reads 4 whole 256-byte rows per instruction), written with an XOR
// swizzle on the 16-byte chunk index (chunk ^ (row & 15)) and read back as MFMA B fragments with
// conflict-free ds_read_b128 (the 16 lanes of a read group hit 16 different slots of the 256-byte
// bank row).  Per tile and wave: 32 x v_mfma_f32_32x32x16_bf16 and a 4-op running top-2 update per
// accumulator element (v_cvt_i32_f32, v_lshl_or_b32, v_med3_u32, v_max_u32); the MFMAs write VGPRs
// (-mllvm -amdgpu-mfma-vgpr-form, see the Makefile), so there is no v_accvgpr_read.
// part: [pair][max_kp][3] = (best dot, second dot, best index or 0xFFFFFFFF)
constexpr int kChunksPerRow = 16;  // 256 B / 16 B

// Workgroup -> (pair, row block).  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2; every
// row block of a pair streams the WHOLE other node (256 KB at 1000 rows), so the row blocks of one pair must land on ONE
// XCD -- one HBM / Infinity-Cache read, the rest L2 hits -- instead of one per XCD (measured: 7.5 GB fetched per launch of
// 4000 pairs with the plain (row block, pair) grid, 8x the data).  XCD x takes pairs x, x + 8, ...; a pair's row blocks
// are consecutive in its XCD's dispatch order.  Grid = n_rb * 8 * ceil(n_pairs / 8) workgroups.
__device__ __forceinline__ void sift_block_to_tile(uint32_t n_rb, uint32_t& pair, uint32_t& rb) {
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  pair = (j / n_rb) * 8u + xcd;
  rb = j % n_rb;
}

template <bool SWAP>
__global__ __launch_bounds__(kSiftThreads) void sift_row_top2_kernel(
    const uint16_t* __restrict__ bf16_pool, const PairWork* __restrict__ work, uint32_t max_kp,
    uint32_t n_pairs, uint32_t n_rb, uint32_t* __restrict__ part) {
  __shared__ uint4 tileY[2][kTile * kChunksPerRow];
  uint32_t pair, rb;
  sift_block_to_tile(n_rb, pair, rb);
  if (pair >= n_pairs) return;
  const PairWork w = work[pair];
  const int nq = (int)min(w.nq, 4096u), nt = (int)min(w.nt, 4096u);  // sift_gpu_wrapper.cpp:231
  const int nx = SWAP ? nt : nq, ny = SWAP ? nq : nt;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((int)(rb * kTile) >= nx || (w.pad & 1u)) return;  // block-uniform; pad bit 0: sift_top2_fast_kernel's pair
  const int r0 = rb * kTile + wv * 32;

  const uint16_t* __restrict__ xpool = bf16_pool + (size_t)(SWAP ? w.t_slot : w.q_slot) * max_kp * kSiftDim;
  const uint16_t* __restrict__ ypool = bf16_pool + (size_t)(SWAP ? w.q_slot : w.t_slot) * max_kp * kSiftDim;

  // A fragments: this wave's 32 rows of X, all of K = 128 (8 k-steps x 8 bf16 per lane)
  bf16x8 A[8];
  {
    int row = r0 + (lane & 31);
    row = row < nx ? row : nx - 1;
    row = row < 0 ? 0 : row;
    const uint4* src = reinterpret_cast<const uint4*>(xpool + (size_t)row * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A[ks] = __builtin_bit_cast(bf16x8, src[ks * 2]);
  }

  // running (best, second) keys per owned row; key = dot << 7 | (127 - sequence): the sequence
  // number of a column inside this lane (col >> 5) grows with the column, so "strict >, first
  // wins" (ProgramCU.cu:1464-1467, :1715-1719) is a plain unsigned max.  dot < 2^23.
  uint32_t rmx[16], rnx[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rmx[r] = rnx[r] = 0u;

  const int n_tiles = (ny + kTile - 1) / kTile;
  const uint4* __restrict__ ysrc = reinterpret_cast<const uint4*>(ypool);
  // this thread's 8 chunks of a tile: global chunk g = i * 256 + tid -> (row g >> 4, chunk g & 15)
#define SIFT_LOAD_TILE(TILE, REGS)                                      \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                       \
    const int g = i * kSiftThreads + tid;                               \
    int row = (TILE) * kTile + (g >> 4);                                \
    row = row < ny ? row : ny - 1;                                      \
    row = row < 0 ? 0 : row; /* ny == 0: nothing is used, but the load is unconditional */ \
    REGS[i] = ysrc[(size_t)row * kChunksPerRow + (g & 15)];             \
  }
#define SIFT_STORE_TILE(BUF, REGS)                                      \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                       \
    const int g = i * kSiftThreads + tid;                               \
    const int row = g >> 4, chunk = g & 15;                             \
    tileY[BUF][row * kChunksPerRow + (chunk ^ (row & 15))] = REGS[i];   \
  }
  {
    uint4 first[8];
    SIFT_LOAD_TILE(0, first)
    SIFT_STORE_TILE(0, first)
  }
  __syncthreads();

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int buf = tile & 1;
    uint4 nxt[8];
    const bool more = tile + 1 < n_tiles;
    {
      const int tn = more ? tile + 1 : tile;  // unconditional: keeps the staging registers out of scratch
      SIFT_LOAD_TILE(tn, nxt)
    }
    const int t0 = tile * kTile;
    const bool full = t0 + kTile <= ny;
    // The 4 column tiles of the Y tile, software pipelined: the 8 MFMAs of column tile ct+1 are issued before
    // the top-2 epilogue of column tile ct, so the matrix pipe works on ct+1 while the VALU digests ct
    // (two accumulators = 32 registers in flight).
    auto dots = [&](int ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const int row = ct * 32 + (lane & 31);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int chunk = ks * 2 + (lane >> 5);
        const bf16x8 B = __builtin_bit_cast(bf16x8, tileY[buf][row * kChunksPerRow + (chunk ^ (row & 15))]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], B, acc, 0, 0, 0);
      }
      return acc;
    };
    // full tiles: no per-element select, one straight-line block so that the scheduler can interleave the
    // MFMAs of ct+1 with the VALU work of ct; the ragged last tile zeroes the keys of out-of-range columns
    auto digest_full = [&](const f32x16& acc, int ct) {
      const uint32_t lo = 127u - (uint32_t)(tile * 4 + ct);  // dot == 0 -> key < 128: inert
#pragma unroll
      for (int r = 0; r < 16; ++r) top2_insert(rmx[r], rnx[r], ((uint32_t)(int)acc[r] << 7) | lo);
    };
    auto digest_ragged = [&](const f32x16& acc, int ct) {
      const uint32_t lo = 127u - (uint32_t)(tile * 4 + ct);
      const bool ok = (t0 + ct * 32 + (lane & 31)) < ny;
#pragma unroll
      for (int r = 0; r < 16; ++r) top2_insert(rmx[r], rnx[r], ok ? (((uint32_t)(int)acc[r] << 7) | lo) : 0u);
    };
    // scheduling hint for one (dots(ct+1), digest(ct)) pair: 8 x { 1 MFMA, 10 VALU }
#define SIFT_INTERLEAVE()                                             \
  _Pragma("unroll") for (int g = 0; g < 8; ++g) {                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);                \
  }
    if (full) {
      const f32x16 a0 = dots(0);
      const f32x16 a1 = dots(1);
      digest_full(a0, 0);
      SIFT_INTERLEAVE()
      const f32x16 a2 = dots(2);
      digest_full(a1, 1);
      SIFT_INTERLEAVE()
      const f32x16 a3 = dots(3);
      digest_full(a2, 2);
      SIFT_INTERLEAVE()
      digest_full(a3, 3);
    } else {
      const f32x16 a0 = dots(0);
      const f32x16 a1 = dots(1);
      digest_ragged(a0, 0);
      const f32x16 a2 = dots(2);
      digest_ragged(a1, 1);
      const f32x16 a3 = dots(3);
      digest_ragged(a2, 2);
      digest_ragged(a3, 3);
    }
    if (more) {
      SIFT_STORE_TILE(buf ^ 1, nxt)
    }
    __syncthreads();
  }
  if (r0 >= nx) return;  // waves beyond the last row (after the last barrier)

  // ---- merge the 32 lanes that share a row.
  // !SWAP: RowMatch_Kernel's 32-thread butterfly (:1726-1736): slot t absorbs slot t+step, the
  //        lower slot wins ties on the dot value.
  //  SWAP: ColMatch_Kernel / MultiplyDescriptor_Kernel: the lowest query row wins ties.
  uint32_t* __restrict__ opart = part + (size_t)pair * max_kp * 3;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    uint32_t dmx = rmx[r] >> 7, dnx = rnx[r] >> 7;
    uint32_t idx = dmx ? (((127u - (rmx[r] & 127u)) << 5) | (uint32_t)(lane & 31)) : 0xFFFFFFFFu;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1) {
      const uint32_t pmx = __shfl_xor(dmx, step), pnx = __shfl_xor(dnx, step), pidx = __shfl_xor(idx, step);
      const bool i_am_low = ((lane & step) == 0);
      const uint32_t v1 = i_am_low ? dmx : pmx, n1 = i_am_low ? dnx : pnx, i1 = i_am_low ? idx : pidx;
      const uint32_t v2 = i_am_low ? pmx : dmx, n2 = i_am_low ? pnx : dnx, i2 = i_am_low ? pidx : idx;
      const bool test = SWAP ? (v2 > v1 || (v2 == v1 && i2 < i1)) : (v2 > v1);
      dnx = test ? max(v1, n2) : max(n1, v2);
      idx = test ? i2 : i1;
      dmx = test ? v2 : v1;
    }
    const int row = r0 + row_of_reg(r, lane);
    if ((lane & 31) == 0 && row < nx) {
      opart[(size_t)row * 3 + 0] = dmx;
      opart[(size_t)row * 3 + 1] = dnx;
      opart[(size_t)row * 3 + 2] = idx;
    }
  }
}

// ---- merge of the 32 lanes that share a row (float keys) and the store of (best dot, second dot, best index)
template <bool SWAP>
__device__ __forceinline__ void sift_merge_store(const uint32_t (&rmx)[16], const uint32_t (&rnx)[16], int r0, int nx,
                                                 int lane, uint32_t* __restrict__ opart) {
  // Float keys leave room for the lane in the key (dot < 2^19): the tie rules become ONE total order --
  //   !SWAP: (dot, lower lane, lower sequence)   RowMatch_Kernel's butterfly prefers the lower slot at every step, so the
  //                                              lowest lane among equal dots survives; inside a lane the first column
  //    SWAP: (dot, lower sequence, lower lane)   = the lowest row index (ColMatch_Kernel)
  // -- so the 32 lanes that share a row merge by a plain max in any pairing, with second = max(loser's best, both
  // seconds) per step (v_min, v_max3, v_max).  The merge is a reduce-scatter: at every step a lane keeps half of its
  // rows and hands the other half to its partner (16 -> 8 -> 4 -> 2 -> 1 rows; xor 1 / 2 as DPP quad permutes, 4 / 8 /
  // 16 through ds_bpermute), 31 merges per lane instead of 80, and ends holding ONE row, which it stores.
  const uint32_t lrev = 31u - (uint32_t)(lane & 31);
  uint32_t K[16], S[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t k32 = (uint32_t)(__uint_as_float(rmx[r]) * 32.0f);  // dot << 5 | (31 - seq): exact, < 2^24
    const uint32_t n32 = (uint32_t)(__uint_as_float(rnx[r]) * 32.0f);
    K[r] = SWAP ? ((k32 << 5) | lrev) : (((k32 >> 5) << 10) | (lrev << 5) | (k32 & 31u));
    S[r] = (n32 >> 5) << 10;
  }
#define SIFT_MERGE(KEY, SEC, PK, PS)                     \
  {                                                      \
    const uint32_t pk = (PK), ps = (PS);                 \
    SEC = max(max(min(KEY, pk), SEC), ps);               \
    KEY = max(KEY, pk);                                  \
  }
  // STEP(N, bit, fetch): N rows -> N / 2; a lane with `bit` set keeps the upper half
#define SIFT_SCATTER_STEP(N, BIT, FETCH)                                              \
  _Pragma("unroll") for (int r = 0; r < (N) / 2; ++r) {                                \
    const uint32_t sendK = (BIT) ? K[r] : K[r + (N) / 2], sendS = (BIT) ? S[r] : S[r + (N) / 2]; \
    uint32_t keepK = (BIT) ? K[r + (N) / 2] : K[r], keepS = (BIT) ? S[r + (N) / 2] : S[r];       \
    SIFT_MERGE(keepK, keepS, FETCH(sendK), FETCH(sendS))                               \
    K[r] = keepK;                                                                      \
    S[r] = keepS;                                                                      \
  }
#define SIFT_DPP_XOR1(V) __builtin_amdgcn_update_dpp(0u, (V), 0xB1, 0xF, 0xF, false)  // quad_perm [1,0,3,2]
#define SIFT_DPP_XOR2(V) __builtin_amdgcn_update_dpp(0u, (V), 0x4E, 0xF, 0xF, false)  // quad_perm [2,3,0,1]
#define SIFT_SHFL4(V) __shfl_xor((V), 4)
#define SIFT_SHFL8(V) __shfl_xor((V), 8)
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
  SIFT_SCATTER_STEP(16, b0, SIFT_DPP_XOR1)
  SIFT_SCATTER_STEP(8, b1, SIFT_DPP_XOR2)
  SIFT_SCATTER_STEP(4, b2, SIFT_SHFL4)
  SIFT_SCATTER_STEP(2, b3, SIFT_SHFL8)
  uint32_t key = K[0], sec = S[0];
  SIFT_MERGE(key, sec, __shfl_xor(key, 16), __shfl_xor(sec, 16))
#undef SIFT_MERGE
#undef SIFT_SCATTER_STEP
#undef SIFT_DPP_XOR1
#undef SIFT_DPP_XOR2
#undef SIFT_SHFL4
#undef SIFT_SHFL8
  const int reg = (b0 ? 8 : 0) + (b1 ? 4 : 0) + (b2 ? 2 : 0) + (b3 ? 1 : 0);  // the row this lane ended up with
  const uint32_t dmx = key >> 10, dnx = sec >> 10;
  const uint32_t hi5 = 31u - ((key >> 5) & 31u), lo5 = 31u - (key & 31u);
  const uint32_t idx = dmx ? (SWAP ? ((hi5 << 5) | lo5) : ((lo5 << 5) | hi5)) : 0xFFFFFFFFu;
  const int row = r0 + row_of_reg(reg, lane);
  if ((lane & 16) == 0 && row < nx) {
    opart[(size_t)row * 3 + 0] = dmx;
    opart[(size_t)row * 3 + 1] = dnx;
    opart[(size_t)row * 3 + 2] = idx;
  }
}

// ---- float keys -------------------------------------------------------------------------------------------------------
// Same block shape and LDS tile as above.  The inner loop is software pipelined by hand over the column tiles (32 columns
// = 8 B fragments = 9 MFMAs): while the matrix pipe works on column tile c, the VALU digests the accumulator of c - 1
// (32 instructions: 4 per 32-cycle MFMA slot) and the LDS returns the B fragments of c + 1; the last accumulator of a
// Y tile is digested beside the first MFMAs of the next one, across the barrier.  The ragged last Y tile (columns beyond
// ny) runs unpipelined with a select per element.
template <bool SWAP>
__global__ __launch_bounds__(kSiftThreads) void sift_top2_fast_kernel(
    const uint16_t* __restrict__ bf16_pool, const PairWork* __restrict__ work, uint32_t max_kp,
    uint32_t n_pairs, uint32_t n_rb, uint32_t* __restrict__ part) {
  __shared__ u32x4 tileY[2][kTile * kChunksPerRow];  // native vectors: the staging array is promoted to registers
  uint32_t pair, rb;
  sift_block_to_tile(n_rb, pair, rb);
  if (pair >= n_pairs) return;
  const PairWork w = work[pair];
  const int nq = (int)w.nq, nt = (int)w.nt;  // <= 1024 (host)
  const int nx = SWAP ? nt : nq, ny = SWAP ? nq : nt;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((int)(rb * kTile) >= nx || !(w.pad & 1u)) return;  // block-uniform
  const int r0 = rb * kTile + wv * 32;

  const uint16_t* __restrict__ xpool = bf16_pool + (size_t)(SWAP ? w.t_slot : w.q_slot) * max_kp * kSiftDim;
  const uint16_t* __restrict__ ypool = bf16_pool + (size_t)(SWAP ? w.q_slot : w.t_slot) * max_kp * kSiftDim;

  bf16x8 A[8];
  {
    int row = r0 + (lane & 31);
    row = row < nx ? row : nx - 1;
    row = row < 0 ? 0 : row;
    const u32x4* src = reinterpret_cast<const u32x4*>(xpool + (size_t)row * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A[ks] = __builtin_bit_cast(bf16x8, src[ks * 2]);
  }
  // ninth k-step: A has 1.0 at k = 0 (lanes 0..31 hold k = 0..7), B the sequence term at k = 0
  const uint32_t k0 = (lane < 32) ? 0xFFFFFFFFu : 0u;
  const bf16x8 A9 = __builtin_bit_cast(bf16x8, u32x4{0x3F80u & k0, 0u, 0u, 0u});
  auto seq_term = [&](int seq) {  // (31 - seq) / 32 as bf16 (at most 5 significant bits: exact)
    const float term = (float)(31 - seq) * 0.03125f;
    return __builtin_bit_cast(bf16x8, u32x4{(__float_as_uint(term) >> 16) & k0, 0u, 0u, 0u});
  };

  uint32_t rmx[16], rnx[16];  // float bit patterns of (dot + (31 - seq) / 32)
#pragma unroll
  for (int r = 0; r < 16; ++r) rmx[r] = rnx[r] = 0u;

  const int n_tiles = (ny + kTile - 1) / kTile;
  const int n_full = ny / kTile;
  const u32x4* __restrict__ ysrc = reinterpret_cast<const u32x4*>(ypool);
  // a tile is 2048 consecutive 16-byte chunks: thread tid moves chunks i * 256 + tid.  No row clamp: the pool is padded
  // (ensure_sift) and rows beyond ny are masked in the ragged tile
#define SIFT_LOAD_TILE_LINEAR(TILE, REGS) \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) REGS[i] = ysrc[(size_t)(TILE) * (kTile * kChunksPerRow) + i * kSiftThreads + tid];
  {
    u32x4 first[8];
    SIFT_LOAD_TILE_LINEAR(0, first)
    SIFT_STORE_TILE(0, first)
  }
  __syncthreads();

  // B fragment slots of this lane inside a column tile: row = ct * 32 + (lane & 31), chunk (ks * 2 + hi) ^ (row & 15)
  int boff[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    boff[ks] = (lane & 31) * kChunksPerRow + ((ks * 2 + (lane >> 5)) ^ (lane & 15));

  f32x16 accX, accY;  // accY enters a tile holding the previous tile's last column tile (zeros: inert keys)
#pragma unroll
  for (int r = 0; r < 16; ++r) accX[r] = accY[r] = 0.0f;
  bf16x8 Bc[8], Bn[8];

#define SIFT_READ_B(DST, BUF, CT)                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                   \
      DST[ks] = __builtin_bit_cast(bf16x8, tileY[BUF][(CT) * 32 * kChunksPerRow + boff[ks]]);
#define SIFT_MFMA9(ACC, B, SEQ)                                                                     \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[r] = 0.0f;                                      \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                   \
      ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], B[ks], ACC, 0, 0, 0);                     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A9, seq_term(SEQ), ACC, 0, 0, 0);
#define SIFT_DIGEST(ACC)                                                                            \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) top2_insert(rmx[r], rnx[r], __float_as_uint(ACC[r]));
#if defined(RGBDFE_SIFT_ABL)  // timing ablations (wrong results): tools/sift_ablation.sh
#if RGBDFE_SIFT_ABL == 1      // no digest (one element keeps the accumulator alive)
#undef SIFT_DIGEST
#define SIFT_DIGEST(ACC) top2_insert(rmx[0], rnx[0], __float_as_uint(ACC[0]));
#elif RGBDFE_SIFT_ABL == 2    // no MFMA
#undef SIFT_MFMA9
#define SIFT_MFMA9(ACC, B, SEQ)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(B[ks]));                    \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[r] = __uint_as_float((uint32_t)((SEQ) + r + lane));
#elif RGBDFE_SIFT_ABL == 6 || RGBDFE_SIFT_ABL == 7   // no MFMA, no digest (7: no LDS reads either)
#undef SIFT_MFMA9
#define SIFT_MFMA9(ACC, B, SEQ)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(B[ks]));                    \
  ACC[0] = __uint_as_float((uint32_t)((SEQ) + lane));
#undef SIFT_DIGEST
#define SIFT_DIGEST(ACC) top2_insert(rmx[0], rnx[0], __float_as_uint(ACC[0]));
#if RGBDFE_SIFT_ABL == 7
#undef SIFT_READ_B
#define SIFT_READ_B(DST, BUF, CT) _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) DST[ks] = A[(ks + (CT)) & 7];
#endif
#elif RGBDFE_SIFT_ABL == 5    // no LDS reads
#undef SIFT_READ_B
#define SIFT_READ_B(DST, BUF, CT) _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) DST[ks] = A[(ks + (CT)) & 7];
#endif
#endif
  // one pipelined step: 9 x { 1 MFMA, 1 LDS read (the first 8), 4 VALU }, fenced so that nothing crosses into the next step
#ifdef RGBDFE_SIFT_BURST
#define SIFT_STEP_SCHED(N_READS)                                       \
  __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);                   \
  __builtin_amdgcn_sched_group_barrier(0x100, (N_READS), 0);           \
  __builtin_amdgcn_sched_group_barrier(0x002, 36, 0);                  \
  __builtin_amdgcn_sched_barrier(0);
#else
#define SIFT_STEP_SCHED(N_READS)                                       \
  _Pragma("unroll") for (int g = 0; g < 9; ++g) {                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    if (g < (N_READS)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                 \
  }                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#endif

  int tile = 0;
  for (; tile < n_full; ++tile) {
    const int buf = tile & 1;
    u32x4 nxt[8];
    SIFT_LOAD_TILE_LINEAR(tile + 1, nxt)
    const int seq0 = tile * 4;
    SIFT_READ_B(Bc, buf, 0)
    __builtin_amdgcn_sched_barrier(0);
    // column tile 0 beside the digest of the previous tile's column tile 3
    SIFT_MFMA9(accX, Bc, seq0)
    SIFT_READ_B(Bn, buf, 1)
    SIFT_DIGEST(accY)
    SIFT_STEP_SCHED(8)
    SIFT_MFMA9(accY, Bn, seq0 + 1)
    SIFT_READ_B(Bc, buf, 2)
    SIFT_DIGEST(accX)
    SIFT_STEP_SCHED(8)
    SIFT_MFMA9(accX, Bc, seq0 + 2)
    SIFT_READ_B(Bn, buf, 3)
    SIFT_DIGEST(accY)
    SIFT_STEP_SCHED(8)
    SIFT_MFMA9(accY, Bn, seq0 + 3)
    SIFT_DIGEST(accX)
    SIFT_STEP_SCHED(0)
    // unconditional (after the last tile the other buffer is never read again): the loop body stays ONE basic block, so
    // the digests cannot be sunk below the MFMAs into a latch block
#if defined(RGBDFE_SIFT_ABL) && (RGBDFE_SIFT_ABL == 3 || RGBDFE_SIFT_ABL == 4)
    asm volatile("" ::"v"(nxt[0]), "v"(nxt[7]));
#else
    SIFT_STORE_TILE(buf ^ 1, nxt)
#endif
#if !(defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 4)
    __syncthreads();
#endif
  }
  SIFT_DIGEST(accY)
  if (tile < n_tiles) {  // ragged last tile: columns beyond ny carry key 0
    const int buf = tile & 1;
    const int t0 = tile * kTile;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      SIFT_READ_B(Bc, buf, ct)
      SIFT_MFMA9(accX, Bc, tile * 4 + ct)
      const bool ok = (t0 + ct * 32 + (lane & 31)) < ny;
#pragma unroll
      for (int r = 0; r < 16; ++r) top2_insert(rmx[r], rnx[r], ok ? __float_as_uint(accX[r]) : 0u);
    }
  }
#undef SIFT_READ_B
#undef SIFT_LOAD_TILE_LINEAR
#undef SIFT_MFMA9
#undef SIFT_DIGEST
#undef SIFT_STEP_SCHED
  if (r0 >= nx) return;  // waves beyond the last row (after the last barrier)
#if defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 8
  if (rmx[0] != 0x12345u) return;
#endif

  sift_merge_store<SWAP>(rmx, rnx, r0, nx, lane, part + (size_t)pair * max_kp * 3);
}

// ---- float keys, 64 rows per wave -----------------------------------------------------------------------------------------
// The 32-row kernel above reads one 16-byte B fragment from LDS per MFMA: four SIMDs at the matrix pipe's rate ask for
// exactly the LDS port's 128 B per clock, so neither unit gets past ~half busy.  Here a wave holds 64 rows of X (two A
// fragment sets) and every B fragment feeds TWO MFMAs; block = 4 waves = 256 rows.  The registers for the second row group
// come from the tile staging: the Y tile goes global -> LDS directly (global_load_lds_dwordx4: destination = wave-uniform
// base + lane * 16, so the XOR swizzle of the 16-byte chunks is applied to the SOURCE address and to the ds_read side).
// Pipelining: the 9 MFMAs of (row group 0, column tile c) run beside the digest of (group 1, c - 1); those of (group 1, c)
// beside the digest of (group 0, c); the B fragments of c + 1 arrive during both.
constexpr int kTile64 = 256;

template <bool SWAP>
__global__ __launch_bounds__(kSiftThreads, 2) void sift_top2_fast64_kernel(
    const uint16_t* __restrict__ bf16_pool, const PairWork* __restrict__ work, uint32_t max_kp,
    uint32_t n_pairs, uint32_t n_rb, uint32_t* __restrict__ part) {
  __shared__ u32x4 tileY[2][kTile * kChunksPerRow];
  uint32_t pair, rb;
  sift_block_to_tile(n_rb, pair, rb);
  if (pair >= n_pairs) return;
  const PairWork w = work[pair];
  const int nq = (int)w.nq, nt = (int)w.nt;  // <= 1024 (host)
  const int nx = SWAP ? nt : nq, ny = SWAP ? nq : nt;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)(rb * kTile64) >= nx || !(w.pad & 1u)) return;  // block-uniform
  const int r0 = rb * kTile64 + wv * 64;

  const uint16_t* __restrict__ xpool = bf16_pool + (size_t)(SWAP ? w.t_slot : w.q_slot) * max_kp * kSiftDim;
  const uint16_t* __restrict__ ypool = bf16_pool + (size_t)(SWAP ? w.q_slot : w.t_slot) * max_kp * kSiftDim;
  const int n_tiles = (ny + kTile - 1) / kTile;
  const int n_full = ny / kTile;

  // Tile staging: wave wv owns LDS slots [wv * 512, wv * 512 + 512) of the 2048, 64 per instruction.  Slot p = row * 16 + c
  // holds the row's chunk c ^ (row & 15); row & 15 = (i * 4 + lane / 16) & 15 for instruction i, i.e. the source chunk of a
  // lane is ((lane & 15) ^ (lane >> 4)) ^ ((i & 3) * 4) in its row.  No row clamp: the pool is padded (ensure_sift).
  // The instruction's immediate offset applies to BOTH addresses, so the 8 instructions of a tile share one LDS base (M0)
  // and four 32-bit source offsets (13-bit signed immediate: both bases sit 4 KB into the wave's 8 KB).
  const int sw_lane = (lane & ~15) | ((lane & 15) ^ (lane >> 4));
  uint32_t voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) voff[j] = (uint32_t)((wv * 512 + (sw_lane ^ (j * 4))) * 16 + 4096);
  const uint64_t ybase = reinterpret_cast<uint64_t>(ypool);
  const uint32_t yb_lo = __builtin_amdgcn_readfirstlane((uint32_t)ybase);
  const uint32_t yb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ybase >> 32));
  // (the offset is made opaque where it is used: hoisted out of the loop as a zero-extended pair it would cost 16 registers
  // and the scalar-base addressing mode)
#define S64_GLDS(TB, BUF, I)                                                                                \
  {                                                                                                         \
    uint32_t vo = voff[(I) & 3];                                                                            \
    asm volatile("" : "+v"(vo));                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((TB) + vo),            \
                                     (__attribute__((address_space(3))) void*)&tileY[BUF][wv * 512 + 256], 16, \
                                     (I) * 1024 - 4096, 0);                                                 \
  }
#define S64_STAGE_TILE(TILE, BUF)                                                                           \
  {                                                                                                         \
    const char* tb = reinterpret_cast<const char*>((((uint64_t)yb_hi << 32) | yb_lo) +                      \
                                                   (uint64_t)(uint32_t)(TILE) * (kTile * kChunksPerRow * 16)); \
    S64_GLDS(tb, BUF, 0) S64_GLDS(tb, BUF, 1) S64_GLDS(tb, BUF, 2) S64_GLDS(tb, BUF, 3)                     \
    S64_GLDS(tb, BUF, 4) S64_GLDS(tb, BUF, 5) S64_GLDS(tb, BUF, 6) S64_GLDS(tb, BUF, 7)                     \
  }
  S64_STAGE_TILE(0, 0)

  if (r0 >= nx) {  // a wave without rows only stages its share of the tiles (same barriers as the working waves)
    __syncthreads();
    for (int tile = 0; tile < n_full; ++tile) {
      if (tile & 1) { S64_STAGE_TILE(tile + 1, 0) } else { S64_STAGE_TILE(tile + 1, 1) }
      __syncthreads();
    }
    return;
  }

  bf16x8 A0[8], A1[8];
  {
    int row = r0 + (lane & 31);
    row = row < nx ? row : nx - 1;
    const u32x4* src = reinterpret_cast<const u32x4*>(xpool + (size_t)row * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A0[ks] = __builtin_bit_cast(bf16x8, src[ks * 2]);
    row = r0 + 32 + (lane & 31);
    row = row < nx ? row : nx - 1;
    src = reinterpret_cast<const u32x4*>(xpool + (size_t)row * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A1[ks] = __builtin_bit_cast(bf16x8, src[ks * 2]);
  }
  const uint32_t k0 = (lane < 32) ? 0xFFFFFFFFu : 0u;
  const bf16x8 A9 = __builtin_bit_cast(bf16x8, u32x4{0x3F80u & k0, 0u, 0u, 0u});
  auto seq_term = [&](int seq) {
    const float term = (float)(31 - seq) * 0.03125f;
    return __builtin_bit_cast(bf16x8, u32x4{(__float_as_uint(term) >> 16) & k0, 0u, 0u, 0u});
  };

  uint32_t mx0[16], sx0[16], mx1[16], sx1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) mx0[r] = sx0[r] = mx1[r] = sx1[r] = 0u;
  __syncthreads();

  // B fragment (ks, column tile ct, buffer) of this lane: row = ct * 32 + (lane & 31), chunk (ks * 2 + hi) ^ (row & 15), i.e.
  // byte (row * 256 + ((hi ^ (lane & 15)) << 4)) ^ (ks << 5), + ct * 8192 + buffer * 32768 as the instruction's immediate
  // (the full-tile loop is unrolled by two, so the buffer is a constant there).  ONE address register: the XOR is an asm
  // statement per read, which keeps it from being hoisted out of the loop into 8 registers.
  const uint32_t abyte = (uint32_t)((lane & 31) * 256 + ((((lane >> 5) ^ lane) & 15) << 4));
  const char* lds_bytes = reinterpret_cast<const char*>(&tileY[0][0]);

  f32x16 acc0, acc1;  // acc1 enters a tile holding the previous tile's last column tile of row group 1 (zeros: inert keys)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
  bf16x8 Bf[8];

  // BOFF: buffer offset in bytes, a constant (full tiles) or a register (the ragged tile)
#define S64_READ_ONE(BOFF, CT, KS)                                                                  \
  {                                                                                                 \
    uint32_t a;                                                                                     \
    asm volatile("v_xor_b32 %0, %2, %1" : "=v"(a) : "v"(abyte), "n"((KS) << 5));                     \
    Bf[KS] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds_bytes + (a + (BOFF)) + (CT) * 8192)); \
  }
#define S64_READ_B(BOFF, CT)                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) S64_READ_ONE(BOFF, CT, ks)
#define S64_MFMA9(ACC, AF, SEQ)                                                                     \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[r] = 0.0f;                                      \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                   \
      ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[ks], Bf[ks], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A9, seq_term(SEQ), ACC, 0, 0, 0);
#if defined(RGBDFE_SIFT_ABL) && (RGBDFE_SIFT_ABL == 2 || RGBDFE_SIFT_ABL == 6)
#define S64_MFMA_ONE(ACC, A, B) asm volatile("" : "+v"(ACC) : "v"(A), "v"(B));
#else
#define S64_MFMA_ONE(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0);
#endif
  // OK: true (full tiles) or the lane's "column exists" flag of the column tile the accumulator belongs to
#define S64_DIGEST(ACC, MX, SX, OK)                                                                 \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) top2_insert(MX[r], SX[r], (OK) ? __float_as_uint(ACC[r]) : 0u);
#if defined(RGBDFE_SIFT_ABL)  // timing ablations (wrong results): tools/sift_ablation.sh
#if RGBDFE_SIFT_ABL == 1 || RGBDFE_SIFT_ABL == 6   // no digest (one element keeps the accumulator alive)
#undef S64_DIGEST
#define S64_DIGEST(ACC, MX, SX, OK) top2_insert(MX[0], SX[0], (OK) ? __float_as_uint(ACC[0]) : 0u);
#endif
#if RGBDFE_SIFT_ABL == 2 || RGBDFE_SIFT_ABL == 6   // no MFMA
#undef S64_MFMA9
#define S64_MFMA9(ACC, AF, SEQ)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(Bf[ks]), "v"(AF[ks]));      \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[r] = __uint_as_float((uint32_t)((SEQ) + r + lane));
#endif
#endif
  // One pipelined half step = the 9 MFMAs of one accumulator chain beside the digest of the OTHER chain's result.  That
  // result comes from the MFMA issued last in the previous half step and a wave issues in order, so a digest instruction
  // right behind it would hold the new chain back for the matrix pipe's latency: the first two MFMAs of a half step are
  // fenced off in front (HEAD), the digest runs beside the other seven (TAIL: 7 x { 1 MFMA, [1 address + 1 LDS read], NV VALU }).
#define S64_FENCE() __builtin_amdgcn_sched_barrier(0);
#define S64_TAIL_SCHED(WITH_READS, NV)                                  \
  _Pragma("unroll") for (int g = 0; g < 7; ++g) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
    if ((WITH_READS) && g < 6) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); \
    if ((WITH_READS) && g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);               \
  }                                                                     \
  S64_FENCE()
  // column tile CT: row group 0's MFMAs beside the digest of (group 1, CT - 1), then group 1's MFMAs beside the digest of
  // (group 0, CT).  ONE set of B fragments: fragment ks of column tile CT + 1 is read right behind the group-1 MFMA that is
  // the last user of fragment ks of CT (an MFMA reads its operands at issue), 9 MFMA slots before its own first use.
#define S64_PIN(ACC) asm volatile("" : "+v"(ACC));  // MFMAs and VALU are pure: only side effects keep them on their side of a fence
#define S64_COLUMN_TILE(BOFF, CT, SEQ, WITH_READS, OK_PREV, OK_THIS, NV)                            \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;                                     \
  S64_MFMA_ONE(acc0, A0[0], Bf[0])                                                                  \
  S64_MFMA_ONE(acc0, A0[1], Bf[1])                                                                  \
  S64_PIN(acc0)                                                                                     \
  S64_FENCE()                                                                                       \
  S64_PIN(acc1)                                                                                     \
  _Pragma("unroll") for (int ks = 2; ks < 8; ++ks) S64_MFMA_ONE(acc0, A0[ks], Bf[ks])                \
  S64_MFMA_ONE(acc0, A9, seq_term(SEQ))                                                             \
  S64_PIN(acc0)                                                                                     \
  S64_DIGEST(acc1, mx1, sx1, OK_PREV)                                                               \
  S64_TAIL_SCHED(false, NV)                                                                         \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                 \
    S64_MFMA_ONE(acc1, A1[ks], Bf[ks])                                                              \
    if (WITH_READS) S64_READ_ONE(BOFF, (CT) + 1, ks)                                                \
  }                                                                                                 \
  S64_PIN(acc1)                                                                                     \
  S64_FENCE()                                                                                       \
  S64_PIN(acc0)                                                                                     \
  _Pragma("unroll") for (int ks = 2; ks < 8; ++ks) {                                                 \
    S64_MFMA_ONE(acc1, A1[ks], Bf[ks])                                                              \
    if (WITH_READS) S64_READ_ONE(BOFF, (CT) + 1, ks)                                                \
  }                                                                                                 \
  S64_MFMA_ONE(acc1, A9, seq_term(SEQ))                                                             \
  S64_PIN(acc1)                                                                                     \
  S64_DIGEST(acc0, mx0, sx0, OK_THIS)                                                               \
  S64_TAIL_SCHED(WITH_READS, NV)
  // one full Y tile out of buffer BUF (a constant: the loop is unrolled by two so that every LDS offset is an immediate)
#if defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 9   // no tile staging, no barriers in the loop
#define S64_LOOP_STAGE(TILE, BUF)
#define S64_LOOP_BARRIER()
#else
#define S64_LOOP_STAGE(TILE, BUF) S64_STAGE_TILE(TILE, BUF)
#define S64_LOOP_BARRIER() __syncthreads();
#endif
#define S64_FULL_TILE(BUF)                                                                          \
  {                                                                                                 \
    S64_LOOP_STAGE(tile + 1, (BUF) ^ 1)                                                             \
    const int seq0 = tile * 4;                                                                      \
    S64_READ_B((BUF) * 32768, 0)                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    S64_COLUMN_TILE((BUF) * 32768, 0, seq0, true, true, true, 6)                                       \
    S64_COLUMN_TILE((BUF) * 32768, 1, seq0 + 1, true, true, true, 6)                                   \
    S64_COLUMN_TILE((BUF) * 32768, 2, seq0 + 2, true, true, true, 6)                                   \
    S64_COLUMN_TILE((BUF) * 32768, 3, seq0 + 3, false, true, true, 6)                                   \
    S64_LOOP_BARRIER()                                                                              \
    ++tile;                                                                                         \
  }

  int tile = 0;
  while (tile + 1 < n_full) {
    S64_FULL_TILE(0)
    S64_FULL_TILE(1)
  }
  if (tile < n_full) S64_FULL_TILE(0)   // n_full odd: `tile` is even here
  S64_DIGEST(acc1, mx1, sx1, true)
  if (tile < n_tiles) {  // ragged last tile: the same pipeline, columns beyond ny carry key 0
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
    const int c0 = tile * kTile + (lane & 31);
    const bool ok0 = c0 < ny, ok1 = c0 + 32 < ny, ok2 = c0 + 64 < ny, ok3 = c0 + 96 < ny;
    const uint32_t boff = (uint32_t)(tile & 1) * 32768u;
    const int seq0 = tile * 4;
    S64_READ_B(boff, 0)
    __builtin_amdgcn_sched_barrier(0);
    S64_COLUMN_TILE(boff, 0, seq0, true, true, ok0, 8)
    S64_COLUMN_TILE(boff, 1, seq0 + 1, true, ok0, ok1, 8)
    S64_COLUMN_TILE(boff, 2, seq0 + 2, true, ok1, ok2, 8)
    S64_COLUMN_TILE(boff, 3, seq0 + 3, false, ok2, ok3, 8)
    S64_DIGEST(acc1, mx1, sx1, ok3)
  }
#undef S64_FULL_TILE
#undef S64_LOOP_STAGE
#undef S64_LOOP_BARRIER
#undef S64_MFMA_ONE
#undef S64_READ_ONE
#undef S64_STAGE_TILE
#undef S64_GLDS
#undef S64_READ_B
#undef S64_MFMA9
#undef S64_DIGEST
#undef S64_TAIL_SCHED
#undef S64_PIN
#undef S64_FENCE
#undef S64_COLUMN_TILE
#if defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 8
  if (mx0[0] != 0x12345u && mx1[3] != 0x777u) return;
#endif
  uint32_t* __restrict__ opart = part + (size_t)pair * max_kp * 3;
  sift_merge_store<SWAP>(mx0, sx0, r0, nx, lane, opart);
  if (r0 + 32 < nx) sift_merge_store<SWAP>(mx1, sx1, r0 + 32, nx, lane, opart);
}

// ---- float keys, ONE pass ---------------------------------------------------------------------------------------------
// sift_top2_fast64_kernel<false> with the per-train-column result taken from the SAME sweep (the second
// pass with swapped operands ran every dot product twice).  What the column side needs is less than what the row side needs:
// ColMatch_Kernel (ProgramCU.cu:1764-1782) accepts column j only when dist(best) < 0.9 * dist(second); dist is a
// non-increasing function of the dot product, so an accepted column's best dot is STRICTLY above its second -- its
// argmax row is unique -- and "col_match[j] == i" for the row i whose best column is j (GetBestMatch, SiftMatchCU.cpp:165)
// is the same statement as "column j is accepted and dot(i, j) == best dot of column j".  So a column keeps only its two
// largest dot products, no row index, and the accumulator element itself is the key once more: within one column every
// element carries the same sequence term, i.e. the float bit patterns of a column order like its dot products.
//   in the sweep   a lane owns column (lane & 31) of the column tile and sees 16 rows of each of the wave's two row groups:
//                  2 more VALU per accumulator element (v_med3_u32 + v_max_u32 into one running pair per lane), one
//                  cross-half exchange and one 8-byte LDS store per column tile and wave;
//   per Y tile     the four waves' column partials (4 column tiles x 32 columns x 8 bytes each, double buffered by tile
//                  parity) are merged behind the tile's barrier -- wave w takes column tile w -- and leave as ONE record
//                  per column and 256-row block: col_blocks[pair][row block][slot][column in tile] = (best, second)
//                  float bits, slot = column tile + 1 (slot 0 takes the pipeline's empty first digest);
//   sift_finish    merges the <= 4 row blocks of a column, applies ColMatch's acceptance and the identity above.
// Rows beyond the node's last one enter as ZERO rows (dot 0: inert for the columns; their own results are never stored).
constexpr int kColSlots = 36;   // 32 column tiles of a 1024-row node + slot 0 + the merge steps' overrun past the last tile
// build-time variants of the one-pass kernel (tools/sweep_sift_onepass.sh): VALU slots per MFMA in the scheduling hint, the
// cross-half exchange as v_permlane32_swap instead of ds_bpermute, the column side as a tree over the accumulator
#ifndef RGBDFE_SIFT1_NV
#define RGBDFE_SIFT1_NV 11
#endif
#ifndef RGBDFE_SIFT1_SWAP
#define RGBDFE_SIFT1_SWAP 1
#endif
#ifndef RGBDFE_SIFT1_DIAG
#define RGBDFE_SIFT1_DIAG 0   // TIMING-ONLY builds (results are void): 1 no digest at all, 2 row side only, 4 column side only --
#endif                        // what the MFMA + LDS + barrier stream costs without (part of) its VALU work (tools/sweep_sift_onepass.sh diag)
#ifndef RGBDFE_SIFT1_TREE
#define RGBDFE_SIFT1_TREE 0
#endif
#ifndef RGBDFE_SIFT1_BURST
#define RGBDFE_SIFT1_BURST 0
#endif
// upper 32 lanes' value in the lower 32 lanes (and vice versa)
__device__ __forceinline__ uint32_t other_half(uint32_t v) {
#if RGBDFE_SIFT1_SWAP
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // lanes 32..63 of the first <-> lanes 0..31 of the second
  return (threadIdx.x & 32u) ? r[0] : r[1];
#else
  return (uint32_t)__shfl_xor((int)v, 32);
#endif
}
// (best, second) of three keys: 2 VALU; merge of two such pairs: 3 VALU
__device__ __forceinline__ void top2_of3(uint32_t a, uint32_t b, uint32_t c, uint32_t& m, uint32_t& n) {
  m = max(max(a, b), c);
  n = max(min(a, b), min(max(a, b), c));   // median of three (v_med3_u32)
}
__device__ __forceinline__ void top2_merge(uint32_t& m, uint32_t& n, uint32_t pm, uint32_t pn) {
  n = max(max(min(m, pm), n), pn);
  m = max(m, pm);
}

__global__ __launch_bounds__(kSiftThreads, 2) void sift_top2_onepass_kernel(
    const uint16_t* __restrict__ bf16_pool, const PairWork* __restrict__ work, uint32_t max_kp,
    uint32_t n_pairs, uint32_t n_rb, uint32_t* __restrict__ part, uint2* __restrict__ col_blocks) {
  __shared__ u32x4 tileY[2][kTile * kChunksPerRow];
  __shared__ uint2 colp[2][4][4][32];   // [tile parity][wave][column tile of the group][column]
  __shared__ uint2 colp_last[4][32];    // [wave][column]: the ragged tile's last column tile
  uint32_t pair, rb;
  sift_block_to_tile(n_rb, pair, rb);
  if (pair >= n_pairs) return;
  const PairWork w = work[pair];
  const int nx = (int)w.nq, ny = (int)w.nt;  // <= 1024 (host)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)(rb * kTile64) >= nx || !(w.pad & 1u)) return;  // block-uniform
  const int r0 = rb * kTile64 + wv * 64;
  const int n_active = min(4, (nx - (int)(rb * kTile64) + 63) / 64);   // waves of this block that hold rows

  const uint16_t* __restrict__ xpool = bf16_pool + (size_t)w.q_slot * max_kp * kSiftDim;
  const uint16_t* __restrict__ ypool = bf16_pool + (size_t)w.t_slot * max_kp * kSiftDim;
  const int n_tiles = (ny + kTile - 1) / kTile;
  const int n_full = ny / kTile;
  uint2* __restrict__ cblk = col_blocks + ((size_t)pair * 4 + rb) * (kColSlots * 32);

  // tile staging: see sift_top2_fast64_kernel
  const int sw_lane = (lane & ~15) | ((lane & 15) ^ (lane >> 4));
  uint32_t voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) voff[j] = (uint32_t)((wv * 512 + (sw_lane ^ (j * 4))) * 16 + 4096);
  const uint64_t ybase = reinterpret_cast<uint64_t>(ypool);
  const uint32_t yb_lo = __builtin_amdgcn_readfirstlane((uint32_t)ybase);
  const uint32_t yb_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ybase >> 32));
#define S1_GLDS(TB, BUF, I)                                                                                 \
  {                                                                                                         \
    uint32_t vo = voff[(I) & 3];                                                                            \
    asm volatile("" : "+v"(vo));                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((TB) + vo),            \
                                     (__attribute__((address_space(3))) void*)&tileY[BUF][wv * 512 + 256], 16, \
                                     (I) * 1024 - 4096, 0);                                                 \
  }
#define S1_STAGE_TILE(TILE, BUF)                                                                            \
  {                                                                                                         \
    const char* tb = reinterpret_cast<const char*>((((uint64_t)yb_hi << 32) | yb_lo) +                      \
                                                   (uint64_t)(uint32_t)(TILE) * (kTile * kChunksPerRow * 16)); \
    S1_GLDS(tb, BUF, 0) S1_GLDS(tb, BUF, 1) S1_GLDS(tb, BUF, 2) S1_GLDS(tb, BUF, 3)                         \
    S1_GLDS(tb, BUF, 4) S1_GLDS(tb, BUF, 5) S1_GLDS(tb, BUF, 6) S1_GLDS(tb, BUF, 7)                         \
  }
  // The column partials of group G (the column tiles whose digests ended between barrier G - 1 and barrier G: column tiles
  // 4 G - 1 .. 4 G + 2, slots 4 G .. 4 G + 3) -> col_blocks: wave wv merges slot 4 G + wv over the block's active waves.
  // Straight-line code for every lane: lanes 32 .. 63 repeat the work of lanes 0 .. 31 (same value to the same address),
  // slot 0 and slots past the node's last column tile receive whatever the buffer held -- nobody reads them.
#define S1_MERGE_GROUP(G)                                                                                   \
  {                                                                                                         \
    const int g_ = (G);                                                                                     \
    uint32_t m_ = 0u, n_ = 0u;                                                                              \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                         \
      const uint2 p = colp[g_ & 1][a][wv][lane & 31];                                                       \
      const uint32_t pm = a < n_active ? p.x : 0u, pn = a < n_active ? p.y : 0u;                            \
      n_ = max(max(min(m_, pm), n_), pn);                                                                   \
      m_ = max(m_, pm);                                                                                     \
    }                                                                                                       \
    const int slot = max(4 * g_ + wv, 0);                                                                   \
    cblk[slot * 32 + (lane & 31)] = make_uint2(m_, n_);                                                     \
  }
  S1_STAGE_TILE(0, 0)

  if (r0 >= nx) {  // a wave without rows: its share of the staging and of the column merges, the same barriers
    __syncthreads();
    for (int tile = 0; tile < n_full; ++tile) {
      S1_MERGE_GROUP(tile - 1)
      if (tile & 1) { S1_STAGE_TILE(tile + 1, 0) } else { S1_STAGE_TILE(tile + 1, 1) }
      __syncthreads();
    }
    S1_MERGE_GROUP(n_full - 1)
    __syncthreads();
    S1_MERGE_GROUP(n_full)   // (the slot behind a ragged tile is wave 0's, and wave 0 holds rows whenever the block runs)
    return;
  }

  bf16x8 A0[8], A1[8];
  {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    int row = r0 + (lane & 31);
    const u32x4* src = reinterpret_cast<const u32x4*>(xpool + (size_t)min(row, nx - 1) * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A0[ks] = __builtin_bit_cast(bf16x8, row < nx ? src[ks * 2] : zero);
    row = r0 + 32 + (lane & 31);
    src = reinterpret_cast<const u32x4*>(xpool + (size_t)min(row, nx - 1) * kSiftDim + (lane >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) A1[ks] = __builtin_bit_cast(bf16x8, row < nx ? src[ks * 2] : zero);
  }
  const uint32_t k0 = (lane < 32) ? 0xFFFFFFFFu : 0u;
  const bf16x8 A9 = __builtin_bit_cast(bf16x8, u32x4{0x3F80u & k0, 0u, 0u, 0u});
  auto seq_term = [&](int seq) {
    const float term = (float)(31 - seq) * 0.03125f;
    return __builtin_bit_cast(bf16x8, u32x4{(__float_as_uint(term) >> 16) & k0, 0u, 0u, 0u});
  };

  uint32_t mx0[16], sx0[16], mx1[16], sx1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) mx0[r] = sx0[r] = mx1[r] = sx1[r] = 0u;
  uint32_t cm = 0u, cn = 0u;   // this lane's column of the column tile in flight: best / second over the wave's rows so far
  __syncthreads();

  const uint32_t abyte = (uint32_t)((lane & 31) * 256 + ((((lane >> 5) ^ lane) & 15) << 4));
  const char* lds_bytes = reinterpret_cast<const char*>(&tileY[0][0]);
  uint2* const my_colp = &colp[0][wv][0][lane & 31];   // + parity * 512 + column tile * 32 (in uint2)

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
  bf16x8 Bf[8];

#define S1_READ_ONE(BOFF, CT, KS)                                                                   \
  {                                                                                                 \
    uint32_t a;                                                                                     \
    asm volatile("v_xor_b32 %0, %2, %1" : "=v"(a) : "v"(abyte), "n"((KS) << 5));                     \
    Bf[KS] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds_bytes + (a + (BOFF)) + (CT) * 8192)); \
  }
#define S1_READ_B(BOFF, CT)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) S1_READ_ONE(BOFF, CT, ks)
#define S1_MFMA_ONE(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC, 0, 0, 0);
  // row side (OK: the lane's "column exists" flag, or true) and column side of one accumulator
#if RGBDFE_SIFT1_TREE
#define S1_DIGEST(ACC, MX, SX, OK)                                                                  \
  {                                                                                                 \
    uint32_t k_[16];                                                                                \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
      k_[r] = __float_as_uint(ACC[r]);                                                              \
      top2_insert(MX[r], SX[r], (OK) ? k_[r] : 0u);                                                 \
    }                                                                                               \
    uint32_t m0, n0, m1, n1, m2, n2, m3, n3, m4, n4;                                                \
    top2_of3(k_[0], k_[1], k_[2], m0, n0);                                                          \
    top2_of3(k_[3], k_[4], k_[5], m1, n1);                                                          \
    top2_of3(k_[6], k_[7], k_[8], m2, n2);                                                          \
    top2_of3(k_[9], k_[10], k_[11], m3, n3);                                                        \
    top2_of3(k_[12], k_[13], k_[14], m4, n4);                                                       \
    top2_insert(m4, n4, k_[15]);                                                                    \
    top2_merge(m0, n0, m1, n1);                                                                     \
    top2_merge(m2, n2, m3, n3);                                                                     \
    top2_merge(m0, n0, m2, n2);                                                                     \
    top2_merge(m0, n0, m4, n4);                                                                     \
    top2_merge(cm, cn, m0, n0);                                                                     \
  }
#elif RGBDFE_SIFT1_DIAG
#define S1_DIGEST(ACC, MX, SX, OK)                                                                  \
  if (RGBDFE_SIFT1_DIAG == 1) { asm volatile("" : : "v"(ACC)); }                                     \
  else _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
    const uint32_t key = __float_as_uint(ACC[r]);                                                   \
    if (RGBDFE_SIFT1_DIAG == 2) top2_insert(MX[r], SX[r], (OK) ? key : 0u);                         \
    if (RGBDFE_SIFT1_DIAG == 4) top2_insert(cm, cn, key);                                           \
  }
#else
#define S1_DIGEST(ACC, MX, SX, OK)                                                                  \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
    const uint32_t key = __float_as_uint(ACC[r]);                                                   \
    top2_insert(MX[r], SX[r], (OK) ? key : 0u);                                                     \
    top2_insert(cm, cn, key);                                                                       \
  }
#endif
  // the column tile whose second accumulator has just been digested is complete: both lane halves hold the same 32
  // columns (rows +0 / +4) -> one pair per column into the wave's slot PSLOT of the tile parity PBUF; start the next one
#define S1_COLUMN_DONE(PBUF_OFF, PSLOT)                                                             \
  {                                                                                                 \
    const uint32_t pm = other_half(cm), pn = other_half(cn);                                        \
    cn = max(max(min(cm, pm), cn), pn);                                                             \
    cm = max(cm, pm);                                                                               \
    my_colp[(PBUF_OFF) + (PSLOT) * 32] = make_uint2(cm, cn);                                        \
    cm = 0u;                                                                                        \
    cn = 0u;                                                                                        \
  }
#define S1_FENCE() __builtin_amdgcn_sched_barrier(0);
#if RGBDFE_SIFT1_BURST
  // the chain's MFMAs back to back (an instruction between two MFMAs on the SAME accumulator takes the dependent one off
  // the matrix pipe's accumulate-forwarding path: +43 cycles each, MI355X_MICROARCH.md), then the B fragment reads, then the
  // digest: the wave sits in the matrix pipe's queue for the length of the chain while the SIMD's other wave digests
#define S1_TAIL_SCHED(WITH_READS, NV)                                   \
  __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);                    \
  _Pragma("unroll") for (int g = 0; g < 6; ++g) {                       \
    if (WITH_READS) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  \
    if (WITH_READS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
  }                                                                     \
  __builtin_amdgcn_sched_group_barrier(0x002, 7 * (NV), 0);             \
  S1_FENCE()
#else
#define S1_TAIL_SCHED(WITH_READS, NV)                                   \
  _Pragma("unroll") for (int g = 0; g < 7; ++g) {                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
    if ((WITH_READS) && g < 6) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); \
    if ((WITH_READS) && g < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0);               \
  }                                                                     \
  S1_FENCE()
#endif
#define S1_PIN(ACC) asm volatile("" : "+v"(ACC));
  // column tile CT of the Y tile in buffer offset BOFF; PBUF_OFF: the tile parity's offset into colp (in uint2)
#define S1_COLUMN_TILE(BOFF, PBUF_OFF, CT, SEQ, WITH_READS, OK_PREV, OK_THIS, NV)                   \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = 0.0f;                                     \
  S1_MFMA_ONE(acc0, A0[0], Bf[0])                                                                   \
  S1_MFMA_ONE(acc0, A0[1], Bf[1])                                                                   \
  S1_PIN(acc0)                                                                                      \
  S1_FENCE()                                                                                        \
  S1_PIN(acc1)                                                                                      \
  _Pragma("unroll") for (int ks = 2; ks < 8; ++ks) S1_MFMA_ONE(acc0, A0[ks], Bf[ks])                 \
  S1_MFMA_ONE(acc0, A9, seq_term(SEQ))                                                              \
  S1_PIN(acc0)                                                                                      \
  S1_DIGEST(acc1, mx1, sx1, OK_PREV)                                                                \
  S1_TAIL_SCHED(false, NV)                                                                          \
  S1_COLUMN_DONE(PBUF_OFF, CT)                                                                      \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                 \
    S1_MFMA_ONE(acc1, A1[ks], Bf[ks])                                                               \
    if (WITH_READS) S1_READ_ONE(BOFF, (CT) + 1, ks)                                                 \
  }                                                                                                 \
  S1_PIN(acc1)                                                                                      \
  S1_FENCE()                                                                                        \
  S1_PIN(acc0)                                                                                      \
  _Pragma("unroll") for (int ks = 2; ks < 8; ++ks) {                                                 \
    S1_MFMA_ONE(acc1, A1[ks], Bf[ks])                                                               \
    if (WITH_READS) S1_READ_ONE(BOFF, (CT) + 1, ks)                                                 \
  }                                                                                                 \
  S1_MFMA_ONE(acc1, A9, seq_term(SEQ))                                                              \
  S1_PIN(acc1)                                                                                      \
  S1_DIGEST(acc0, mx0, sx0, OK_THIS)                                                                \
  S1_TAIL_SCHED(WITH_READS, NV)
  // one full Y tile out of buffer BUF (a constant: the loop is unrolled by two); the column partials the previous tile's
  // waves left behind its barrier go out first
#define S1_FULL_TILE(BUF)                                                                           \
  {                                                                                                 \
    S1_MERGE_GROUP(tile - 1)                                                                        \
    S1_STAGE_TILE(tile + 1, (BUF) ^ 1)                                                              \
    const int seq0 = tile * 4;                                                                      \
    S1_READ_B((BUF) * 32768, 0)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    S1_COLUMN_TILE((BUF) * 32768, (BUF) * 512, 0, seq0, true, true, true, RGBDFE_SIFT1_NV)                       \
    S1_COLUMN_TILE((BUF) * 32768, (BUF) * 512, 1, seq0 + 1, true, true, true, RGBDFE_SIFT1_NV)                   \
    S1_COLUMN_TILE((BUF) * 32768, (BUF) * 512, 2, seq0 + 2, true, true, true, RGBDFE_SIFT1_NV)                   \
    S1_COLUMN_TILE((BUF) * 32768, (BUF) * 512, 3, seq0 + 3, false, true, true, RGBDFE_SIFT1_NV)                  \
    __syncthreads();                                                                                \
    ++tile;                                                                                         \
  }

  int tile = 0;
  while (tile + 1 < n_full) {
    S1_FULL_TILE(0)
    S1_FULL_TILE(1)
  }
  if (tile < n_full) S1_FULL_TILE(0)   // n_full odd: `tile` is even here
  // Behind the last full tile's barrier: its group goes out; acc1 still holds the last column tile of that tile (or the
  // pipeline's empty start) = slot 0 of group `tile`.
  S1_MERGE_GROUP(tile - 1)
  {
    const int pb = (tile & 1) * 512;
    if (tile < n_tiles) {  // ragged last tile: the same pipeline, columns beyond ny carry key 0 on the row side
      const int c0 = tile * kTile + (lane & 31);
      const bool ok0 = c0 < ny, ok1 = c0 + 32 < ny, ok2 = c0 + 64 < ny, ok3 = c0 + 96 < ny;
      const uint32_t boff = (uint32_t)(tile & 1) * 32768u;
      const int seq0 = tile * 4;
      S1_READ_B(boff, 0)
      __builtin_amdgcn_sched_barrier(0);
      S1_COLUMN_TILE(boff, pb, 0, seq0, true, true, ok0, 13)       // (its first half digests the column tile in flight)
      S1_COLUMN_TILE(boff, pb, 1, seq0 + 1, true, ok0, ok1, 13)
      S1_COLUMN_TILE(boff, pb, 2, seq0 + 2, true, ok1, ok2, 13)
      S1_COLUMN_TILE(boff, pb, 3, seq0 + 3, false, ok2, ok3, 13)
      // the ragged tile's last column tile = slot 0 of the group after this one: a buffer of its own (the other parity may
      // still be read by a wave that has not merged the last full tile's group yet)
      S1_DIGEST(acc1, mx1, sx1, ok3)
      {
        const uint32_t pm = other_half(cm), pn = other_half(cn);
        colp_last[wv][lane & 31] = make_uint2(max(cm, pm), max(max(min(cm, pm), cn), pn));
      }
    } else {
      S1_DIGEST(acc1, mx1, sx1, true)
      S1_COLUMN_DONE(pb, 0)
    }
  }
  __syncthreads();
  S1_MERGE_GROUP(tile)
  if (tile < n_tiles && wv == 0) {   // slot 0 of the group after the ragged tile's
    uint32_t m_ = 0u, n_ = 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint2 p = colp_last[a][lane & 31];
      const uint32_t pm = a < n_active ? p.x : 0u, pn = a < n_active ? p.y : 0u;
      n_ = max(max(min(m_, pm), n_), pn);
      m_ = max(m_, pm);
    }
    cblk[(4 * tile + 4) * 32 + (lane & 31)] = make_uint2(m_, n_);
  }
#undef S1_FULL_TILE
#undef S1_MFMA_ONE
#undef S1_READ_ONE
#undef S1_STAGE_TILE
#undef S1_GLDS
#undef S1_READ_B
#undef S1_DIGEST
#undef S1_COLUMN_DONE
#undef S1_TAIL_SCHED
#undef S1_PIN
#undef S1_FENCE
#undef S1_COLUMN_TILE
#undef S1_MERGE_GROUP
  uint32_t* __restrict__ opart = part + (size_t)pair * max_kp * 3;
  sift_merge_store<false>(mx0, sx0, r0, nx, lane, opart);
  if (r0 + 32 < nx) sift_merge_store<false>(mx1, sx1, r0 + 32, nx, lane, opart);
}

__device__ __forceinline__ float sift_angle(uint32_t dot) {
  // ProgramCU.cu:1738: acos(min(dot * 0.000003814697265625f, 1.0)): float product, double min/acos
  const float prod = (float)(int)dot * 0.000003814697265625f;
#if defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 10
  return 1.0f - prod;
#endif
  const double v = (double)prod < 1.0 ? (double)prod : 1.0;
  return (float)acos(v);
}

// One block per pair: row/column acceptance tests (RowMatch_Kernel :1738-1742, ColMatch_Kernel
// :1778-1781), mutual-best list in ascending query order (GetBestMatch), L2 distances of the raw
// descriptors (sift_gpu_wrapper.cpp:211-217).
__global__ __launch_bounds__(kSiftThreads) void sift_finish_kernel(
    const float* __restrict__ f32_pool, const PairWork* __restrict__ work, uint32_t max_kp,
    const uint32_t* __restrict__ row_part, uint32_t* __restrict__ col_part, const uint2* __restrict__ col_blocks,
    uint16_t* __restrict__ sm_q, uint16_t* __restrict__ sm_t, float* __restrict__ sm_d,
    int32_t* __restrict__ sm_n, uint32_t n_pairs) {
  __shared__ uint32_t wave_cnt[4];
  __shared__ int s_total;
  // Workgroups go round-robin over the 8 XCDs: XCD x takes the x-th contiguous eighth of the pair list, so that pairs
  // that share a node (a frame's candidate pairs are neighbours in the list) find its descriptors in ONE L2
  const uint32_t per_xcd = (n_pairs + 7u) / 8u;
  const uint32_t pair = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (pair >= n_pairs) return;
  const PairWork w = work[pair];
  const int nq = (int)min(w.nq, 4096u), nt = (int)min(w.nt, 4096u);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float distmax = 0.9f, ratiomax = 0.9f;  // sift_gpu_wrapper.cpp:185
  uint16_t* __restrict__ oq = sm_q + (size_t)pair * max_kp;
  uint16_t* __restrict__ ot = sm_t + (size_t)pair * max_kp;
  float* __restrict__ od = sm_d + (size_t)pair * max_kp;
  if (nq <= 0 || nt <= 0) {  // SiftMatchCU.cpp:141
    if (tid == 0) sm_n[pair] = 0;
    return;
  }
  uint32_t* __restrict__ cbase = col_part + (size_t)pair * max_kp * 3;
  // one-pass pairs (sift_top2_onepass_kernel): a column's two largest dot products per 256-row block instead of
  // (best, second, argmax row); the mutual-best test below compares dot products instead of row indices
  const bool onepass = col_blocks != nullptr && (w.pad & 1u);
  if (onepass) {
    const uint2* __restrict__ cb = col_blocks + (size_t)pair * 4 * (kColSlots * 32);
    const int n_blk = (nq + kTile64 - 1) / kTile64;
    for (int j = tid; j < nt; j += kSiftThreads) {
      uint32_t m = 0u, n = 0u;
      for (int b = 0; b < n_blk; ++b) {
        const uint2 p = cb[(size_t)b * (kColSlots * 32) + 32 + j];   // slot = column tile + 1
        n = max(max(min(m, p.x), n), p.y);
        m = max(m, p.x);
      }
      const uint32_t dot = (uint32_t)__uint_as_float(m), dotn = (uint32_t)__uint_as_float(n);   // dot + (31 - seq) / 32
      const float dist = sift_angle(dot), distn = sift_angle(dotn);
      // accepted (:1781) => best > second strictly => the argmax row is the one row whose dot equals `dot`
      cbase[(size_t)j * 3] = (dist < distmax) && (dist < distn * ratiomax) ? dot : 0xFFFFFFFFu;
    }
  } else {
    // ---- ColMatch_Kernel acceptance: col_match[j] replaces the best-dot slot
    for (int j = tid; j < nt; j += kSiftThreads) {
      const uint32_t dot = cbase[(size_t)j * 3], dotn = cbase[(size_t)j * 3 + 1];
      const int row = (int)cbase[(size_t)j * 3 + 2];
      const float dist = sift_angle(dot), distn = sift_angle(dotn);
      const int cm = (dist < distmax) && (dist < distn * ratiomax) ? row : -1;  // :1781
      cbase[(size_t)j * 3] = (uint32_t)cm;
    }
  }
  __syncthreads();
  // ---- RowMatch acceptance (:1738-1742) + mutual best in ascending query order
  const uint32_t* __restrict__ rpart = row_part + (size_t)pair * max_kp * 3;
  uint32_t base = 0;
  for (int i0 = 0; i0 < nq; i0 += kSiftThreads) {
    const int i = i0 + tid;
    bool keep = false;
    int j = -1;
    if (i < nq) {
      const uint32_t dot = rpart[(size_t)i * 3], dotn = rpart[(size_t)i * 3 + 1];
      const int idx = (int)rpart[(size_t)i * 3 + 2];
      const float dist = sift_angle(dot), distn = sift_angle(dotn);
      j = (dist < distmax) && (dist < distn * ratiomax) ? idx : -1;
      keep = j >= 0 && cbase[(size_t)j * 3] == (onepass ? dot : (uint32_t)i);  // SiftMatchCU.cpp:165
    }
    const uint64_t m = __ballot(keep);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = base;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    if (keep) {
      oq[off + rank] = (uint16_t)i;
      ot[off + rank] = (uint16_t)j;
    }
    base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  int number = (int)base;
  // ---- the wrapper's "context error" heuristic (sift_gpu_wrapper.cpp:199-209): more than half of
  // the matches touching index 0 clears everything.  counter <= 2, so only number <= 3 can trigger.
  if (tid == 0) {
    int n_out = number;
    if (number <= 3) {
      int counter = 0;
      for (int m = 0; m < number; ++m) {
        if (oq[m] == 0 || ot[m] == 0) counter++;
        if ((double)counter > 0.5 * (double)number) { n_out = 0; break; }
      }
    }
    s_total = n_out;
    sm_n[pair] = n_out;
  }
  __syncthreads();
  number = s_total;
  // ---- DMatch.distance: float L2 of the raw descriptors, sequential sum (:211-217)
  const float* __restrict__ qf = f32_pool + (size_t)w.q_slot * max_kp * kSiftDim;
  const float* __restrict__ tf = f32_pool + (size_t)w.t_slot * max_kp * kSiftDim;
#if defined(RGBDFE_SIFT_ABL) && RGBDFE_SIFT_ABL == 11
  for (int m = tid; m < number; m += kSiftThreads) od[m] = (float)(oq[m] + ot[m]);
  return;
#endif
  // Lane = match, but a lane reading its own two 512-byte rows makes every load instruction touch 64 cache lines (the
  // texture addresser, not the arithmetic, set the time: 0.30 of the kernel's 0.36 ms per 4000 pairs).  Each wave stages
  // 16 floats of the 2 x 64 rows of its 64 matches at a time through LDS instead: 4 lanes fetch one 64-byte row piece
  // (16 rows per instruction), rows padded to 20 floats (ds_read_b128 of 16 consecutive lanes hits 16 different bank
  // groups), and every lane then adds its own row pair in the reference's order.  No block barrier: a wave's LDS
  // operations execute in order and the four waves own disjoint staging areas.
  constexpr int kPiece = 16, kStride = 20;
  __shared__ float4 s_stage[4][2][64 * kStride / 4];
  float4* sq4 = s_stage[wv][0];
  float4* st4 = s_stage[wv][1];
  for (int base = wv * 64; base < number; base += kSiftThreads) {
    const int m = base + lane;
    const bool act = m < number;
    const uint32_t qi = act ? (uint32_t)oq[m] : 0u, ti = act ? (uint32_t)ot[m] : 0u;
    float sum = 0.0f;
    for (int c = 0; c < kSiftDim / kPiece; ++c) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + (lane >> 2);  // the match (lane of this wave) whose rows this lane helps to fetch
        const uint32_t rq = (uint32_t)__shfl((int)qi, r), rt = (uint32_t)__shfl((int)ti, r);
        const float4 vq = *reinterpret_cast<const float4*>(qf + (size_t)rq * kSiftDim + c * kPiece + (lane & 3) * 4);
        const float4 vt = *reinterpret_cast<const float4*>(tf + (size_t)rt * kSiftDim + c * kPiece + (lane & 3) * 4);
        sq4[r * (kStride / 4) + (lane & 3)] = vq;
        st4[r * (kStride / 4) + (lane & 3)] = vt;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k4 = 0; k4 < kPiece / 4; ++k4) {
        const float4 x = sq4[lane * (kStride / 4) + k4], y = st4[lane * (kStride / 4) + k4];
        float d;
        d = x.x - y.x; sum += d * d;
        d = x.y - y.y; sum += d * d;
        d = x.z - y.z; sum += d * d;
        d = x.w - y.w; sum += d * d;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (act) od[m] = sqrtf(sum);
  }
}

// RGBDFE_SIFT_ONEPASS=0: the two-pass form of the float-key path (A/B runs)
static bool sift_onepass_enabled() {
  static const bool on = !(getenv("RGBDFE_SIFT_ONEPASS") && atoi(getenv("RGBDFE_SIFT_ONEPASS")) == 0);
  return on;
}
size_t sift_col_block_bytes_per_pair() { return (size_t)4 * kColSlots * 32 * sizeof(uint2); }

void launch_sift_dot(const uint16_t* bf16_pool, const PairWork* work, uint32_t max_kp,
                     uint32_t n_pairs, uint32_t max_nq, uint32_t max_nt, uint32_t key_kinds,
                     uint32_t* row_part, uint32_t* col_part, uint2* col_blocks, hipStream_t stream) {
  if (n_pairs == 0) return;
  uint32_t rbq = (max_nq + kTile - 1) / kTile, rbt = (max_nt + kTile - 1) / kTile;
  if (rbq < 1) rbq = 1;
  if (rbt < 1) rbt = 1;
  // key_kinds: bit 0 = some pair of the batch carries float keys (PairWork::pad bit 0), bit 1 = some pair integer keys;
  // each kernel leaves the other kind's pairs at once
  const uint32_t groups = (n_pairs + 7u) / 8u * 8u;
  if ((key_kinds & 1u) && col_blocks && sift_onepass_enabled()) {
    // one sweep: row results + per-block column partials (fast pairs hold at most 1024 rows: at most 4 row blocks)
    uint32_t rb = (max_nq + kTile64 - 1) / kTile64;
    if (rb < 1) rb = 1;
    hipLaunchKernelGGL(sift_top2_onepass_kernel, dim3(rb * groups), dim3(kSiftThreads), 0, stream, bf16_pool, work, max_kp,
                       n_pairs, rb, row_part, col_blocks);
  } else if (key_kinds & 1u) {
    // 64 rows per wave (256-row blocks) unless the batch's nodes fit one 128-row block; RGBDFE_SIFT_ROWS64=0/1 forces
    // one kernel for A/B runs
    static const int rows64_env = [] { const char* e = getenv("RGBDFE_SIFT_ROWS64"); return e ? atoi(e) : -1; }();
    auto launch = [&](auto k32, auto k64, uint32_t max_n, uint32_t* out) {
      const bool rows64 = rows64_env >= 0 ? rows64_env != 0 : max_n > (uint32_t)kTile;
      if (rows64) {
        uint32_t rb = (max_n + kTile64 - 1) / kTile64;
        if (rb < 1) rb = 1;
        hipLaunchKernelGGL(k64, dim3(rb * groups), dim3(kSiftThreads), 0, stream, bf16_pool, work, max_kp, n_pairs, rb, out);
      } else {
        uint32_t rb = (max_n + kTile - 1) / kTile;
        if (rb < 1) rb = 1;
        hipLaunchKernelGGL(k32, dim3(rb * groups), dim3(kSiftThreads), 0, stream, bf16_pool, work, max_kp, n_pairs, rb, out);
      }
    };
    launch(sift_top2_fast_kernel<false>, sift_top2_fast64_kernel<false>, max_nq, row_part);
    launch(sift_top2_fast_kernel<true>, sift_top2_fast64_kernel<true>, max_nt, col_part);
  }
  if (key_kinds & 2u) {
    hipLaunchKernelGGL(sift_row_top2_kernel<false>, dim3(rbq * groups), dim3(kSiftThreads), 0, stream,
                       bf16_pool, work, max_kp, n_pairs, rbq, row_part);
    hipLaunchKernelGGL(sift_row_top2_kernel<true>, dim3(rbt * groups), dim3(kSiftThreads), 0, stream,
                       bf16_pool, work, max_kp, n_pairs, rbt, col_part);
  }
}

void launch_sift_finish(const float* f32_pool, const PairWork* work, uint32_t max_kp,
                        uint32_t n_pairs, const uint32_t* row_part, uint32_t* col_part, const uint2* col_blocks,
                        uint16_t* sm_q, uint16_t* sm_t, float* sm_d, int32_t* sm_n,
                        hipStream_t stream) {
  if (n_pairs == 0) return;
  hipLaunchKernelGGL(sift_finish_kernel, dim3((n_pairs + 7u) / 8u * 8u), dim3(kSiftThreads), 0, stream, f32_pool, work,
                     max_kp, row_part, col_part, sift_onepass_enabled() ? col_blocks : (const uint2*)nullptr, sm_q, sm_t, sm_d,
                     sm_n, n_pairs);
}

}  // namespace rgbdfe
