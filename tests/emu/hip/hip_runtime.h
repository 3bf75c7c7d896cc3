// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY (found before the real header through -I tests/emu).
// A HIP-on-CPU vocabulary just large enough to compile csrc/orb_kernels.hip and csrc/sift_pyramid_kernels.h with g++ and RUN those of their kernels that use
// no wave-level operation (orb_pyramid_kernel, orb_resize_kernel, orb_blur_kernel; the SIFT pyramid and extremum kernels) on the host: one OS thread per HIP thread
// of a workgroup, workgroups one after the other, __shared__ = static storage, __syncthreads() = a barrier over the
// workgroup's threads.  Ballot and shuffles are served for one-wave workgroups (through that barrier); DPP, mbcnt, readlane and
// wave intrinsics in larger workgroups compile to calls that abort:
// the kernels built on them are not run here.  Nothing under rgbdslam_v2_amd/ includes this file.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline ushort4 make_ushort4(unsigned short a, unsigned short b, unsigned short c, unsigned short d) { return ushort4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef int hipError_t;
enum { hipSuccess = 0 };

extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __constant__ static
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyToSymbol(void* dst, const void* src, size_t n) { memcpy(dst, src, n); return hipSuccess; }
enum { hipMemcpyHostToDevice = 1 };
static inline hipError_t hipMemcpyToSymbolAsync(void* dst, const void* src, size_t n, size_t off, int, hipStream_t) {
  memcpy((char*)dst + off, src, n);
  return hipSuccess;
}

// the overloads device code gets from the HIP headers
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return a < b ? a : b; }
static inline float max(float a, float b) { return a > b ? a : b; }
static inline int __float2int_rn(float v) { return (int)rintf(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

void hipemu_barrier();
#define __syncthreads() hipemu_barrier()
void hipemu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipemu_launch(grid, block, [&] { kernel(__VA_ARGS__); })

// wave-level operations: atomics, readfirstlane of a uniform value and -- for one-wave workgroups -- ballot / shuffles are
// served; kernels that use anything else are compiled but must not run on this emulation
[[noreturn]] static inline void hipemu_no_wave_ops(const char* what) {
  fprintf(stderr, "hip emulation: %s is a wave-level operation (this kernel cannot run here)\n", what);
  abort();
}
// __ballot / __shfl / __shfl_up: served for workgroups that are ONE wave (64 threads) whose lanes all take part -- the
// workgroup barrier is then the wave's rendezvous: every lane deposits its value, all wait, every lane reads its source.
// (sift_row_scan_kernel, sift_key_emit_kernel.)  Any other workgroup shape aborts as before.
extern int hipemu_wave_slot[64];
static inline void hipemu_need_one_wave(const char* what) {
  if (blockDim.x * blockDim.y * blockDim.z != 64) hipemu_no_wave_ops(what);
}
static inline int hipemu_wave_fetch(int v, int src_lane, const char* what) {
  hipemu_need_one_wave(what);
  const int lane = (int)threadIdx.x;
  hipemu_wave_slot[lane] = v;
  hipemu_barrier();
  const int r = hipemu_wave_slot[(src_lane >= 0 && src_lane < 64) ? src_lane : lane];
  hipemu_barrier();
  return r;
}
static inline unsigned long long __ballot(int pred) {
  unsigned long long m = 0;
  hipemu_need_one_wave("__ballot");
  const int lane = (int)threadIdx.x;
  hipemu_wave_slot[lane] = pred ? 1 : 0;
  hipemu_barrier();
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)(hipemu_wave_slot[i] & 1) << i;
  hipemu_barrier();
  return m;
}
static inline int __shfl(int v, int src_lane) { return hipemu_wave_fetch(v, src_lane, "__shfl"); }
static inline int __shfl_up(int v, int delta) { return hipemu_wave_fetch(v, (int)threadIdx.x - delta, "__shfl_up"); }
static inline int __shfl_xor(int, int) { hipemu_no_wave_ops("__shfl_xor"); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }   // LDS or global: one address space here
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only ever applied to wave-uniform values
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned, unsigned) { hipemu_no_wave_ops("mbcnt"); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned, unsigned) { hipemu_no_wave_ops("mbcnt"); }
static inline int __builtin_amdgcn_update_dpp(int, int, int, int, int, bool) { hipemu_no_wave_ops("update_dpp"); }
static inline int __builtin_amdgcn_readlane(int, int) { hipemu_no_wave_ops("readlane"); }
static inline void __builtin_amdgcn_wave_barrier() { hipemu_no_wave_ops("wave_barrier"); }
