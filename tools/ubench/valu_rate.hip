// Micro-benchmark: issue cost (cycles per wave64 instruction on one SIMD) of the integer VALU
// instructions the Hamming kernel is made of.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int WHICH>
__global__ void k(uint32_t* out, uint64_t* cyc, uint32_t seed, int iters) {
  uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u,
           a6 = a0 * 17u, a7 = a0 * 19u;
  uint32_t s = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (WHICH == 0) {  // v_xor_b32 vgpr, vgpr
      REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                        "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 1) {  // v_xor_b32 sgpr, vgpr
      REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                        "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    } else if (WHICH == 2) {  // v_bcnt_u32_b32 accumulate
      REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                        "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 3) {  // v_add_u32
      REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                        "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 4) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                        "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0001f));)
    } else if (WHICH == 5) {  // v_min_u32
      REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                        "v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 6) {  // v_min3_u32
      REP8(asm volatile("v_min3_u32 %0, %0, %8, %1\n v_min3_u32 %1, %1, %8, %2\n v_min3_u32 %2, %2, %8, %3\n v_min3_u32 %3, %3, %8, %4\n"
                        "v_min3_u32 %4, %4, %8, %5\n v_min3_u32 %5, %5, %8, %6\n v_min3_u32 %6, %6, %8, %7\n v_min3_u32 %7, %7, %8, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 7) {  // v_lshl_or_b32
      REP8(asm volatile("v_lshl_or_b32 %0, %0, 16, %8\n v_lshl_or_b32 %1, %1, 16, %8\n v_lshl_or_b32 %2, %2, 16, %8\n v_lshl_or_b32 %3, %3, 16, %8\n"
                        "v_lshl_or_b32 %4, %4, 16, %8\n v_lshl_or_b32 %5, %5, 16, %8\n v_lshl_or_b32 %6, %6, 16, %8\n v_lshl_or_b32 %7, %7, 16, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));)
    } else if (WHICH == 8) {  // v_fma_f64
      double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
      REP8(asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                        "v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));)
      f0 += (float)(d0 + d1 + d2 + d3);
    } else if (WHICH == 9) {  // v_xor with sgpr then bcnt (the kernel's pair), 4 + 4
      REP8(asm volatile("v_xor_b32 %4, %8, %0\n v_xor_b32 %5, %8, %1\n v_xor_b32 %6, %8, %2\n v_xor_b32 %7, %8, %3\n"
                        "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^
      __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W>
void run(const char* name, int waves_per_simd) {
  const int iters = 2000, cus = 256;
  const int threads = 64 * 4 * waves_per_simd;  // one block per CU: waves_per_simd waves on each SIMD
  uint32_t* out; uint64_t* cyc;
  hipMalloc(&out, (size_t)cus * threads * 4); hipMalloc(&cyc, cus * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<W><<<cus, threads>>>(out, cyc, 12345u, 10);
  hipEventRecord(e0);
  k<W><<<cus, threads>>>(out, cyc, 12345u, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(cus); hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= cus;
  const double instr_per_wave = (double)iters * 64;
  printf("%-28s waves/SIMD %d: %.2f s_memtime ticks per instr per wave, %.2f ticks per instr per SIMD; wall %.3f ms => %.2f ns per instr per SIMD\n",
         name, waves_per_simd, mean / instr_per_wave, mean / instr_per_wave / waves_per_simd, ms,
         ms * 1e6 / (instr_per_wave * waves_per_simd));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_xor_b32 v,v", w); run<1>("v_xor_b32 s,v", w); run<2>("v_bcnt_u32_b32 (acc)", w);
    run<3>("v_add_u32", w); run<4>("v_fma_f32", w); run<5>("v_min_u32", w); run<6>("v_min3_u32", w);
    run<7>("v_lshl_or_b32", w); run<8>("v_fma_f64", w); run<9>("xor(s)+bcnt pairs", w);
  }
  return 0;
}
