// Micro-benchmark: what a BARE stream of v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4, the instruction of hamming_mfma.hip) sustains
// on the box it runs on -- the ceiling the Hamming kernel's MFMA fraction is to be read against ("calibrate ...
// on the same box in the same call").  Every wave issues chains of dependent MFMAs on NACC independent accumulators, nothing
// else in the loop; 256 CUs x 4 SIMDs x W waves.  Prints TFLOP/s (2 x 32 x 32 x 64 flop per MFMA), the fraction of the
// 10 PFLOP/s dense fp4 figure, and the clock the launch ran at (s_memrealtime is 100 MHz, s_memtime counts core cycles).
//   hipcc --offload-arch=gfx950 -O3 mfma_fp4_rate.hip -o mfma_fp4_rate && ./mfma_fp4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x22222222 ^ (int)(threadIdx.x * 0x01010101u * (unsigned)i); b[i] = 0x11111111 * (i & 1); }
  v16f acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const uint64_t c0 = __builtin_readcyclecounter();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 4, 4, 0, 0, 0, 0);
  }
  const uint64_t c1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = c1 - c0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int waves_per_simd, int cus) {
  const int iters = 4000;
  const int threads = 64 * 4 * waves_per_simd;
  float* out; uint64_t* cyc;
  hipMalloc(&out, (size_t)cus * threads * 4); hipMalloc(&cyc, (size_t)cus * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(threads), 0, 0, out, cyc, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h((size_t)cus * 2); hipMemcpy(h.data(), cyc, (size_t)cus * 16, hipMemcpyDeviceToHost);
  double core = 0, real = 0; for (int i = 0; i < cus; ++i) { core += h[2 * i]; real += h[2 * i + 1]; }
  const double mfmas = (double)cus * 4 * waves_per_simd * iters * 8 * NACC;
  const double tflops = mfmas * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;
  printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"launch_ms\": %.3f, \"tflops\": %.1f, \"frac_of_10PF\": %.3f, "
         "\"core_clock_GHz\": %.3f, \"pipe_cycles_per_mfma_per_simd\": %.2f}\n",
         NACC, waves_per_simd, ms, tflops, tflops / 10000.0, core / real / 10.0,
         (core / cus) / ((double)waves_per_simd * iters * 8 * NACC));
  hipFree(out); hipFree(cyc);
}

int main() {
  int cus = 256;
  hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(1, cus); run<2>(1, cus); run<4>(1, cus);
    run<1>(2, cus); run<2>(2, cus); run<1>(3, cus); run<2>(3, cus); run<1>(4, cus); run<2>(4, cus);
  }
  return 0;
}
