// tests/emu/emu_orb.cpp -- TEST INFRASTRUCTURE ONLY: csrc/orb_kernels.hip compiled as host C++ over tests/emu/hip/hip_runtime.h
// (tests/test_emu_orb_pyramid.py builds it: the kernel SOURCE of the product runs, one OS thread per HIP thread).
#include "hip/hip_runtime.h"

thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
std::mutex g_mu;
std::condition_variable g_cv;
unsigned g_waiting = 0, g_generation = 0, g_block_threads = 1;
}  // namespace

void hipemu_barrier() {
  std::unique_lock<std::mutex> lk(g_mu);
  const unsigned gen = g_generation;
  if (++g_waiting == g_block_threads) {
    g_waiting = 0;
    ++g_generation;
    g_cv.notify_all();
  } else {
    g_cv.wait(lk, [&] { return g_generation != gen; });
  }
}

// workgroups one after the other; a thread that returns from the kernel early simply ends (the kernels run here only
// return early for a whole workgroup, or after their last barrier)
void hipemu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  blockDim = block; gridDim = grid;
  const unsigned n = block.x * block.y * block.z;
  for (unsigned b = 0; b < grid.x; ++b) {
    g_block_threads = n; g_waiting = 0;
    std::vector<std::thread> th;
    th.reserve(n);
    for (unsigned t = 0; t < n; ++t)
      th.emplace_back([&, t] {
        threadIdx = dim3(t % block.x, t / block.x, 0);
        blockIdx = dim3(b, 0, 0);
        body();
      });
    for (auto& x : th) x.join();
  }
}

namespace rgbdfe { alignas(16) uint8_t pyr_lds[64 * 1024]; }   // the kernel's `extern __shared__` array

#include "orb_kernels_emu.inc"   // csrc/orb_kernels.hip with its one `extern __shared__` declaration made a plain extern

// what rgbdfe_debug_pyramid_plan_check2 calls for the fused side: the product's launcher over the product's kernel
extern "C" void emu_orb_pyramid(uint8_t* pool, const rgbdfe::ResizeJob* jobs, const rgbdfe::PyrTile* tiles, int n_tiles,
                                const rgbdfe::PyrPlan* plan) {
  rgbdfe::launch_orb_pyramid(pool, jobs, tiles, n_tiles, *plan, nullptr);
}
