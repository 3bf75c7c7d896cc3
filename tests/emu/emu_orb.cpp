// tests/emu/emu_orb.cpp -- TEST INFRASTRUCTURE ONLY: csrc/orb_kernels.hip compiled as host C++ over tests/emu/hip/hip_runtime.h
// (tests/test_emu_orb_kernels.py builds it: the kernel SOURCE of the product runs, one OS thread per HIP thread).
#include "hip/hip_runtime.h"

#include "hipemu_runtime.inc"

namespace rgbdfe { alignas(16) uint8_t pyr_lds[64 * 1024]; }   // the kernel's `extern __shared__` array

#include "orb_internal.h"
// the setup stream lives in orb_host.hip, which is not part of this library: run the operation in place
namespace rgbdfe { hipError_t orb_setup_stream_run(const std::function<hipError_t(hipStream_t)>& op) { return op(nullptr); } }

#include "orb_kernels_emu.inc"   // csrc/orb_kernels.hip with its one `extern __shared__` declaration made a plain extern

// what rgbdfe_debug_pyramid_plan_check2 calls for the fused side: the product's launcher over the product's kernel
extern "C" void emu_orb_pyramid(uint8_t* pool, const rgbdfe::ResizeJob* jobs, const rgbdfe::PyrTile* tiles, int n_tiles,
                                const rgbdfe::PyrPlan* plan) {
  rgbdfe::launch_orb_pyramid(pool, jobs, tiles, n_tiles, *plan, nullptr);
}

// One image through the product's per-level kernels (the `levels` path's resize, the 7x7 blur): the tables the launchers
// expect are built here the way OrbWorkspace::prepare builds them (64 x 16 pixel workgroup tiles; scale = 1 / (dw / sw)).
extern "C" void emu_orb_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int is_mask) {
  std::vector<uint8_t> pool((size_t)sw * sh + (size_t)dw * dh + 256, 0);
  memcpy(pool.data(), src, (size_t)sw * sh);
  rgbdfe::ResizeJob j{};
  j.src_off = 0; j.dst_off = (uint32_t)((size_t)sw * sh);
  j.sw = sw; j.sh = sh; j.sstride = sw; j.dw = dw; j.dh = dh; j.is_mask = is_mask;
  j.scale_x = 1. / ((double)dw / sw);
  j.scale_y = 1. / ((double)dh / sh);
  std::vector<rgbdfe::TileUnit> units;
  for (int by = 0; by < (dh + 15) / 16; ++by)
    for (int bx = 0; bx < (dw + 63) / 64; ++bx) units.push_back(rgbdfe::TileUnit{0, (uint16_t)bx, (uint16_t)by, 0});
  rgbdfe::launch_orb_resize(pool.data(), &j, units.data(), (int)units.size(), nullptr);
  memcpy(dst, pool.data() + j.dst_off, (size_t)dw * dh);
}

extern "C" void emu_orb_blur(const uint8_t* src, int w, int h, uint8_t* dst) {
  std::vector<uint8_t> pool((size_t)w * h + 256, 0), blur((size_t)w * h + 256, 0);
  memcpy(pool.data(), src, (size_t)w * h);
  rgbdfe::ImgDesc im{};
  im.off = 0; im.w = w; im.h = h; im.stride = w; im.score_off = 0;
  std::vector<rgbdfe::TileUnit> units;
  for (int by = 0; by < (h + 15) / 16; ++by)
    for (int bx = 0; bx < (w + 63) / 64; ++bx) units.push_back(rgbdfe::TileUnit{0, (uint16_t)bx, (uint16_t)by, 0});
  rgbdfe::launch_orb_blur_always(pool.data(), &im, units.data(), (int)units.size(), blur.data(), nullptr);
  memcpy(dst, blur.data(), (size_t)w * h);
}
