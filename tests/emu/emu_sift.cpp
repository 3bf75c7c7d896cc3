// tests/emu/emu_sift.cpp -- TEST INFRASTRUCTURE ONLY: csrc/sift_pyramid_kernels.h (the product's SIFT pyramid and extremum
// kernels, the candidate-list kernels, their launch chains and the extractor's geometry) compiled as host C++ over tests/emu/hip/hip_runtime.h.
// tests/test_emu_sift_kernels.py builds it and holds the kernel SOURCES against SiftGPU's own kernels
// (oracle/_ref/libref_siftgpu.so): one OS thread per HIP thread, workgroups one after the other.
#include "hip/hip_runtime.h"

#include "hipemu_runtime.inc"

#include "sift_pyramid_kernels.h"

namespace rgbdfe {
// the extractor's device-buffer management lives in sift_extract.hip, which is not part of this library: the emulation
// owns the (host) buffers it binds below
SiftExtractor::~SiftExtractor() {}
void SiftExtractor::release() {}
}  // namespace rgbdfe

using rgbdfe::SiftExtractor;

namespace {
struct Run {
  SiftExtractor E;
  std::vector<uint8_t> gray;
  std::vector<float> input, up, planes;
  std::vector<int8_t> flags;
  std::vector<int> rowcnt;   // rowcnt [rows] | rowoff [rows] | row2lvl [rows] | lvltot [64]: the extractor's layout for one frame
  std::vector<int> rowcnt_after_flags;   // (the emit kernel resets the counts it has used)
  std::vector<float> cand;
  std::vector<int> level_offset;   // first candidate of every (octave, dog level), and the total
};
Run* g_run = nullptr;
}  // namespace

// The shape-static half of SiftExtractor::begin_batch for ONE frame: plan_geometry + bind_levels (the product's), then
// launch_pyramid and launch_key_flags (the product's launch chains over the product's kernels).  filter_choice: the tile shape
// of every Gaussian level's launch (filter_tile_choice's codes: 0 = 16 x 16, 1 = 64 x 16, 2 = 64 x 32, 3 = 64 x 64), -1 = the
// product's choice by plane size.  Returns the number of octaves (< 0: the geometry was refused).
extern "C" int emu_sift_run(const uint8_t* gray, int cols, int rows, int filter_choice) {
  delete g_run;
  g_run = new Run();
  Run& R = *g_run;
  SiftExtractor& E = R.E;
  E.init_params();
  std::string err;
  const int rc = E.plan_geometry(rows, cols, err);
  if (rc != 0) return rc;
  R.gray.assign(gray, gray + (size_t)rows * cols);
  R.input.assign(E.input_floats, 0.f);
  R.up.assign(E.oct[0].plane, 0.f);
  R.planes.assign(E.planes_floats, 0.f);
  R.flags.assign(E.flags_bytes, 0);
  R.rowcnt.assign((size_t)E.total_rows * 3 + 64, 0);
  E.frames_cap = 1;
  E.cand_cap = (size_t)1 << 16;
  R.cand.assign(E.cand_cap * 6, 0.f);
  E.d_gray = R.gray.data(); E.d_input = R.input.data(); E.d_up = R.up.data(); E.d_planes = R.planes.data();
  E.d_flags = R.flags.data(); E.d_rowcnt = R.rowcnt.data();
  E.d_rowoff = E.d_rowcnt + E.total_rows;
  E.d_lvltot = E.d_rowcnt + (size_t)E.total_rows * 3;
  E.d_cand = R.cand.data();
  E.bind_levels();
  E.d_levels = E.h_levels.data();
  E.d_key_tiles = E.h_key_tiles.data();
  E.d_octs = E.h_octs.data();
  E.n_key_tiles = (int)E.h_key_tiles.size();
  rgbdfe::launch_pyramid(E, 1, nullptr, filter_choice);
  rgbdfe::FrameStrides st{};
  st.planes = E.planes_floats; st.flags = E.flags_bytes; st.rows = E.total_rows; st.lvltot = 64;
  st.cand = E.cand_cap * 6;
  rgbdfe::launch_key_flags(E, 1, st, nullptr);
  memcpy(E.d_rowcnt + (size_t)E.total_rows * 2, E.h_row2lvl.data(), sizeof(int) * (size_t)E.total_rows);
  R.rowcnt_after_flags.assign(R.rowcnt.begin(), R.rowcnt.begin() + E.total_rows);
  rgbdfe::launch_key_lists(E, 1, st, nullptr);   // the scan + the ordered emit: one-wave workgroups, ballot / shuffles served
  const int nlv = E.octave_num * SiftExtractor::kDogLevels;
  R.level_offset.assign((size_t)nlv + 1, 0);
  for (int i = 0; i < nlv; ++i) R.level_offset[(size_t)i + 1] = R.level_offset[(size_t)i] + E.d_lvltot[i];
  return E.octave_num;
}

extern "C" int emu_sift_octave_size(int octave, int* w, int* h) {
  if (!g_run || octave < 0 || octave >= g_run->E.octave_num) return -1;
  *w = g_run->E.oct[octave].w; *h = g_run->E.oct[octave].h;
  return 0;
}
extern "C" const float* emu_sift_plane(int octave, int level) { return g_run->E.oct[octave].g[level]; }
extern "C" const int8_t* emu_sift_flags(int octave, int dog_level) {
  return g_run->E.h_levels[(size_t)octave * SiftExtractor::kDogLevels + dog_level].flags;
}
extern "C" const int* emu_sift_rowcnt(int octave, int dog_level) {
  return g_run->rowcnt_after_flags.data() + g_run->E.h_levels[(size_t)octave * SiftExtractor::kDogLevels + dog_level].row0;
}

// the candidate list of one (octave, dog level): rows of (x, y, sign, dx, dy, ds) in list order; returns the count
extern "C" int emu_sift_candidates(int octave, int dog_level, const float** rows) {
  const size_t i = (size_t)octave * SiftExtractor::kDogLevels + dog_level;
  *rows = g_run->cand.data() + (size_t)g_run->level_offset[i] * 6;
  return g_run->level_offset[i + 1] - g_run->level_offset[i];
}

// one Gaussian level of one plane through the product's launcher (any size, any tile shape)
extern "C" int emu_sift_filter(const float* src, int w, int h, float sigma, float* dst, int choice) {
  const rgbdfe::Taps t = rgbdfe::make_taps(sigma);
  rgbdfe::launch_filter_any(rgbdfe::FilterArgs{src, dst, w, h, 1, 0, 0}, t, nullptr, choice);
  return t.fw;
}
