// oracle/ref_stubs/siftgpu_glue.cpp -- TEST INFRASTRUCTURE ONLY (see siftgpu_emu_prelude.h).
// Appended behind the reference's own code in the translation unit oracle/Makefile assembles: the fiber scheduler, the
// CuTexImage storage (plain host memory), the GL-only members of PyramidCU that the extraction path never calls, and the
// C entry points the tests use.

dim3 threadIdx, blockIdx, blockDim, gridDim;

// ---- fibers: callee-saved registers + stack pointer (x86-64 System V) -----------------------------------------------
extern "C" void cuemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl cuemu_switch
.type cuemu_switch,@function
cuemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
)");

namespace {
constexpr size_t kStack = 64 * 1024;
struct Fiber { void* sp; char* stack; bool done; dim3 tid; };
std::vector<Fiber> g_fibers;
void* g_sched_sp = nullptr;
int g_cur = -1;
CuemuBody* g_body = nullptr;

void fiber_entry() {
  g_body->run();
  g_fibers[(size_t)g_cur].done = true;
  void* dummy;
  cuemu_switch(&dummy, g_sched_sp);   // never resumed
  abort();
}
}  // namespace

void cuemu_barrier() {
  Fiber& f = g_fibers[(size_t)g_cur];
  cuemu_switch(&f.sp, g_sched_sp);
  threadIdx = f.tid;                  // resumed: restore this thread's index
}

void cuemu_run(dim3 grid, dim3 block, CuemuBody& body) {
  const size_t nt = (size_t)block.x * block.y * block.z;
  if (g_fibers.size() < nt) {
    const size_t old = g_fibers.size();
    g_fibers.resize(nt);
    for (size_t i = old; i < nt; ++i) g_fibers[i].stack = (char*)aligned_alloc(64, kStack);
  }
  blockDim = block; gridDim = grid;
  g_body = &body;
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      blockIdx = dim3(bx, by, 0);
      for (size_t t = 0; t < nt; ++t) {
        Fiber& f = g_fibers[t];
        f.done = false;
        f.tid = dim3((unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y)));
        void** top = (void**)(f.stack + kStack);   // 16-byte aligned
        top[-1] = nullptr;                         // padding: the entry sees rsp = 8 (mod 16), as after a call
        top[-2] = (void*)&fiber_entry;             // "return address" of the first switch
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;   // rbp rbx r12 r13 r14 r15
        f.sp = (void*)(top - 8);
      }
      size_t alive = nt;
      while (alive) {                              // round robin: every pass runs each live thread to its next barrier
        for (size_t t = 0; t < nt; ++t) {
          Fiber& f = g_fibers[t];
          if (f.done) continue;
          g_cur = (int)t;
          threadIdx = f.tid;
          cuemu_switch(&g_sched_sp, f.sp);
          if (f.done) --alive;
        }
      }
    }
  g_body = nullptr;
}

// ---- CuTexImage: linear host memory (CuTexImage.cpp:47-184 minus the CUDA / GL calls) -----------------------------------
CuTexImage::CuTexImage() {
  _cuData = NULL; _cuData2D = NULL; _fromPBO = 0;
  _numChannel = _numBytes = 0;
  _imgWidth = _imgHeight = _texWidth = _texHeight = 0;
}
CuTexImage::CuTexImage(int, int, int, GLuint) {   // PBO-backed images: display path only
  _cuData = NULL; _cuData2D = NULL; _fromPBO = 0;
  _numChannel = _numBytes = 0;
  _imgWidth = _imgHeight = _texWidth = _texHeight = 0;
}
CuTexImage::~CuTexImage() { if (_cuData) free(_cuData); }
void CuTexImage::SetImageSize(int width, int height) { _imgWidth = width; _imgHeight = height; }
void CuTexImage::InitTexture(int width, int height, int nchannel) {   // CuTexImage.cpp:127-146: grows, never shrinks
  _imgWidth = width; _imgHeight = height;
  _numChannel = min(max(nchannel, 1), 4);
  const int size = width * height * _numChannel * (int)sizeof(float);
  if (size <= _numBytes) return;
  if (_cuData) free(_cuData);
  // cudaMalloc leaves the memory undefined: zeros make the emulation repeatable.  + 8 KiB: ListGen_Kernel has no
  // `idx1 < len` guard (ProgramCU.cu:738-761) and writes up to 127 int4 past a feature list -- harmless inside cudaMalloc's
  // allocation granularity on the device, fatal on a host heap without slack
  _cuData = calloc((size_t)size + 8192, 1);
  _numBytes = size;
}
void CuTexImage::CopyFromHost(const void* buf) {
  if (_cuData) memcpy(_cuData, buf, (size_t)_imgWidth * _imgHeight * _numChannel * sizeof(float));
}
void CuTexImage::CopyToHost(void* buf) {
  if (_cuData) memcpy(buf, _cuData, (size_t)_imgWidth * _imgHeight * _numChannel * sizeof(float));
}
void CuTexImage::CopyToHost(void* buf, int) { CopyToHost(buf); }
void CuTexImage::InitTexture2D() {}     // SIFTGPU_ENABLE_LINEAR_TEX2D (CuTexImage.h:33): 2-D fetches read the linear memory
void CuTexImage::CopyToTexture2D() {}
int CuTexImage::CopyToPBO(GLuint) { return 0; }
void CuTexImage::CopyFromPBO(int, int, GLuint) {}
int CuTexImage::DebugCopyToTexture2D() { return 1; }
// ProgramCU.cu:1340-1354
inline void CuTexImage::BindTexture(textureReference& texRef) { texRef.ptr = _cuData; texRef.bytes = _numBytes; }
inline void CuTexImage::BindTexture2D(textureReference& texRef) {
  texRef.ptr = _cuData; texRef.bytes = _numBytes;
  texRef.width = _imgWidth; texRef.height = _imgHeight; texRef.pitch_elems = _imgWidth;
}

// ---- ProgramCU / GlobalUtil members outside the compiled ranges ------------------------------------------------------
void ProgramCU::FinishCUDA() {}
int ProgramCU::CheckErrorCUDA(const char*) { return 0; }
int ProgramCU::CheckCudaDevice(int) { return 1; }
void ProgramCU::DisplayConvertDOG(CuTexImage*, CuTexImage*) {}
void ProgramCU::DisplayConvertGRD(CuTexImage*, CuTexImage*) {}
void ProgramCU::DisplayConvertKEY(CuTexImage*, CuTexImage*, CuTexImage*) {}
void ProgramCU::DisplayKeyPoint(CuTexImage*, CuTexImage*) {}
void ProgramCU::DisplayKeyBox(CuTexImage*, CuTexImage*) {}
ClockTimer GlobalUtil::_globalTimer;
double ClockTimer::CLOCK() { return 0.0; }
void ClockTimer::StopTimer(int) {}
void ClockTimer::StartTimer(const char*, int) {}
float ClockTimer::GetElapsedTime() { return 0.f; }
void GlobalUtil::InitGLParam(int) {}
void GlobalUtil::SetGLParam() {}

// ---- PyramidCU members that only serve the OpenGL display path ------------------------------------------------------
PyramidCU::~PyramidCU() {
  DestroyPerLevelData();
  DestroySharedData();
  DestroyPyramidData();
  if (_inputTex) delete _inputTex;
}
void PyramidCU::GenerateFeatureDisplayVBO() {}
GLTexImage* PyramidCU::GetLevelTexture(int, int) { return nullptr; }
GLTexImage* PyramidCU::GetLevelTexture(int, int, int) { return nullptr; }
GLTexImage* PyramidCU::ConvertTexCU2GL(CuTexImage*, int) { return nullptr; }
void SiftPyramid::SaveSIFT(const char*) {}

// ---- the C entry points ---------------------------------------------------------------------------------------------
namespace {
struct RefSift {
  SiftParam param;
  PyramidCU* pyramid = nullptr;
  GLTexInput input;
  std::vector<float> pixels;
  int w = 0, h = 0;
};
RefSift* g_sift = nullptr;

// SiftGPUWrapper::SiftGPUWrapper (src/sift_gpu_wrapper.cpp:29-88) hands SiftGPU::ParseParam
//   -cuda -s 1 -tc2 <max_keypoints> -fo -1 -v 0 -unn -d 5 -e 10.0 -ofix-not
// whose effect on the statics is (SiftGPU.cpp:716-1180): "-s 1" _SubpixelLocalization = 1 (:889-897); "-tc2 N"
// _TruncateMethod = 1, _FeatureCountThreshold = N (:1088-1107); "-fo -1" _octave_min_default = -1 (:990-999); "-v 0"
// _verbose = _timingS = 0 (:396-407); "-unn" _NormalizedSIFT = 0 (:851-853); "-d 5" _dog_level_num = 5 (:1048-1057);
// "-e 10.0" _edge_threshold = 10 (:1038-1047); "-ofix-not" _FixedOrientation = 0 (:898-900); "-cuda" _UseCUDA = 1.
void configure(int max_features) {
  GlobalUtil::_UseCUDA = 1;
  GlobalUtil::_SubpixelLocalization = 1;
  GlobalUtil::_TruncateMethod = 1;
  GlobalUtil::_FeatureCountThreshold = max_features;
  GlobalUtil::_octave_min_default = -1;
  GlobalUtil::_verbose = 0;
  GlobalUtil::_timingS = 0;
  GlobalUtil::_NormalizedSIFT = 0;
  GlobalUtil::_FixedOrientation = 0;
  GlobalUtil::_GoodOpenGL = 1;
}
}  // namespace

extern "C" {

// SiftGPUWrapper::detect (src/sift_gpu_wrapper.cpp:113-167) for a mono8 image: cvMatToSiftGPU + SiftGPU::RunSIFT(w, h, data,
// GL_LUMINANCE, GL_UNSIGNED_BYTE) (SiftGPU.cpp:223-257: GLTexInput::SetImageData -> InitPyramid -> SiftPyramid::RunSIFT) +
// GetFeatureVector.  keys: n x 4 (x, y, scale, orientation) as SiftGPU returns them; desc: n x 128.  Returns n, or -n-1
// when `capacity` rows do not hold the n features found.
int ref_siftgpu_run(const unsigned char* gray, int width, int height, int max_features, float* keys, float* desc,
                    int capacity) {
  configure(max_features);
  // a fresh instance per call: SiftGPU keeps state between frames (a level skipped by the "-tc2" limit keeps the previous
  // frame's feature list and count, PyramidCU.cpp:797-815 -- DESIGN.md 4.11); the pin answers "what does a new
  // SiftGPUWrapper return for this image"
  if (g_sift) { delete g_sift->pyramid; delete[] g_sift->param._sigma; delete g_sift; g_sift = nullptr; }
  if (!g_sift) {
    g_sift = new RefSift();
    g_sift->param._dog_level_num = 5;
    g_sift->param._edge_threshold = 10.0f;
    g_sift->param.ParseSiftParam();          // SiftGPU::InitSiftGPU (SiftGPU.cpp:171)
    g_sift->pyramid = new PyramidCU(g_sift->param);
  }
  RefSift& s = *g_sift;
  // GLTexInput::SetImageData, CUDA branch (GLTexImage.cpp:971-1009) with DownSamplePixelDataI2F (:808-831): luminance
  // bytes / 255.0f, the width truncated to a multiple of 4 (TruncateWidthCU), no CPU down-sampling (octave_min <= 0)
  const int tw = GLTexInput::TruncateWidthCU(width);
  s.pixels.resize((size_t)tw * height);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < tw; ++x) s.pixels[(size_t)y * tw + x] = gray[(size_t)y * width + x] / 255.0f;
  s.input._pixel_data = s.pixels.data();
  s.input._imgWidth = tw;
  s.input._imgHeight = height;
  s.input._down_sampled = 0;
  s.pyramid->InitPyramid(width, height, 0);
  s.pyramid->RunSIFT(&s.input);
  const int n = s.pyramid->GetFeatureNum();
  if (n > capacity) return -n - 1;
  s.pyramid->CopyFeatureVector(keys, desc);
  return n;
}

// SiftGPUWrapper::detect with a keypoint list (src/sift_gpu_wrapper.cpp:132-142): SiftGPU::SetKeypointList(num, keys) --
// keys_have_orientation defaults to 1 (SiftGPU.h:150) -> SiftPyramid::SetKeypointList(num, keys, 0, 1) (SiftGPU.cpp:365-368) --
// then RunSIFT and GetFeatureVector(NULL, descriptors).  keys: n x 4 (x, y, scale, orientation).  desc: n x 128.
int ref_siftgpu_describe(const unsigned char* gray, int width, int height, const float* keys, int n, float* desc) {
  configure(1000);
  if (g_sift) { delete g_sift->pyramid; delete[] g_sift->param._sigma; delete g_sift; g_sift = nullptr; }
  g_sift = new RefSift();
  g_sift->param._dog_level_num = 5;
  g_sift->param._edge_threshold = 10.0f;
  g_sift->param.ParseSiftParam();
  g_sift->pyramid = new PyramidCU(g_sift->param);
  RefSift& s = *g_sift;
  const int tw = GLTexInput::TruncateWidthCU(width);
  s.pixels.resize((size_t)tw * height);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < tw; ++x) s.pixels[(size_t)y * tw + x] = gray[(size_t)y * width + x] / 255.0f;
  s.input._pixel_data = s.pixels.data();
  s.input._imgWidth = tw;
  s.input._imgHeight = height;
  s.input._down_sampled = 0;
  s.pyramid->SetKeypointList(n, keys, 0, 1);
  s.pyramid->InitPyramid(width, height, 0);
  s.pyramid->RunSIFT(&s.input);
  if (s.pyramid->GetFeatureNum() != n) return -1;
  s.pyramid->CopyFeatureVector(nullptr, desc);
  return n;
}

// pyramid geometry of the last run + one level's planes, for stage-by-stage checks.  data: 0 Gaussian (1 float), 1 DoG
// (1), 2 keypoint map (4: extremum sign, dx, dy, ds), 3 gradient (2: magnitude, angle).  level: 0 .. level_num-1
// (= SiftParam level _level_min + level).  Returns the number of floats written (w x h x channels) or 0.
int ref_siftgpu_geometry(int* octave_min, int* octave_num, int* level_num, int* dog_level_num) {
  if (!g_sift) return 0;
  *octave_min = g_sift->pyramid->_octave_min;
  *octave_num = g_sift->pyramid->_octave_num;
  *level_num = g_sift->param._level_num;
  *dog_level_num = g_sift->param._dog_level_num;
  return 1;
}
int ref_siftgpu_level(int octave_index, int level, int data, float* out, int* w, int* h) {
  if (!g_sift) return 0;
  PyramidCU* p = g_sift->pyramid;
  CuTexImage* base = p->GetBaseLevel(p->_octave_min + octave_index, data);
  if (!base) return 0;
  CuTexImage* t = base + level;
  const int ch = data == 2 ? 4 : (data == 3 ? 2 : 1);
  *w = t->GetImgWidth(); *h = t->GetImgHeight();
  if (!t->_cuData || t->GetDataSize() < (*w) * (*h) * ch * 4) return 0;
  memcpy(out, t->_cuData, (size_t)(*w) * (*h) * ch * 4);
  return (*w) * (*h) * ch;
}
// features per (octave, dog level) of the last run, after LimitFeatureCount / ReshapeFeatureListCPU
int ref_siftgpu_level_counts(int* counts, int capacity) {
  if (!g_sift) return 0;
  const int n = g_sift->pyramid->_octave_num * g_sift->param._dog_level_num;
  const int* c = g_sift->pyramid->GetLevelFeatureNum();
  for (int i = 0; i < n && i < capacity; ++i) counts[i] = c[i];
  return n;
}
// the filter taps ProgramCU::CreateFilterKernel builds for a sigma (ProgramCU.cu:370-398)
int ref_siftgpu_filter_kernel(float sigma, float* kernel) {
  int width = 0;
  ProgramCU::CreateFilterKernel(sigma, kernel, width);
  return width;
}
void ref_siftgpu_params(float* sigmas, int* n_sigma, float* sigma_skip0, float* sigma_skip1, float* dog_threshold,
                        float* edge_threshold, float* initial_sigma) {
  SiftParam p;
  p._dog_level_num = 5;
  p._edge_threshold = 10.0f;
  GlobalUtil::_octave_min_default = -1;
  p.ParseSiftParam();
  *n_sigma = p._sigma_num;
  for (int i = 0; i < p._sigma_num; ++i) sigmas[i] = p._sigma[i];
  *sigma_skip0 = p._sigma_skip0; *sigma_skip1 = p._sigma_skip1;
  *dog_threshold = p._dog_threshold; *edge_threshold = p._edge_threshold;
  *initial_sigma = p.GetInitialSmoothSigma(-1);
}

}  // extern "C"
