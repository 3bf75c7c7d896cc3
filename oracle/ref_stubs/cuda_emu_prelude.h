// oracle/ref_stubs/cuda_emu_prelude.h -- TEST INFRASTRUCTURE ONLY.
// A tiny CUDA-on-CPU emulation plus stand-ins for CuTexImage / SiftMatchGPU / cv::DMatch, so that the SiftGPU
// matcher the reference ships in its tree compiles FROM WHERE IT LIES (/root/reference) as plain C++ into
// oracle/_ref/libref_siftmatch.so:
//   MultiplyDescriptor_Kernel   external/SiftGPU/src/SiftGPU/ProgramCU.cu:1395-1482
//   RowMatch_Kernel             external/SiftGPU/src/SiftGPU/ProgramCU.cu:1682-1743
//   ColMatch_Kernel             external/SiftGPU/src/SiftGPU/ProgramCU.cu:1755-1782
//   SiftMatchCU::SetDescriptors (u8, float), GetSiftMatch, GetBestMatch
//                               external/SiftGPU/src/SiftGPU/SiftMatchCU.cpp:71-100, 133-177
//   SiftGPUWrapper::match       src/sift_gpu_wrapper.cpp:169-227
// Emulation: one OS thread per CUDA thread of a block, blocks one after the other; __shared__ = static storage,
// __syncthreads() = a barrier over the block's threads, textures = plain pointers.  The three kernel launchers
// (ProgramCU::MultiplyDescriptor / GetRowMatch / GetColMatch, :1490-1506, :1746-1754, :1784-1793) use the <<< >>>
// syntax and are restated in siftmatch_glue.cpp with the same grid / block shapes.
#ifndef REF_STUB_CUDA_EMU_PRELUDE_H
#define REF_STUB_CUDA_EMU_PRELUDE_H
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define ROS_WARN(...)
#define ROS_ERROR(...)
#define ROS_DEBUG(...)
#define ROS_INFO(...)

// ---- CUDA vocabulary ------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint4 { unsigned x, y, z, w; };
struct int3 { int x, y, z; };
static inline int3 make_int3(int a, int b, int c) { int3 r; r.x = a; r.y = b; r.z = c; return r; }
extern thread_local dim3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
#define __global__
#define __shared__ static
#define __mul24(a, b) ((a) * (b))
enum cudaTextureReadMode { cudaReadModeElementType };
template <class T, int D, cudaTextureReadMode M> struct texture { const T* ptr = nullptr; };
template <class T, int D, cudaTextureReadMode M> static inline T tex1Dfetch(const texture<T, D, M>& t, int i) { return t.ptr[i]; }
// CUDA's mixed-type overloads used by the kernels
static inline int max(int a, int b) { return a > b ? a : b; }
static inline double min(float a, double b) { return (double)a < b ? (double)a : b; }

void cuemu_barrier();
#define __syncthreads() cuemu_barrier()
void cuemu_launch(dim3 grid, dim3 block, const std::function<void()>& kernel_body);

// ---- CuTexImage: a linear device buffer -----------------------------------------------------------
struct CuTexImage {
  void* _cuData = nullptr;
  int _w = 0, _h = 0, _nc = 0;
  std::vector<unsigned char> store;
  void InitTexture(int width, int height, int nchannel = 1) {
    _w = width; _h = height; _nc = nchannel;
    store.assign((size_t)width * height * nchannel * sizeof(float) + 64, 0);
    _cuData = store.data();
  }
  int GetImgWidth() const { return _w; }
  int GetImgHeight() const { return _h; }
  void CopyFromHost(const void* p) { std::memcpy(_cuData, p, (size_t)_w * _h * _nc * sizeof(float)); }
  void CopyToHost(void* p) { std::memcpy(p, _cuData, (size_t)_w * _h * _nc * sizeof(float)); }
  template <class T, int D, cudaTextureReadMode M> void BindTexture(texture<T, D, M>& t) { t.ptr = (const T*)_cuData; }
};

struct ProgramCU {
  static void MultiplyDescriptor(CuTexImage* des1, CuTexImage* des2, CuTexImage* texDot, CuTexImage* texCRT);
  static void GetRowMatch(CuTexImage* texDot, CuTexImage* texMatch, float distmax, float ratiomax);
  static void GetColMatch(CuTexImage* texCRT, CuTexImage* texMatch, float distmax, float ratiomax);
};

// ---- SiftMatchCU as declared in SiftMatchCU.h:29-64 (members the compiled functions touch) ------------
using std::vector;
struct SiftMatchGPU { virtual ~SiftMatchGPU() {} };
class SiftMatchCU : public SiftMatchGPU {
 public:
  CuTexImage _texLoc[2], _texDes[2], _texDot, _texMatch[2], _texCRT;
  int _max_sift = 4096;  // SiftMatchGPU(4096), sift_gpu_wrapper.cpp:231
  int _num_sift[2] = {0, 0}, _id_sift[2] = {0, 0}, _have_loc[2] = {0, 0};
  int _initialized = 1;
  vector<int> sift_buffer;
  int GetBestMatch(int max_match, int match_buffer[][2], float distmax, float ratiomax, int mbm);
  void SetDescriptors(int index, int num, const unsigned char* descriptor, int id = -1);
  void SetDescriptors(int index, int num, const float* descriptor, int id = -1);
  int GetSiftMatch(int max_match, int match_buffer[][2], float distmax = 0.7, float ratiomax = 0.8, int mbm = 1);
};

namespace cv {
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0.f; };
}
struct QMutexStub { void lock() {} void unlock() {} };
class SiftGPUWrapper {
 public:
  bool isMatcherInitialized = true;
  void initializeMatcher() {}
  QMutexStub gpu_mutex;
  SiftMatchCU* matcher = nullptr;
  int match(const std::vector<float>& descriptors1, int num1, const std::vector<float>& descriptors2, int num2,
            std::vector<cv::DMatch>* matches);
};
#endif
