// oracle/ref_stubs/siftgpu_emu_prelude.h -- TEST INFRASTRUCTURE ONLY.
// CUDA-on-CPU emulation + GL / CuTexImage stand-ins so that the SIFT EXTRACTION pipeline the reference vendors --
//   external/SiftGPU/src/SiftGPU/ProgramCU.cu:32-1191   (FilterH / FilterV / Upsample / Downsample / ComputeDOG /
//                                                         ComputeKEY / InitHist / ReduceHist / ListGen /
//                                                         ComputeOrientation / ComputeDescriptor / NormalizeDescriptor
//                                                         kernels AND their ProgramCU:: launchers)
//   external/SiftGPU/src/SiftGPU/PyramidCU.cpp          (pyramid allocation, BuildPyramid, DetectKeypointsEX, feature
//                                                         lists, orientations, descriptors, keypoint download)
//   external/SiftGPU/src/SiftGPU/SiftPyramid.cpp:49-262 (SiftPyramid::RunSIFT, LimitFeatureCount, ...)
//   external/SiftGPU/src/SiftGPU/SiftGPU.cpp:411-473,1200-1203 (SiftParam), GlobalUtil.cpp:48-139 (defaults)
// -- compiles FROM WHERE IT LIES (/root/reference) as plain C++ into oracle/_ref/libref_siftgpu.so (oracle/Makefile).
// The headers GlobalUtil.h, SiftGPU.h, SiftPyramid.h, CuTexImage.h, ProgramCU.h, PyramidCU.h are the reference's own
// (-I); this file supplies what they expect from CUDA / OpenGL.
//
// Emulation: every CUDA thread of a block is a user-level fiber on ONE OS thread (a 12-instruction context switch);
// __syncthreads() yields to the next fiber of the block, so a kernel's __shared__ arrays (plain statics) behave as on the
// device.  `K<<<grid, block>>>(args)` is rewritten by sed into CUEMU_KERNEL_CALL(grid, block, K)(args).
// Device math maps to glibc's float functions (exp -> expf ...): results are those of THIS emulation, not of an NVIDIA
// GPU -- see DESIGN.md 4.11 for what is compared exactly and what within a tolerance.
#ifndef REF_STUB_SIFTGPU_EMU_PRELUDE_H
#define REF_STUB_SIFTGPU_EMU_PRELUDE_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <utility>
#include <vector>

// ---- OpenGL vocabulary the headers mention -------------------------------------------------------------------------
typedef unsigned int GLuint;
typedef unsigned int GLenum;
typedef int GLint;
#define GL_TEXTURE_RECTANGLE_ARB 0x84F5
#define GL_RGBA32F_ARB 0x8814
#define GL_LUMINANCE 0x1909
#define GL_RGBA 0x1908
static inline void glDeleteBuffers(int, const GLuint*) {}
static inline void glGenBuffers(int, GLuint* b) { if (b) *b = 0; }
#define CUDA_SIFTGPU_ENABLED 1

// ---- CUDA vocabulary ---------------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float a, float b) { float2 r; r.x = a; r.y = b; return r; }
static inline float4 make_float4(float a, float b, float c, float d) { float4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }
static inline int4 make_int4(int a, int b, int c, int d) { int4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }
extern dim3 threadIdx, blockIdx, blockDim, gridDim;   // one OS thread: plain globals, set by the fiber scheduler
#define __global__
#define __device__
#define __constant__
#define __shared__ static
#define __mul24(a, b) ((a) * (b))
#define __fdividef(a, b) ((a) / (b))
#define __sincosf cuemu_sincosf   /* (glibc owns the name __sincosf) */
static inline void cuemu_sincosf(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }
static inline float rsqrt(float a) { return 1.0f / sqrtf(a); }
static inline float __int_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
// CUDA's float overloads of min / max (the int ones come from <algorithm> through `using namespace std`)
static inline float max(float a, float b) { return a > b ? a : b; }   // fmaxf semantics differ only for NaN
static inline float min(float a, float b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
enum cudaTextureReadMode { cudaReadModeElementType, cudaReadModeNormalizedFloat };
enum cudaMemcpyKind { cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
struct cudaArray;
struct textureReference {
  const void* ptr = nullptr;
  int bytes = 0;                                // size of the bound allocation (cudaBindTexture(..., _numBytes))
  int width = 0, height = 0, pitch_elems = 0;   // 2-D binding (cudaBindTexture2D over linear memory, point sampling)
};
template <class T, int D, cudaTextureReadMode M> struct texture : textureReference {};
// a fetch outside the bound range returns zero (linear-memory texture semantics): the kernels do fetch index - width on
// row 0 and index + 1 past the last row (ComputeDOG_Kernel); inside the allocation but outside the image they see what
// the allocation holds (zeros / an earlier, larger image -- InitTexture never shrinks)
template <class T, int D, cudaTextureReadMode M> static inline T tex1Dfetch(const texture<T, D, M>& t, int i) {
  if (i < 0 || (size_t)(i + 1) * sizeof(T) > (size_t)t.bytes) { T z; memset(&z, 0, sizeof(T)); return z; }
  return ((const T*)t.ptr)[i];
}
static inline float tex1Dfetch(const texture<unsigned char, 1, cudaReadModeNormalizedFloat>& t, int i) {
  if (i < 0 || i >= t.bytes) return 0.f;
  return ((const unsigned char*)t.ptr)[i] / 255.0f;
}
// unnormalised coordinates, cudaFilterModePoint, cudaAddressModeClamp: texel floor(x), floor(y)
template <class T> static inline T tex2D(const texture<T, 2, cudaReadModeElementType>& t, float x, float y) {
  int ix = (int)floorf(x), iy = (int)floorf(y);
  ix = ix < 0 ? 0 : (ix >= t.width ? t.width - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= t.height ? t.height - 1 : iy);
  return ((const T*)t.ptr)[(size_t)iy * t.pitch_elems + ix];
}
template <class S> static inline void cudaMemcpyToSymbol(S& sym, const void* src, size_t bytes, size_t off, cudaMemcpyKind) {
  memcpy((char*)&sym + off, src, bytes);
}

// ---- the fiber scheduler -------------------------------------------------------------------------------------------
void cuemu_barrier();
#define __syncthreads() cuemu_barrier()
struct CuemuBody { virtual void run() = 0; virtual ~CuemuBody() {} };
void cuemu_run(dim3 grid, dim3 block, CuemuBody& body);
template <class F> struct CuemuLauncher {
  dim3 grid, block;
  F kernel;
  template <class... A> void operator()(A... args) {
    struct Body : CuemuBody {
      F& k; std::tuple<A...> a;
      Body(F& kk, A... aa) : k(kk), a(aa...) {}
      void run() override { std::apply(k, a); }
    } body(kernel, args...);
    cuemu_run(grid, block, body);
  }
};
template <class F> static inline CuemuLauncher<F> cuemu_make_launcher(dim3 g, dim3 b, F f) { return CuemuLauncher<F>{g, b, f}; }
// generic lambda: overloaded kernels (ComputeDOG_Kernel) and template instances (FilterH<FW>) resolve at the call
#define CUEMU_KERNEL_CALL(G, B, ...) cuemu_make_launcher(G, B, [](auto... cuemu_a) { __VA_ARGS__(cuemu_a...); })

// ---- GL-side classes the pyramid code mentions -----------------------------------------------------------------------
class GLTexImage {
 public:
  void InitTexture(int, int) {}
  void SetImageSize(int, int) {}
  int GetTexWidth() { return 0; }
  int GetTexHeight() { return 0; }
  int GetImgWidth() { return 0; }
  int GetImgHeight() { return 0; }
  void CopyFromPBO(GLuint, int, int, GLenum) {}
};
// what PyramidCU reads of the input image (GLTexImage.h:108-135): the luminance floats SetImageData prepared
class GLTexInput : public GLTexImage {
 public:
  int _down_sampled = 0, _rgb_converted = 1, _data_modified = 0;
  float* _converted_data = nullptr;
  const void* _pixel_data = nullptr;
  int _imgWidth = 0, _imgHeight = 0;
  static int TruncateWidthCU(int w) { return w & 0xfffffffc; }   // GLTexImage.h:125
  int GetImgWidth() { return _imgWidth; }
  int GetImgHeight() { return _imgHeight; }
  int CopyToPBO(GLuint, int, int, GLenum = GL_RGBA) { return 0; }
};
#endif
