// oracle/ref_stubs/siftmatch_glue.cpp -- TEST INFRASTRUCTURE ONLY (see cuda_emu_prelude.h).
thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int count = 0, waiting = 0, generation = 0;
  void wait() {
    std::unique_lock<std::mutex> l(m);
    const int gen = generation;
    if (++waiting == count) { waiting = 0; ++generation; cv.notify_all(); }
    else cv.wait(l, [&] { return gen != generation; });
  }
} g_barrier;
}  // namespace
void cuemu_barrier() { g_barrier.wait(); }

// Every emulated thread of the block walks over all blocks of the grid; a second barrier (block_end) keeps the
// block's static __shared__ storage alive until all of its threads are done.  A kernel that returns early must do so
// after its last __syncthreads() (true for the three kernels compiled here).
void cuemu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  blockDim = block; gridDim = grid;
  g_barrier.count = nthreads;
  static Barrier block_end;
  block_end.count = nthreads;
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) {
    pool.emplace_back([&, t] {
      threadIdx = dim3((unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y));
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
          blockIdx = dim3(bx, by, 0);
          body();
          block_end.wait();
        }
    });
  }
  for (auto& th : pool) th.join();
}

// ProgramCU.cu:1490-1506
void ProgramCU::MultiplyDescriptor(CuTexImage* des1, CuTexImage* des2, CuTexImage* texDot, CuTexImage* texCRT) {
  int num1 = des1->GetImgWidth() / 8;
  int num2 = des2->GetImgWidth() / 8;
  dim3 grid((num2 + MULT_BLOCK_DIMX - 1) / MULT_BLOCK_DIMX, (num1 + MULT_BLOCK_DIMY - 1) / MULT_BLOCK_DIMY);
  dim3 block(MULT_TBLOCK_DIMX, MULT_TBLOCK_DIMY);
  texDot->InitTexture(num2, num1);
  if (texCRT) texCRT->InitTexture(num2, (num1 + MULT_BLOCK_DIMY - 1) / MULT_BLOCK_DIMY, 32);
  des1->BindTexture(texDes1);
  des2->BindTexture(texDes2);
  int* d_result = (int*)texDot->_cuData;
  int3* d_temp = texCRT ? (int3*)texCRT->_cuData : NULL;
  cuemu_launch(grid, block, [=] { MultiplyDescriptor_Kernel(d_result, num1, num2, d_temp); });
}
// ProgramCU.cu:1746-1754
void ProgramCU::GetRowMatch(CuTexImage* texDot, CuTexImage* texMatch, float distmax, float ratiomax) {
  int num1 = texDot->GetImgHeight();
  int num2 = texDot->GetImgWidth();
  dim3 grid(1, num1 / ROWMATCH_BLOCK_HEIGHT);
  dim3 block(ROWMATCH_BLOCK_WIDTH, ROWMATCH_BLOCK_HEIGHT);
  texDot->BindTexture(texDOT);
  int* d_dot = (int*)texDot->_cuData;
  int* d_res = (int*)texMatch->_cuData;
  cuemu_launch(grid, block, [=] { RowMatch_Kernel(d_dot, d_res, num2, distmax, ratiomax); });
}
// ProgramCU.cu:1784-1793
void ProgramCU::GetColMatch(CuTexImage* texCRT, CuTexImage* texMatch, float distmax, float ratiomax) {
  int height = texCRT->GetImgHeight();
  int num2 = texCRT->GetImgWidth();
  dim3 grid((num2 + COLMATCH_BLOCK_WIDTH - 1) / COLMATCH_BLOCK_WIDTH);
  dim3 block(COLMATCH_BLOCK_WIDTH);
  int3* d_crt = (int3*)texCRT->_cuData;
  int* d_res = (int*)texMatch->_cuData;
  cuemu_launch(grid, block, [=] { ColMatch_Kernel(d_crt, d_res, height, num2, distmax, ratiomax); });
}

extern "C" int ref_sift_match(const float* d1, int n1, const float* d2, int n2, int32_t* mq, int32_t* mt, float* md) {
  SiftMatchCU m;
  SiftGPUWrapper w;
  w.matcher = &m;
  std::vector<float> a(d1, d1 + (size_t)n1 * 128), b(d2, d2 + (size_t)n2 * 128);
  std::vector<cv::DMatch> out;
  w.match(a, n1, b, n2, &out);
  for (size_t i = 0; i < out.size(); ++i) { mq[i] = out[i].queryIdx; mt[i] = out[i].trainIdx; md[i] = out[i].distance; }
  return (int)out.size();
}
