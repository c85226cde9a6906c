// Calibration: sustained rate of v_mfma_f32_16x16x4_f32 / 32x32x2 in a bare loop (dev microbenchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, int mfma_per_iter, double flop_per_mfma, float* d) {
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * mfma_per_iter * flop_per_mfma;
  printf("%-30s blocks %5d (%d waves/SIMD): %7.1f TFLOP/s  (%.2f ms)\n", name, blocks, blocks / 256, fl / ms / 1e9, ms);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    run("16x16x4 f32, 1 accumulator", k16<1>, 256 * bpc, 16, 2048.0, d);
    run("16x16x4 f32, 2 accumulators", k16<2>, 256 * bpc, 16, 2048.0, d);
    run("16x16x4 f32, 4 accumulators", k16<4>, 256 * bpc, 16, 2048.0, d);
    run("32x32x2 f32, 1 accumulator", k32<1>, 256 * bpc, 8, 4096.0, d);
    run("32x32x2 f32, 2 accumulators", k32<2>, 256 * bpc, 8, 4096.0, d);
  }
  return 0;
}
