// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of this repository's gather kernels (dev microbenchmark).
// MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read; "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern".  Every kernel below reads each byte of a 1-GiB buffer (4x the
// Infinity Cache) exactly once; the expected FETCH_SIZE is therefore 1 GiB (1048576 KiB) if the counter is exact, 524288 if it halves.
//   build:  hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
//   run:    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fetch_calib -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

// (a) streaming: lane l of a wave reads 16 B at wave_base + 16 l  (1 KiB per wave instruction)
__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ x, float* out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) out[0] = s;
}
// (b) texel gather as warp_vec8_kernel<64>: 8 lanes own a 256-B texel (two 16-B loads per lane, 32 B apart... +16), the 8 texels of a
// wave are ROWS apart (a wave touches 8 separate 256-B segments per instruction pair); every texel of the buffer is read once
template <int LANES_PER_TEXEL, int LOADS>      // texel bytes = LANES_PER_TEXEL * LOADS * 16
__global__ __launch_bounds__(256) void texel_gather(const char* __restrict__ x, float* out, size_t ntexel, size_t row_texels) {
  constexpr int TB = LANES_PER_TEXEL * LOADS * 16;
  constexpr int TPB = 256 / LANES_PER_TEXEL;          // texels per block per step
  const int sub = threadIdx.x % LANES_PER_TEXEL, tl = threadIdx.x / LANES_PER_TEXEL;
  const size_t rows = ntexel / row_texels;
  float s = 0.f;
  // texel t of step k: row = (tl + TPB * k') ... neighbouring lanes-groups take DIFFERENT rows, same column; columns advance per block step
  for (size_t col = blockIdx.x; col < row_texels; col += gridDim.x)
    for (size_t r = tl; r < rows; r += TPB) {
      const char* p = x + (r * row_texels + col) * TB + sub * (LOADS * 16);
#pragma unroll
      for (int j = 0; j < LOADS; ++j) { const float4 v = *reinterpret_cast<const float4*>(p + 16 * j); s += v.x + v.y + v.z + v.w; }
    }
  if (s == 12345.f) out[0] = s;
}
// (c) 4 bytes per lane, the lanes of a wave 4 KiB apart (the consistency filter's depth taps when nothing coalesces); reads every dword once
__global__ __launch_bounds__(256) void dword_scatter(const float* __restrict__ x, float* out, size_t n) {
  const size_t stride = 1024;   // floats
  float s = 0.f;
  const size_t lanes = (size_t)gridDim.x * 256;
  for (size_t base = 0; base < n; base += lanes * stride) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t k = 0; k < stride; ++k) { const size_t i = base + t * stride + k; if (i < n) s += x[i]; }
  }
  if (s == 12345.f) out[0] = s;
}

int main() {
  const size_t bytes = 1ull << 30;
  char* x; float* out;
  hipMalloc(&x, bytes); hipMalloc(&out, 64);
  hipMemset(x, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const float4*)x, out, bytes / 16);
    hipLaunchKernelGGL((texel_gather<8, 2>), dim3(2048), dim3(256), 0, 0, x, out, bytes / 256, (size_t)2048);   // 256-B texels, 32 B per lane
    hipLaunchKernelGGL((texel_gather<8, 1>), dim3(2048), dim3(256), 0, 0, x, out, bytes / 128, (size_t)2048);   // 128-B texels, 16 B per lane (C = 32 at 4 ch/lane)
    hipLaunchKernelGGL((texel_gather<4, 1>), dim3(2048), dim3(256), 0, 0, x, out, bytes / 64, (size_t)2048);    // 64-B texels (C = 16)
    hipLaunchKernelGGL((texel_gather<4, 2>), dim3(2048), dim3(256), 0, 0, x, out, bytes / 128, (size_t)2048);   // 128-B texels, 32 B per lane (C = 32 vec8)
    hipLaunchKernelGGL(dword_scatter, dim3(256), dim3(256), 0, 0, (const float*)x, out, bytes / 4 / 4);         // a quarter of the buffer
    hipDeviceSynchronize();
  }
  printf("done: every kernel read %zu bytes once (dword_scatter: %zu)\n", bytes, bytes / 4);
  return 0;
}
